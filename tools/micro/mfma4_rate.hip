// Issue rate of v_mfma_f64_4x4x4_4b_f64 against v_mfma_f64_16x16x4_f64 (independent accumulators, 1 and 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, int iters)
{
    const double x = threadIdx.x * 1e-3, y = 1.0 + x;
    double r = 0;
    if (MODE == 0)
    {
        f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        for (int i = 0; i < iters; ++i)
        {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    }
    else
    {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(j & 1 ? x : y, j & 2 ? x : y, a[j], 0, 0, 0);
        for (int j = 0; j < 8; ++j) r += a[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
float run(double *d, int blocks, int threads, int iters)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    double *d; (void)hipMalloc(&d, 256 * 512 * sizeof(double));
    const int iters = 20000;
    // per wave and iteration: 4 x 16x16x4 (2048 flop each) or 8 x 4x4x4_4b (512 flop each)
    for (int threads : {256, 512})
    {
        const float t16 = run<0>(d, 256, threads, iters), t4 = run<1>(d, 256, threads, iters);
        const double waves = 256.0 * threads / 64;
        printf("%d waves/SIMD: 16x16x4 %.3f ms = %.1f cycles/instr (at 2.4 GHz, per SIMD)  4x4x4_4b %.3f ms = %.1f cycles/instr\n", threads / 256, t16,
               t16 * 1e-3 * 2.4e9 / (4.0 * iters) / (waves / 1024), t4, t4 * 1e-3 * 2.4e9 / (8.0 * iters) / (waves / 1024));
    }
    return 0;
}
