// Microbenchmark: can v_mfma_f64_16x16x4_f64 issued by one wave overlap FP64 VALU FMAs issued by another wave of the
// same SIMD?  4 waves per workgroup = 1 per SIMD ... 8 waves = 2 per SIMD.  Modes: 0 all waves MFMA, 1 all waves FMA,
// 2 even waves MFMA / odd waves FMA (wave w and w+4 share a SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || (MODE == 2 && (wave < 4));
    double r = 0;
    if (do_mfma)
    {
        f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const double x = threadIdx.x * 1e-3, y = 1.0 + x;
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
        double v[16];
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3 + j;
        const double m = 1.0000001, c = 1e-9;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_fma(v[j], m, c);
        for (int j = 0; j < 16; ++j) r += v[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
float run(double *d, int blocks, int threads, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    double *d; hipMalloc(&d, 256 * 512 * sizeof(double));
    const int iters = 20000;
    // per wave: MFMA mode 4 x 64 cycles x iters; FMA mode 16 x 4 cycles x iters (same pipe time if rates are as documented)
    printf("1 wave/SIMD  mfma %.3f ms  fma %.3f ms\n", run<0>(d, 256, 256, iters), run<1>(d, 256, 256, iters));
    printf("2 waves/SIMD mfma %.3f ms  fma %.3f ms  mixed (wave w mfma, w+4 fma) %.3f ms\n", run<0>(d, 256, 512, iters),
           run<1>(d, 256, 512, iters), run<2>(d, 256, 512, iters));
    return 0;
}
