// v_mfma_f64_4x4x4_4b_f64 with N independent accumulator chains (N = 2, 4, 7, 8, 14), 1 / 2 / 3 waves per SIMD: how many
// chains does the instruction need to issue at its full rate?  (Question behind the 7-instruction outer product.)
// Also: the same chains with two independent FP64 FMAs of the SAME wave between the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, bool FMA>
__global__ __launch_bounds__(768) void k(double *out, int iters)
{
    const double x = threadIdx.x * 1e-3, y = 1.0 + x;
    double a[N], v0 = x, v1 = y, v2 = x + 2, v3 = y + 3;
    for (int j = 0; j < N; ++j) a[j] = 0;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j)
        {
            a[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(j & 1 ? x : y, j & 2 ? x : y, a[j], 0, 0, 0);
            if (FMA)
            {
                if (j & 1) { v0 = __builtin_fma(v0, 1.0000001, 1e-9); v1 = __builtin_fma(v1, 1.0000001, 1e-9); }
                else { v2 = __builtin_fma(v2, 1.0000001, 1e-9); v3 = __builtin_fma(v3, 1.0000001, 1e-9); }
            }
        }
    double r = v0 + v1 + v2 + v3;
    for (int j = 0; j < N; ++j) r += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int N, bool FMA>
float run(double *d, int threads, int iters)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<N, FMA>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<N, FMA>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
template <int N>
void row(double *d)
{
    const int iters = 8000;
    printf("chains %2d:", N);
    for (int threads : {256, 512, 768})
    {
        const float t = run<N, false>(d, threads, iters), tf = run<N, true>(d, threads, iters);
        // cycles of the SIMD per MFMA (all waves of the SIMD together), at 2.4 GHz
        const double per = t * 1e-3 * 2.4e9 / ((double)N * iters * (threads / 256));
        const double perf = tf * 1e-3 * 2.4e9 / ((double)N * iters * (threads / 256));
        printf("   %d w/SIMD %.1f (+2 FMA: %.1f)", threads / 256, per, perf);
    }
    printf("\n");
}
int main()
{
    double *d; (void)hipMalloc(&d, 256 * 768 * sizeof(double));
    row<1>(d); row<2>(d); row<4>(d); row<7>(d); row<8>(d); row<14>(d);
    return 0;
}
