// Microbenchmark: what a launch shaped like k_fused costs before it does anything.  Kernel durations from the
// dispatch's own begin / end timestamps (hipExtLaunchKernelGGL events), averaged over launches that alternate with a
// small "other" kernel (so caches are as cold as between the kernels of an evaluation).
//   empty         : nothing
//   lds           : same with 153.6 KB of dynamic LDS (one workgroup per CU)
//   chain<D>      : D dependent global loads (pointer chase through a table written by the host), then one store
//   store326      : every workgroup writes 326 doubles (the tile partial), nothing else
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_other(double *p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0; }
__global__ __launch_bounds__(768) void k_empty(double *out) {}
// every wave spins for `ticks` of the 100 MHz real-time counter: a kernel of known GPU-side duration
__global__ __launch_bounds__(768) void k_spin(double *out, long long ticks)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    if (ticks < 0) out[0] = 1.0;
}
__global__ __launch_bounds__(768) void k_lds(double *out)
{
    extern __shared__ double lds[];
    if (out == nullptr) lds[threadIdx.x] = 1.0; // never: keeps the allocation
}
template <int D>
__global__ __launch_bounds__(768) void k_chain(const int *__restrict__ next, double *out)
{
    extern __shared__ double lds[];
    int i = blockIdx.x * 16 + (threadIdx.x >> 6);
#pragma unroll
    for (int d = 0; d < D; ++d) i = next[i];
    if (i == -1) out[0] = 1.0;
}
__global__ __launch_bounds__(768) void k_store(double *out)
{
    extern __shared__ double lds[];
    if (threadIdx.x < 326) out[(size_t)blockIdx.x * 328 + threadIdx.x] = (double)threadIdx.x;
}
template <class F>
static double timed(F launch, double *scratch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double sum = 0; int n = 0;
    for (int it = 0; it < 60; ++it)
    {
        hipLaunchKernelGGL(k_other, dim3(64), dim3(256), 0, 0, scratch);
        launch(a, b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it >= 10) { sum += ms; ++n; }
    }
    return sum / n * 1e3;
}
int main()
{
    const int G = 255, T = 768; const size_t LDS = 153600;
    double *out; hipMalloc(&out, (size_t)G * 328 * 8 + 64 * 256 * 8); hipMemset(out, 0, (size_t)G * 328 * 8 + 64 * 256 * 8);
    double *scratch = out + (size_t)G * 328;
    // pointer-chase table: 4 levels, each level in its own region, strided by 64 ints (one cache line per entry)
    const int NE = G * 16, LV = 5, STR = 16;
    std::vector<int> h((size_t)LV * NE * STR, 0);
    for (int l = 0; l < LV; ++l)
        for (int e = 0; e < NE; ++e) h[((size_t)l * NE + e) * STR] = (int)((((size_t)(l + 1) % LV) * NE + (e * 7 + 3) % NE) * STR);
    int *next; hipMalloc(&next, h.size() * 4); hipMemcpy(next, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute((const void *)k_store, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute((const void *)k_chain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute((const void *)k_chain<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute((const void *)k_chain<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute((const void *)k_chain<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    printf("empty 1x64        %.2f us\n", timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, a, b, 0, out); }, scratch));
    printf("empty 255x768     %.2f us\n", timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(G), dim3(T), 0, 0, a, b, 0, out); }, scratch));
    printf("lds   255x768     %.2f us\n", timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_lds, dim3(G), dim3(T), LDS, 0, a, b, 0, out); }, scratch));
    printf("chain1            %.2f us\n", timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_chain<1>, dim3(G), dim3(T), LDS, 0, a, b, 0, (const int *)next, out); }, scratch));
    printf("chain2            %.2f us\n", timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_chain<2>, dim3(G), dim3(T), LDS, 0, a, b, 0, (const int *)next, out); }, scratch));
    printf("chain3            %.2f us\n", timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_chain<3>, dim3(G), dim3(T), LDS, 0, a, b, 0, (const int *)next, out); }, scratch));
    printf("chain4            %.2f us\n", timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_chain<4>, dim3(G), dim3(T), LDS, 0, a, b, 0, (const int *)next, out); }, scratch));
    printf("store326          %.2f us\n", timed([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_store, dim3(G), dim3(T), LDS, 0, a, b, 0, out); }, scratch));
    // wall clock per launch of back-to-back launches in one stream (what a kernel boundary costs in a pipeline)
    auto wall = [&](int G2, int T2, size_t lds, int kind) {
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const int N = 2000;
        hipEventRecord(a);
        for (int i = 0; i < N; ++i)
        {
            if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(G2), dim3(T2), 0, 0, out);
            else if (kind == 1) hipLaunchKernelGGL(k_lds, dim3(G2), dim3(T2), lds, 0, out);
            else hipLaunchKernelGGL(k_store, dim3(G2), dim3(T2), lds, 0, out);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return ms / N * 1e3;
    };
    // GPU-side gap between dependent kernels: back-to-back kernels that each spin for 10 us (1000 ticks)
    {
        (void)hipDeviceSynchronize();
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        const int N = 1000;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_spin, dim3(G), dim3(T), 0, 0, out, 1000LL);
        (void)hipEventRecord(a);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_spin, dim3(G), dim3(T), 0, 0, out, 1000LL);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("back-to-back 10 us spin kernels: %.2f us per launch -> %.2f us between kernels\n", ms / N * 1e3, ms / N * 1e3 - 10.0);
    }
    printf("back-to-back empty 1x64      %.2f us per launch\n", wall(1, 64, 0, 0));
    printf("back-to-back empty 255x768   %.2f us per launch\n", wall(G, T, 0, 0));
    printf("back-to-back lds 255x768     %.2f us per launch\n", wall(G, T, LDS, 1));
    printf("back-to-back store 255x768   %.2f us per launch\n", wall(G, T, LDS, 2));
    return 0;
}
