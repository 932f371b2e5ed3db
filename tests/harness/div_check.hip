// pixel_math.h: reciprocal() and quotient() -- the runtime's fp64 division sequence without range scaling and
// special-case fix-up -- against the compiler's IEEE division, bit for bit, over the operand ranges they are used on
// (unit-ray / depth / focal-length magnitudes, sample counts) and well beyond.  Also tri_decode against the row search.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pixel_math.h"

__global__ void k_div(const double *n, const double *d, int count, unsigned long long *mism)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double x = d[i], y = n[i];
    const double r0 = 1.0 / x, r1 = mbavo::reciprocal(x);
    const double q0 = y / x, q1 = mbavo::quotient(y, x);
    if (__double_as_longlong(r0) != __double_as_longlong(r1)) atomicAdd(&mism[0], 1ull);
    if (__double_as_longlong(q0) != __double_as_longlong(q1)) atomicAdd(&mism[1], 1ull);
}

__global__ void k_tri(int nd, unsigned long long *mism)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nd * (nd + 1) / 2) return;
    int i = 0, rem = e;
    while (rem >= nd - i) { rem -= nd - i; ++i; }
    int a, b;
    mbavo::tri_decode(e, nd, a, b);
    if (a != i || b != i + rem) atomicAdd(&mism[2], 1ull);
}

static double rnd(double lo, double hi) { return lo + (hi - lo) * (double)rand() / RAND_MAX; }

int main()
{
    const int N = 1 << 22;
    std::vector<double> hn(N), hd(N);
    srand(7);
    for (int i = 0; i < N; ++i)
    {
        const int kind = i & 7;
        double d, n;
        if (kind == 0) { d = rnd(0.05, 1.0); n = 1.0; }                       // rho_z of a sample
        else if (kind == 1) { d = rnd(1.0, 4.0); n = 1.0; }                   // sqrt(1 + xh^2 + yh^2)
        else if (kind == 2) { d = rnd(0.01, 100.0); n = rnd(-1e4, 1e4); }     // depths
        else if (kind == 3) { d = rnd(50.0, 4000.0); n = rnd(-4000.0, 4000.0); } // (px - cx) / fx
        else if (kind == 4) { d = (double)(1 + rand() % 64); n = rnd(0.0, 255.0 * 64); } // isum / S
        else if (kind == 5) { d = -rnd(1e-6, 1e6); n = rnd(-1e6, 1e6); }      // negative denominators
        else if (kind == 6) { d = rnd(1e-30, 1e-20); n = rnd(1e-10, 1e10); }  // far outside the use, still normal
        else { d = rnd(1e20, 1e30); n = (i & 8) ? 0.0 : rnd(-1.0, 1.0); }     // large denominators, zero numerators
        hn[i] = n; hd[i] = d;
    }
    double *dn, *dd; unsigned long long *dm, hm[3] = {0, 0, 0};
    if (hipMalloc(&dn, N * 8) != hipSuccess || hipMalloc(&dd, N * 8) != hipSuccess || hipMalloc(&dm, sizeof(hm)) != hipSuccess) return 2;
    (void)hipMemcpy(dn, hn.data(), N * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dd, hd.data(), N * 8, hipMemcpyHostToDevice);
    (void)hipMemset(dm, 0, sizeof(hm));
    hipLaunchKernelGGL(k_div, dim3(N / 256), dim3(256), 0, 0, dn, dd, N, dm);
    for (int nd : {1, 2, 12, 13, 24, 25, 37, 49, 96, 200})
        hipLaunchKernelGGL(k_tri, dim3((nd * (nd + 1) / 2 + 255) / 256), dim3(256), 0, 0, nd, dm);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    (void)hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
    printf("reciprocal mismatches %llu, quotient mismatches %llu of %d; tri_decode mismatches %llu\n", hm[0], hm[1], N, hm[2]);
    if (hm[0] || hm[1] || hm[2]) { printf("DIV CHECK FAILED\n"); return 1; }
    printf("DIV CHECK PASSED\n");
    return 0;
}
