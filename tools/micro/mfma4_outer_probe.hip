// Check of the planned 25-entry outer product on v_mfma_f64_4x4x4_4b_f64: rows of 4 pixels, 7 groups of 4 entries
// (25 padded to 28), instruction n / block d computes group n x group (n + d) % 7.  Lane = 16 * pixel + 4 * d + e.
// MODE 0: A loaded explicitly; MODE 1: A of block 0 broadcast (cbsz = 2, abid = 0).  Also prints what cbsz / abid do.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int MODE>
__global__ void k(const double *rows /*[4][28]*/, double *out /*[7][64]*/)
{
    const int lane = threadIdx.x, kq = lane >> 4, d = (lane >> 2) & 3, e = lane & 3;
    double W[7], A[7], acc[7];
    for (int m = 0; m < 7; ++m) { W[m] = rows[kq * 28 + 4 * ((m + d) % 7) + e]; A[m] = rows[kq * 28 + 4 * m + e]; acc[m] = 0.0; }
#pragma unroll
    for (int n = 0; n < 7; ++n)
        acc[n] = MODE == 0 ? __builtin_amdgcn_mfma_f64_4x4x4f64(A[n], W[n], acc[n], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f64_4x4x4f64(W[n], W[n], acc[n], 2, 0, 0);
    for (int n = 0; n < 7; ++n) out[n * 64 + lane] = acc[n];
}
template <int CBSZ, int ABID>
__global__ void k1(int la, double *out)
{
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(lane == la ? 1.0 : 0.0, 1.0, 0.0, CBSZ, ABID, 0);
}
static int check(const double *h, const double *o)
{
    double worst = 0; int bad = 0;
    for (int r = 0; r < 28; ++r)
        for (int c = r; c < 28; ++c)
        {
            double want = 0;
            for (int p = 0; p < 4; ++p) want += h[p * 28 + r] * h[p * 28 + c];
            const int I = r / 4, J = c / 4, dd = (J - I + 7) % 7;
            int n, d, i, j;
            if (dd <= 3) { n = I; d = dd; i = r % 4; j = c % 4; } else { n = J; d = 7 - dd; i = c % 4; j = r % 4; }
            const double err = fabs(o[n * 64 + 16 * i + 4 * d + j] - want);
            if (err > 1e-12) ++bad;
            if (err > worst) worst = err;
        }
    printf("bad %d worst %.3g\n", bad, worst);
    return bad;
}
int main()
{
    double h[4 * 28], o[7 * 64], *dr, *dout;
    for (int i = 0; i < 4 * 28; ++i) h[i] = sin(1.0 + 0.37 * i) + 0.01 * i;
    (void)hipMalloc(&dr, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
    (void)hipMemcpy(dr, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dr, dout);
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("explicit A: "); check(h, o);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dr, dout);
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("cbsz=2 abid=0: "); check(h, o);
    // what the broadcast controls do: A one-hot at lane la, B all ones -> D lanes that become 1
    for (int la : {0, 1, 4, 5, 8, 12, 16, 20})
    {
        double r[64];
        auto show = [&](const char *tag) {
            (void)hipMemcpy(r, dout, 64 * 8, hipMemcpyDeviceToHost);
            printf("A one-hot lane %2d %-14s ->", la, tag);
            for (int l = 0; l < 64; ++l) if (r[l] != 0.0) printf(" %d", l);
            printf("\n");
        };
        hipLaunchKernelGGL((k1<0, 0>), dim3(1), dim3(64), 0, 0, la, dout); show("cbsz0");
        hipLaunchKernelGGL((k1<1, 0>), dim3(1), dim3(64), 0, 0, la, dout); show("cbsz1 abid0");
        hipLaunchKernelGGL((k1<1, 1>), dim3(1), dim3(64), 0, 0, la, dout); show("cbsz1 abid1");
        hipLaunchKernelGGL((k1<2, 0>), dim3(1), dim3(64), 0, 0, la, dout); show("cbsz2 abid0");
        hipLaunchKernelGGL((k1<2, 1>), dim3(1), dim3(64), 0, 0, la, dout); show("cbsz2 abid1");
    }
    return 0;
}
