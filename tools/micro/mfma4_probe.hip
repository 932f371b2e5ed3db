// Probe the operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 with one-hot inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int la, int lb, double *out)
{
    const int lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
int main()
{
    double *d, h[64];
    hipMalloc(&d, 64 * 8);
    // for every (la, lb) record which output lanes are non-zero
    int hit_count = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb)
        {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, la, lb, d);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            for (int o = 0; o < 64; ++o)
                if (h[o] != 0.0)
                {
                    if (la < 20 && lb < 20 || hit_count < 0) printf("A lane %2d x B lane %2d -> D lane %2d (%g)\n", la, lb, o, h[o]);
                    ++hit_count;
                }
        }
    printf("total hits %d\n", hit_count);
    // summary: for A lane la, which B lanes pair with it, and where the result goes
    for (int la = 0; la < 64; la += 1)
    {
        printf("A%2d:", la);
        for (int lb = 0; lb < 64; ++lb)
        {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, la, lb, d);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            for (int o = 0; o < 64; ++o) if (h[o] != 0.0) printf(" B%d->D%d", lb, o);
        }
        printf("\n");
    }
    return 0;
}
