// What a round of the workgroup-parallel Jacobi (lm_solvers.h: eig_sweeps) is made of: cycles (s_memtime) per iteration of
// loops that add one ingredient at a time, for one workgroup of NT threads alone on a CU.
//   hipcc --offload-arch=gfx950 -O3 -I mba-vo_amd/csrc -I include tools/micro/eig_round.hip -o tools/micro/eig_round && tools/micro/eig_round
#include "lm_solvers.h"
#include <cstdio>
#include <vector>

using namespace mbavo;

template <int VAR>
__global__ void k_round(double *out, long long *cycles, int iters, int n)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, nn = n * n;
    for (int i = tid; i < 4 * nn; i += blockDim.x) lds[i] = 1.0 + 1e-3 * (i % 97) + (i / n == i % n ? 5.0 : 0.0);
    __syncthreads();
    const int half = n / 2, i = tid % half, j = (tid / half) % half;
    const int o_d = 2 * i * n + 2 * i, o_b = 2 * j * n + 2 * i, src = j % 64;
    double acc = 0.0, x = 1.0 + tid * 1e-6;
    int cur = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
    {
        const double *As = lds + (cur ? nn : 0);
        double *Ad = lds + (cur ? 0 : nn);
        if (VAR == 0)
        { // barrier only
        }
        else if (VAR == 1)
        { // dependent fma chain: 32 per iteration
#pragma unroll
            for (int k = 0; k < 32; ++k) x = __builtin_fma(x, 0.999999, 1e-9);
        }
        else if (VAR == 2)
        { // rcp -> sqrt -> rsq chain, 4 of each
#pragma unroll
            for (int k = 0; k < 4; ++k) x = __builtin_amdgcn_rsq(__builtin_amdgcn_sqrt(__builtin_amdgcn_rcp(x) + 1.0) + 1.0) + 1.0;
        }
        else if (VAR == 3)
        { // 4 x b128 read -> 4 x b64 write (other buffer)
            const D2 a = *(const D2 *)(As + o_d), b = *(const D2 *)(As + o_d + n), c = *(const D2 *)(As + o_b), d = *(const D2 *)(As + o_b + n);
            Ad[o_b] = a.x + c.x; Ad[o_b + 1] = a.y + c.y; Ad[o_b + n] = b.x + d.x; Ad[o_b + n + 1] = b.y + d.y;
        }
        else if (VAR == 4)
        { // reads -> one rotation -> apply to the block -> writes
            const D2 a = *(const D2 *)(As + o_d), b = *(const D2 *)(As + o_d + n);
            D2 c = *(const D2 *)(As + o_b), d = *(const D2 *)(As + o_b + n);
            bool big;
            const JacobiRot r = jacobi_rot(a.x, b.y, b.x, big);
            jacobi_apply(r, c.x, c.y);
            jacobi_apply(r, d.x, d.y);
            Ad[o_b] = c.x; Ad[o_b + 1] = c.y; Ad[o_b + n] = d.x; Ad[o_b + n + 1] = d.y;
        }
        else if (VAR == 5)
        { // + the second rotation by shuffle, rows then columns
            const D2 a = *(const D2 *)(As + o_d), b = *(const D2 *)(As + o_d + n);
            D2 c = *(const D2 *)(As + o_b), d = *(const D2 *)(As + o_b + n);
            bool big;
            const JacobiRot r = jacobi_rot(a.x, b.y, b.x, big);
            JacobiRot r2;
            r2.c = shfl_f64(r.c, src);
            r2.s = shfl_f64(r.s, src);
            jacobi_apply(r, c.x, c.y);
            jacobi_apply(r, d.x, d.y);
            jacobi_apply(r2, c.x, d.x);
            jacobi_apply(r2, c.y, d.y);
            Ad[o_b] = c.x; Ad[o_b + 1] = c.y; Ad[o_b + n] = d.x; Ad[o_b + n + 1] = d.y;
        }
        else if (VAR == 6)
        { // the rotation alone, dependent from iteration to iteration (no LDS)
            bool big;
            const JacobiRot r = jacobi_rot(x, x + 1.0, 0.25 + acc, big);
            acc = r.s;
        }
        else if (VAR == 7)
        { // one b128 read feeding the address of the next (LDS latency)
            const D2 a = *(const D2 *)(As + ((int)x & 15) * 2);
            x = a.x;
        }
        cur ^= 1;
        if (VAR != 1 && VAR != 2 && VAR != 6 && VAR != 7) __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = x + acc + lds[tid];
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

// eig_sweeps<1> itself on a fixed matrix: cycles per round (the ablations of MBAVO_EIG_ABL break convergence: 30 sweeps)
__global__ __launch_bounds__(kEigT) void k_sweeps(long long *cycles, int n)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, ld = eig_ld(n), sz = n * ld;
    int *flags = (int *)(lds + 4 * sz);
    long long best = 0x7fffffffffffffffll;
    int sweeps = 0;
    for (int rep = 0; rep < 10; ++rep)
    {
        for (int e = tid; e < n * n; e += kEigT)
        {
            const int c = e / n, r = e % n;
            lds[c * ld + r] = (r == c ? 10.0 + r : 0.0) + 1.0 / (1.0 + r + c) + 0.01 * ((r * 7 + c * 7) % 5);
            lds[2 * sz + c * ld + r] = r == c ? 1.0 : 0.0;
        }
        if (tid < 4) flags[tid] = 0;
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
        eig_sweeps<1>(lds, flags, n, ld, tid);
        const long long t1 = __builtin_amdgcn_s_memtime();
        best = t1 - t0 < best ? t1 - t0 : best;
        sweeps = flags[3];
        __syncthreads();
    }
    if (tid == 0) { cycles[0] = best; cycles[1] = sweeps; }
}

template <int VAR>
static void run(const char *what, int nt, int n, int iters = 2000)
{
    double *out;
    long long *cyc, h = 0;
    hipMalloc(&out, 8 * 1024);
    hipMalloc(&cyc, 8);
    const size_t lds = (size_t)4 * n * n * 8;
    hipFuncSetAttribute((const void *)k_round<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep)
    {
        hipLaunchKernelGGL(k_round<VAR>, dim3(1), dim3(nt), lds, 0, out, cyc, iters, n);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-78s NT=%3d n=%2d  %7.1f cycles per iteration\n", what, nt, n, (double)h / iters);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int n : {12, 24, 30})
    {
        long long *cyc, h[2] = {0, 0};
        hipMalloc(&cyc, 16);
        const size_t lds = eig_lds_doubles(n) * 8 + 64;
        hipLaunchKernelGGL(k_sweeps, dim3(1), dim3(kEigT), lds, 0, cyc, n);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("eig_sweeps<1> (MBAVO_EIG_ABL=%d) n=%2d: %lld sweeps, %lld cycles: %.0f per round\n", MBAVO_EIG_ABL, n, h[1], h[0], (double)h[0] / (h[1] * (n - 1)));
        hipFree(cyc);
    }
#if MBAVO_EIG_ABL != 0
    return 0;
#endif
    for (int nt : {64, 256, 512})
    {
        run<0>("barrier only", nt, 24);
        run<3>("4 x ds_read_b128 -> 4 x ds_write_b64 -> barrier", nt, 24);
        run<4>("reads -> rotation (rcp, sqrt, rcp, rsq + 2 Newton) -> 2 applies -> writes -> barrier", nt, 24);
        run<5>("reads -> rotation -> shuffle of (c, s) -> 4 applies -> writes -> barrier", nt, 24);
    }
    run<1>("32 dependent v_fma_f64", 64, 24);
    run<2>("4 x (v_rcp_f64 -> add -> v_sqrt_f64 -> add -> v_rsq_f64 -> add), dependent", 64, 24);
    run<6>("one rotation's parameters, dependent on the last", 64, 24);
    run<7>("dependent ds_read_b128", 64, 24);
    // s_memtime against the wall clock
    {
        double *out; long long *cyc, h = 0;
        hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_round<1>, dim3(1), dim3(64), 4 * 24 * 24 * 8, 0, out, cyc, 200000, 24);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_round<1>, dim3(1), dim3(64), 4 * 24 * 24 * 8, 0, out, cyc, 200000, 24);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("s_memtime: %lld ticks in %.3f ms = %.1f MHz\n", h, ms, h / (ms * 1e3));
    }
    return 0;
}
