// Solver check: device one-wave Jacobi SVD / LDL^T (lm_solvers.h) against the host solvers (host_math.cpp) on random
// symmetric positive semi-definite systems of the sizes the LM loop produces (n = 6N), including rank-deficient ones.
// Built by tests/harness/build.sh; run by tests/test_gpu_cxx_harness.py.
#include "lm_solvers.h"
#include "host_math.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void k_solve(const double *A, const double *b, double *x, int n, int solver)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, ld = n + 1;
    // the LDS layout of k_lm_solve: V and G (n x (n + 1) each) + vectors; the system itself stays in global memory
    double *V = lds, *g = V + n * ld, *xx = g + n, *tmp = xx + n;
    int *order = (int *)(tmp + n);
    double *G = tmp + 2 * n;
    for (int i = lane; i < n; i += 64) g[i] = b[i];
    __syncthreads();
    if (solver == 1)
    {
        for (int i = lane; i < n * n; i += 64) V[i] = A[i];
        __syncthreads();
        mbavo::ldlt_solve(V, g, xx, tmp, order, n, lane);
    }
    else
    {
        for (int i = lane; i < n * n; i += 64) G[(i / n) * ld + i % n] = A[i];
        __syncthreads();
        mbavo::svd_solve(G, V, g, xx, tmp, n, ld, lane);
    }
    for (int i = lane; i < n; i += 64) x[i] = -xx[i];
}

// the workgroup-parallel two-sided Jacobi (lm_solvers.h: eig_solve) in the LDS layout of k_lm_solve
__global__ __launch_bounds__(mbavo::kEigT) void k_solve_eig(const double *A, const double *b, double *x, int n, long long *cycles, int reps)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    double *bufs = lds, *g = bufs + mbavo::eig_lds_doubles(n), *xx = g + n, *tmp = xx + n, *Hs = tmp + n;
    int *flags = (int *)(Hs + n * n);
    long long best = 0x7fffffffffffffffll;
    for (int rep = 0; rep < reps; ++rep)
    { // the same solve `reps` times: the shortest is the one at full clock
        for (int i = tid; i < n; i += mbavo::kEigT) g[i] = b[i];
        for (int i = tid; i < n * n; i += mbavo::kEigT) Hs[i] = A[i];
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
        mbavo::eig_solve(bufs, Hs, g, xx, tmp, flags, n, tid);
        const long long t1 = __builtin_amdgcn_s_memtime();
        best = t1 - t0 < best ? t1 - t0 : best;
        __syncthreads();
    }
    for (int i = tid; i < n; i += mbavo::kEigT) x[i] = -xx[i];
    if (tid == 0 && cycles) { cycles[0] = best; cycles[1] = flags[3]; }
}

// LDL^T in registers with double-double refinement (lm_solvers.h: spd_solve_regs_impl<NN, true>), one wave
template <int NN>
__global__ __launch_bounds__(64) void k_solve_refined(const double *A, const double *b, double *x, int *ok, double max_ratio, double max_refined,
                                                      long long *cycles)
{
    __shared__ double As[NN * NN], bs[NN], xs[NN];
    const int lane = threadIdx.x;
    for (int i = lane; i < NN * NN; i += 64) As[i] = A[i];
    for (int i = lane; i < NN; i += 64) bs[i] = b[i];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const bool good = mbavo::spd_solve_regs_impl<NN, true>(As, bs, xs, lane, max_ratio, max_refined);
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    for (int i = lane; i < NN; i += 64) x[i] = xs[i];
    if (lane == 0) { *ok = good ? 1 : 0; *cycles = t1 - t0; }
}

// the workgroup form for the sizes the register form does not take (lm_solvers.h: spd_solve_coop), system in LDS or global memory
template <int T>
__global__ __launch_bounds__(T) void k_solve_coop(const double *A, const double *b, double *x, int *ok, int n, int a_in_lds, double max_ratio,
                                                  double max_refined, long long *cycles)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    double *F = lds, *As = F + n * (n + 1), *bs = As + n * n, *xs = bs + n, *rd = xs + n;
    int *flag = (int *)(rd + n);
    for (int i = tid; i < n * n; i += T) As[i] = A[i];
    for (int i = tid; i < n; i += T) bs[i] = b[i];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const bool good = mbavo::spd_solve_coop<T>(F, a_in_lds ? As : A, bs, xs, rd, flag, n, tid, max_ratio, max_refined);
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    for (int i = tid; i < n; i += T) x[i] = xs[i];
    if (tid == 0) { *ok = good ? 1 : 0; *cycles = t1 - t0; }
}

// Systems with an exactly known solution: J integer with every column = one common column times 2^e + small integers (nearly
// dependent columns: cond(J^T J) ~ 2^2e), x small integers, A = J^T J and b = A x exact in double (all integers below 2^53).
// The plain LDL^T lands within ~cond eps of x, the refined one must land within 1e-12 |x|_inf -- or say that it did not converge.
template <int NN, int COOP = 0> // COOP: 0 = register form (n = NN), 64 / 256 = workgroup form with that many threads (64: system in global memory)
static int refined_case(int e, double max_ratio, const char *label)
{
    const int n = NN, m = n + 8;
    std::vector<double> J((size_t)m * n), A((size_t)n * n), b(n), xt(n), xd(n), u(m);
    const int ur = n > 24 ? 7 : 15, xr = n > 24 ? 3 : 7; // smaller integers for the larger systems: every sum below stays under 2^53
    for (int r = 0; r < m; ++r) u[r] = (double)(rand() % ur - ur / 2);
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < n; ++c) J[(size_t)r * n + c] = ldexp(u[r], e) + (double)(rand() % ur - ur / 2);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c)
        {
            double a = 0; // n <= 24: |J| < 2^(e + 3.1), 32 terms: below 2^(2 e + 11.2); n <= 60: |J| < 2^(e + 2), 68 terms: below 2^(2 e + 10.1)
            for (int k = 0; k < m; ++k) a += J[(size_t)k * n + r] * J[(size_t)k * n + c];
            A[(size_t)c * n + r] = a;
        }
    for (int c = 0; c < n; ++c) xt[c] = (double)(rand() % xr - xr / 2);
    for (int r = 0; r < n; ++r) { double a = 0; for (int c = 0; c < n; ++c) a += A[(size_t)c * n + r] * xt[c]; b[r] = a; } // below 2^(2 e + 17.4) / 2^(2 e + 16): exact for e <= 17
    double *dA, *db, *dx; int *dok, ok = 0; long long *dcyc, cyc = 0;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&db, n * 8); hipMalloc(&dx, n * 8); hipMalloc(&dok, 4); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    int bad = 0;
    for (int refined = 0; refined < 2; ++refined)
    {
        // plain: every ratio admitted, no refinement; refined: nothing admitted unrefined
        if constexpr (COOP == 0)
            hipLaunchKernelGGL(k_solve_refined<NN>, dim3(1), dim3(64), 0, 0, dA, db, dx, dok, refined ? max_ratio : 1e300, refined ? 1e13 : 0.0, dcyc);
        else
        {
            const size_t lds = ((size_t)n * (n + 1) + (size_t)n * n + 3 * n) * 8 + 16;
            hipFuncSetAttribute((const void *)k_solve_coop<COOP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_solve_coop<COOP>, dim3(1), dim3(COOP), lds, 0, dA, db, dx, dok, n, COOP == 256 ? 1 : 0, refined ? max_ratio : 1e300,
                               refined ? 1e13 : 0.0, dcyc);
        }
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(xd.data(), dx, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
        double err = 0, nrm = 0;
        for (int i = 0; i < n; ++i) { err = fmax(err, fabs(xd[i] - xt[i])); nrm = fmax(nrm, fabs(xt[i])); }
        // a refined result that claims convergence must be within 1e-12; one that does not claim it sends the caller to Jacobi
        const bool pass = e == hipSuccess && (refined ? (!ok || err <= 1e-12 * nrm) : true);
        printf("n=%3d %s %s  max|x_dev - x_exact| / |x| = %.3e  accepted %d  %lld cycles %s\n", n,
               COOP == 0 ? (refined ? "LDLT refined" : "LDLT plain  ") : COOP == 64 ? (refined ? "LDLT refined, 1 wave  " : "LDLT plain, 1 wave    ")
                                                                                     : (refined ? "LDLT refined, 4 waves " : "LDLT plain, 4 waves   "), label,
               err / nrm, ok, cyc, pass ? "ok" : "FAIL");
        bad += !pass;
    }
    hipFree(dA); hipFree(db); hipFree(dx); hipFree(dok); hipFree(dcyc);
    return bad;
}

int main()
{
    int bad = 0;
    srand(3);
    for (int e = 2; e <= 17; e += 5)
    { // cond ~ 2^2e times the small part's own: 1e2 .. 1e11
        char label[64];
        snprintf(label, sizeof(label), "near-dependent 2^%-2d", e);
        bad += refined_case<12>(e, 0.0, label);
        bad += refined_case<18>(e, 0.0, label);
        bad += refined_case<24>(e, 0.0, label);
        bad += refined_case<30, 256>(e, 0.0, label);
        bad += refined_case<36, 64>(e, 0.0, label);
        bad += refined_case<48, 256>(e, 0.0, label);
        bad += refined_case<60, 64>(e, 0.0, label);
    }
    for (int N = 2; N <= 16; N += (N < 8 ? 1 : 4)) // up to the reference's max_num_ctrl_knots = 16 (n = 96)
        for (int solver = 0; solver < 2; ++solver)
            for (int rankdef = 0; rankdef < 3; ++rankdef)
            { // 2: full rank with graded columns (cond ~1e9: the cubic spline's damped normal equations look like this)
                if (solver == 1 && rankdef) continue; // LDLT is for full-rank systems
                const int n = 6 * N, m = rankdef == 1 ? n - 6 : n + 8;
                std::vector<double> J((size_t)m * n), A((size_t)n * n, 0.0), b(n), xh(n), xd(n);
                for (auto &v : J) v = rand() / (double)RAND_MAX - 0.5;
                if (rankdef == 2)
                    for (int r = 0; r < m; ++r)
                        for (int c = 0; c < n; ++c) J[(size_t)r * n + c] *= pow(10.0, -4.5 * ((c * 7) % n) / (double)(n - 1));
                for (int r = 0; r < n; ++r)
                    for (int c = 0; c < n; ++c)
                    {
                        double a = 0;
                        for (int k = 0; k < m; ++k) a += J[(size_t)k * n + r] * J[(size_t)k * n + c];
                        A[(size_t)c * n + r] = a;
                    }
                for (int r = 0; r < n; ++r) { double a = 0; for (int k = 0; k < m; ++k) a += J[(size_t)k * n + r]; b[r] = a; } // in range(A)
                mbavo::solve_normal_equation_host(A.data(), b.data(), n, solver, xh.data());
                double *dA, *db, *dx;
                hipMalloc(&dA, A.size() * 8); hipMalloc(&db, n * 8); hipMalloc(&dx, n * 8);
                hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
                hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
                const size_t lds = ((size_t)2 * n * (n + 1) + 6 * n) * 8 + n * 4;
                hipFuncSetAttribute((const void *)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(k_solve, dim3(1), dim3(64), lds, 0, dA, db, dx, n, solver);
                hipError_t e = hipDeviceSynchronize();
                hipMemcpy(xd.data(), dx, n * 8, hipMemcpyDeviceToHost);
                double err = 0, nrm = 0;
                for (int i = 0; i < n; ++i) { err = fmax(err, fabs(xd[i] - xh[i])); nrm = fmax(nrm, fabs(xh[i])); }
                const bool ok = e == hipSuccess && err <= 1e-8 * fmax(nrm, 1.0);
                printf("n=%3d %s %s  max|x_dev - x_host| = %.3e (|x| %.3e) %s\n", n, solver ? "LDLT" : "SVD ", rankdef == 1 ? "rank-deficient" : rankdef ? "graded        " : "full rank     ",
                       err, nrm, ok ? "ok" : "FAIL");
                bad += !ok;
                if (solver == 0 && n <= mbavo::kEigMaxN)
                { // the same system through the workgroup-parallel eigenvalue Jacobi
                    long long *dcyc, cyc[2] = {0, 0};
                    hipMalloc(&dcyc, 16);
                    const size_t lds2 = (mbavo::eig_lds_doubles(n) + 3 * n + n * n) * 8 + (4 + 2 * n) * 4; // + four flag words and the sorted order
                    hipFuncSetAttribute((const void *)k_solve_eig, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                    hipMemset(dx, 0, n * 8);
                    hipLaunchKernelGGL(k_solve_eig, dim3(1), dim3(mbavo::kEigT), lds2, 0, dA, db, dx, n, dcyc, 40);
                    e = hipDeviceSynchronize();
                    hipMemcpy(xd.data(), dx, n * 8, hipMemcpyDeviceToHost);
                    hipMemcpy(cyc, dcyc, 16, hipMemcpyDeviceToHost);
                    err = 0;
                    for (int i = 0; i < n; ++i) err = fmax(err, fabs(xd[i] - xh[i]));
                    const bool ok2 = e == hipSuccess && err <= 1e-8 * fmax(nrm, 1.0);
                    printf("n=%3d EIG  %s  max|x_dev - x_host| = %.3e (|x| %.3e) %s  [%lld sweeps, %lld cycles (best of 40): %.0f per round]\n", n,
                           rankdef == 1 ? "rank-deficient" : rankdef ? "graded        " : "full rank     ", err, nrm, ok2 ? "ok" : "FAIL", cyc[1], cyc[0], (double)cyc[0] / (double)(cyc[1] * (n - 1)));
                    bad += !ok2;
#if defined(MBAVO_EIG_STAMPS)
                    long long st[8];
                    hipMemcpyFromSymbol(st, HIP_SYMBOL(mbavo::g_eig_stamps), sizeof(st));
                    printf("        stamps: sort %lld gather %lld factor %lld scale %lld LtL %lld sweeps %lld solve %lld\n", st[1] - st[0], st[2] - st[1],
                           st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6]);
#endif
                    hipFree(dcyc);
                }
                hipFree(dA); hipFree(db); hipFree(dx);
            }
    printf("%s\n", bad ? "SOLVER CHECK FAILED" : "SOLVER CHECK PASSED");
    return bad;
}
