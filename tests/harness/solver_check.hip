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

int main()
{
    int bad = 0;
    srand(3);
    for (int N = 2; N <= 16; N += (N < 8 ? 1 : 4)) // up to the reference's max_num_ctrl_knots = 16 (n = 96)
        for (int solver = 0; solver < 2; ++solver)
            for (int rankdef = 0; rankdef < 2; ++rankdef)
            {
                if (solver == 1 && rankdef) continue; // LDLT is for full-rank systems
                const int n = 6 * N, m = rankdef ? n - 6 : n + 8;
                std::vector<double> J((size_t)m * n), A((size_t)n * n, 0.0), b(n), xh(n), xd(n);
                for (auto &v : J) v = rand() / (double)RAND_MAX - 0.5;
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
                printf("n=%3d %s %s  max|x_dev - x_host| = %.3e (|x| %.3e) %s\n", n, solver ? "LDLT" : "SVD ", rankdef ? "rank-deficient" : "full rank     ",
                       err, nrm, ok ? "ok" : "FAIL");
                bad += !ok;
                hipFree(dA); hipFree(db); hipFree(dx);
            }
    printf("%s\n", bad ? "SOLVER CHECK FAILED" : "SOLVER CHECK PASSED");
    return bad;
}
