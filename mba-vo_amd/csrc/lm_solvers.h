// lm_solvers.h -- the two normal-equation solvers of solve_normal_equation.h:10-35 for ONE WAVE (device code):
// minimum-norm solve by one-sided Jacobi SVD and pivoted LDL^T, the algorithms of host_math.cpp with the rows /
// column pairs spread over the 64 lanes.  All matrices live in LDS, column-major.  Callers: lm_batch.hip and the
// solver check in tests/harness/solver_check.hip.
#ifndef MBAVO_LM_SOLVERS_H
#define MBAVO_LM_SOLVERS_H

#include <cfloat>
#include <hip/hip_runtime.h>
// Several lanes per column pair (svd_sweeps<2>, <4>).
// All solver functions are force-inlined.  As real (called) device functions they fault inside k_lm_solve
// (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION) while passing the standalone check -- root cause (round 2, rocgdb +
// ISA, profiles/r02_lm_solve_fault.txt; reproducer tools/micro/lm_solve_calls.sh): a MISCOMPILE of the CALLER by this
// toolchain (ROCm 7.2.0, AMD clang 22.0.0git roc-7.2.0, -O2 / -O3; -O1 is correct).  To keep `lane` alive across the call
// the register allocator copies it into a callee-saved VGPR (v_mov_b32 v40, v0) and places that copy in the exit block of
// the preceding divergent loop (the H -> G copy) BEFORE the s_or_b64 that restores the exec mask: the loop leaves with
// exec == 0, so the copy writes no lane and the callee receives whatever v40 held.  The solvers themselves are not at
// fault (no lane of theirs touches another's data out of turn; the standalone kernel's block layout happens to be
// compiled correctly); inlined there is no call and no copy.
#ifndef MBAVO_SVD_MULTILANE
#define MBAVO_SVD_MULTILANE 1
#endif

// -DMBAVO_SOLVERS_NOINLINE builds the three solver entry points as real (called) device functions: the configuration
// that faulted in round 1; kept as a build variant for the reproducer (tools/micro/lm_solve_calls.sh).
#if defined(MBAVO_SOLVERS_LDS_PTR) // experiment: the matrices as LDS-typed pointers instead of generic ones
#define MBAVO_LDS __attribute__((address_space(3)))
#else
#define MBAVO_LDS
#endif
#if defined(MBAVO_SOLVERS_NOINLINE)
#define MBAVO_SOLVER_FN __device__ __noinline__
#else
#define MBAVO_SOLVER_FN __device__ __forceinline__
#endif
// finer switches of the reproducer: one function at a time as a real call
#if defined(MBAVO_NOINLINE_SWEEPS)
#define MBAVO_SWEEPS_FN __device__ __noinline__
#else
#define MBAVO_SWEEPS_FN MBAVO_SOLVER_FN
#endif
#if defined(MBAVO_NOINLINE_SVD)
#define MBAVO_SVD_FN __device__ __noinline__
#else
#define MBAVO_SVD_FN MBAVO_SOLVER_FN
#endif
#if defined(MBAVO_NOINLINE_LDLT)
#define MBAVO_LDLT_FN __device__ __noinline__
#else
#define MBAVO_LDLT_FN MBAVO_SOLVER_FN
#endif

// The solvers are written for ONE wave; between their steps the wave's LDS writes must be visible to its other lanes.
// In a one-wave workgroup (lm_batch.hip, the solver check) that is a workgroup barrier; a wave working alone inside a larger
// workgroup (the resident LM kernel of engine.hip) defines MBAVO_SOLVER_SYNC as a wave-level fence before including this.
#ifndef MBAVO_SOLVER_SYNC
#define MBAVO_SOLVER_SYNC() __syncthreads()
#endif

namespace mbavo
{
    namespace
    {
        __device__ __forceinline__ double wsum(double v)
        {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            return v;
        }
        __device__ __forceinline__ double wmax(double v)
        {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
            return v;
        }

        // x = pinv(A) b by one-sided (Hestenes) Jacobi, the algorithm of host_math.cpp:solve_svd with the rotations of
        // a sweep taken in round-robin (tournament) order: the n/2 column pairs of a round are disjoint, so ONE LANE
        // PER PAIR computes its three dot products and applies its rotation with plain sequential loops -- no
        // cross-lane reduction and one barrier per round instead of one per rotation (the row-cyclic order with
        // wave-wide reductions measured 2.2 us per rotation, 4.9 ms per 24 x 24 solve; this order 37x less).
        // G holds A on entry, column-major with leading dimension ld = n + 1 (odd in doubles: the lanes of a round
        // read different columns at the same row, a stride of n doubles would put them on 4 LDS banks); V the same.
        // One sweep structure for SUB lanes per column pair: the N six-row groups of a column are dealt out to the SUB
        // lanes of a pair, partial dot products meet by xor-shuffles inside the (adjacent) lane group.
        template <int SUB>
        MBAVO_SWEEPS_FN bool svd_sweeps(double *G, double *V, int n, int ld, int lane)
        {
            const double eps = DBL_EPSILON;
            const int half = n / 2, m1 = n - 1, N6 = n / 6; // n = 6N is even
            const int pair = lane / SUB, sub = lane % SUB;
            for (int sweep = 0; sweep < 60; ++sweep)
            {
                bool rotated = false;
                for (int r = 0; r < m1; ++r)
                {
                    // every lane takes part in the shuffles; lanes past the last pair work on pair 0's columns
                    // without writing (their results are discarded)
                    const bool live = pair < half;
                    const int pr = live ? pair : 0;
                    int p = pr == 0 ? m1 : (r + pr) % m1, q = pr == 0 ? r : (r - pr + m1) % m1;
                    if (p > q) { const int t = p; p = q; q = t; }
                    double *gp = G + p * ld, *gq = G + q * ld;
                    double a = 0, c = 0, d = 0;
                    for (int g6 = sub; g6 < N6; g6 += SUB)
                    { // six rows at a time, all loads issued before the (in-order) accumulation
                        const int i0 = 6 * g6;
                        double u[6], w[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) { u[j] = gp[i0 + j]; w[j] = gq[i0 + j]; }
#pragma unroll
                        for (int j = 0; j < 6; ++j) { a += u[j] * u[j]; c += w[j] * w[j]; d += u[j] * w[j]; }
                    }
#pragma unroll
                    for (int o = 1; o < SUB; o <<= 1)
                    {
                        a += __shfl_xor(a, o, 64);
                        c += __shfl_xor(c, o, 64);
                        d += __shfl_xor(d, o, 64);
                    }
                    if (live && !(d == 0.0 || fabs(d) <= eps * sqrt(a * c)))
                    {
                        rotated = true;
                        const double zeta = (c - a) / (2.0 * d);
                        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                        double *vp = V + p * ld, *vq = V + q * ld;
                        for (int g6 = sub; g6 < N6; g6 += SUB)
                        {
                            const int i0 = 6 * g6;
                            double u[6], w[6], y[6], z[6];
#pragma unroll
                            for (int j = 0; j < 6; ++j) { u[j] = gp[i0 + j]; w[j] = gq[i0 + j]; y[j] = vp[i0 + j]; z[j] = vq[i0 + j]; }
#pragma unroll
                            for (int j = 0; j < 6; ++j)
                            {
                                gp[i0 + j] = cs * u[j] - sn * w[j];
                                gq[i0 + j] = sn * u[j] + cs * w[j];
                                vp[i0 + j] = cs * y[j] - sn * z[j];
                                vq[i0 + j] = sn * y[j] + cs * z[j];
                            }
                        }
                    }
                    MBAVO_SOLVER_SYNC();
                }
                if (__ballot(rotated) == 0ull) return true;
            }
            return false;
        }

        MBAVO_SVD_FN void svd_solve(double *G, double *V, const double *b, double *x, double *tmp, int n, int ld, int lane)
        {
            for (int i = lane; i < n * n; i += 64) V[(i / n) * ld + i % n] = (i / n == i % n) ? 1.0 : 0.0;
            MBAVO_SOLVER_SYNC();
            const double eps = DBL_EPSILON;
            // lanes per column pair: as many as fit the wave and have a six-row group to work on
            const int half = n / 2, N6 = n / 6;
            if (MBAVO_SVD_MULTILANE && half * 4 <= 64 && N6 >= 4) svd_sweeps<4>(G, V, n, ld, lane);
            else if (MBAVO_SVD_MULTILANE && half * 2 <= 64 && N6 >= 2) svd_sweeps<2>(G, V, n, ld, lane);
            else svd_sweeps<1>(G, V, n, ld, lane);
            // squared singular values, one column per lane
            double smax2 = 0.0;
            for (int j = lane; j < n; j += 64)
            {
                double a = 0;
                for (int i = 0; i < n; ++i) a += G[j * ld + i] * G[j * ld + i];
                tmp[j] = a;
                smax2 = fmax(smax2, a);
            }
            smax2 = wmax(smax2);
            const double thr = fmax((double)(n > 1 ? n : 1) * eps * sqrt(smax2), DBL_MIN);
            for (int j = lane; j < n; j += 64)
            {
                const double s2 = tmp[j];
                double dot = 0.0;
                if (sqrt(s2) >= thr && s2 != 0.0)
                {
                    for (int i = 0; i < n; ++i) dot += G[j * ld + i] * b[i];
                    dot /= s2;
                }
                tmp[j] = dot;
            }
            MBAVO_SOLVER_SYNC();
            for (int i = lane; i < n; i += 64)
            {
                double acc = 0.0;
                for (int j = 0; j < n; ++j)
                    if (tmp[j] != 0.0) acc += V[j * ld + i] * tmp[j];
                x[i] = acc;
            }
            MBAVO_SOLVER_SYNC();
        }


        // ---- small SPD systems in registers (n = NN <= 24, one wave): x = A^-1 b by LDL^T WITHOUT pivoting, lane i holding row i of
        // the (symmetric, fully updated) trailing matrix in NN registers.  Every index is a compile-time constant: the pivot row is
        // broadcast with v_readlane (constant lane), no LDS traffic and no barrier inside -- ~NN^2 / 2 x 3 instructions against
        // ~2 000 cycles per elimination step of ldlt_solve above (shuffle pivot search, row / column swaps and three LDS passes
        // with barriers: 27 us for a 12 x 12 system inside the resident LM kernel, ~1.5 us with this form).
        // Unpivoted LDL^T is backward stable for positive definite A; `ok` = every pivot positive and max / min pivot <= max_ratio
        // (the guard of host_math.cpp:solve_spd_fast; its pivots come in diagonal-pivoting order, so the two guards can disagree
        // next to the threshold -- both paths compute the same x to rounding x cond(A)).  A: column-major n x n in LDS (symmetric).
        __device__ __forceinline__ double bcast_lane(double v, int src) // src: compile-time constant after unrolling
        {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
            return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        }
        template <int NN>
        __device__ __forceinline__ bool spd_solve_regs(const double *A, const double *b, double *x, int lane, double max_ratio)
        {
            const int i = lane < NN ? lane : NN - 1; // lanes past the matrix shadow the last row (results unused)
            double a[NN];
#pragma unroll
            for (int j = 0; j < NN; ++j) a[j] = A[j * NN + i];
            double rhs = b[i];
            double dmax = 0.0, dmin = DBL_MAX;
            bool pos = true;
#pragma unroll
            for (int k = 0; k < NN; ++k)
            {
                const double d = bcast_lane(a[k], k);
                pos = pos && d > 0.0;
                dmax = fmax(dmax, d);
                dmin = fmin(dmin, d);
                const double rd = 1.0 / d;
                const double l = a[k] * rd; // L[i][k] for the lanes i > k
#pragma unroll
                for (int j = k + 1; j < NN; ++j)
                {
                    const double rkj = bcast_lane(a[j], k); // A(k)[k][j] = d * L[j][k]
                    if (lane > k) a[j] -= l * rkj;
                }
                // forward substitution rides along: y_k is final once rows 0 .. k-1 were eliminated
                const double yk = bcast_lane(rhs, k);
                if (lane > k) { rhs -= l * yk; a[k] = l; }
            }
            // z = D^-1 y; then x = L^-T z from the last unknown up: lane k holds d_k L[j][k] in a[j], j > k (its final row)
            double diag = 1.0; // d_i sits in a[i]: a dynamic register index, resolved through selects
#pragma unroll
            for (int j = 0; j < NN; ++j) diag = lane == j ? a[j] : diag;
            const double rdi = 1.0 / diag;
            double xv = rhs * rdi;
#pragma unroll
            for (int j = NN - 1; j >= 1; --j)
            {
                const double xj = bcast_lane(xv, j);
                if (lane < j) xv -= (a[j] * rdi) * xj;
            }
            if (lane < NN) x[lane] = xv;
            return pos && dmax <= max_ratio * dmin;
        }

        // x = A^-1 b by LDL^T with diagonal pivoting (host_math.cpp:solve_ldlt); M holds A on entry
        MBAVO_LDLT_FN void ldlt_solve(double *M, const double *b, double *x, double *y, int *order, int n, int lane)
        {
            for (int i = lane; i < n; i += 64) order[i] = i;
            MBAVO_SOLVER_SYNC();
            for (int k = 0; k < n; ++k)
            {
                // pivot: the largest |diagonal| of the trailing block, the first one on ties
                double best = -1.0;
                int piv = 0x7fffffff;
                for (int i = k + lane; i < n; i += 64)
                {
                    const double v = fabs(M[i * n + i]);
                    if (v > best) { best = v; piv = i; }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1)
                {
                    const double ob = __shfl_xor(best, o, 64);
                    const int op = __shfl_xor(piv, o, 64);
                    if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
                }
                if (piv == 0x7fffffff) piv = k; // every remaining diagonal entry is NaN: the host scan keeps k (no swap)
                // the host scan keeps k unless a later entry is strictly larger: identical to first-maximum
                if (piv != k)
                {
                    for (int c = lane; c < n; c += 64) { const double t = M[c * n + k]; M[c * n + k] = M[c * n + piv]; M[c * n + piv] = t; }
                    MBAVO_SOLVER_SYNC();
                    for (int r = lane; r < n; r += 64) { const double t = M[k * n + r]; M[k * n + r] = M[piv * n + r]; M[piv * n + r] = t; }
                    if (lane == 0) { const int t = order[k]; order[k] = order[piv]; order[piv] = t; }
                    MBAVO_SOLVER_SYNC();
                }
                const double d = M[k * n + k];
                if (d == 0.0) continue;
                for (int i = k + 1 + lane; i < n; i += 64) M[k * n + i] /= d; // column k of L
                MBAVO_SOLVER_SYNC();
                const int m = n - k - 1;
                for (int idx = lane; idx < m * m; idx += 64)
                { // lower triangle of the trailing block in place (each entry reads itself and column k only) ...
                    const int i = k + 1 + idx / m, j = k + 1 + idx % m;
                    if (i >= j) M[j * n + i] -= M[k * n + i] * (M[k * n + j] * d);
                }
                MBAVO_SOLVER_SYNC();
                for (int idx = lane; idx < m * m; idx += 64)
                { // ... then mirrored, so that later pivots see a full symmetric block
                    const int i = k + 1 + idx / m, j = k + 1 + idx % m;
                    if (i > j) M[i * n + j] = M[j * n + i];
                }
                MBAVO_SOLVER_SYNC();
            }
            for (int i = lane; i < n; i += 64) y[i] = b[order[i]];
            MBAVO_SOLVER_SYNC();
            for (int c = 0; c < n; ++c)
            {
                const double yc = y[c];
                for (int r = c + 1 + lane; r < n; r += 64) y[r] -= M[c * n + r] * yc;
                MBAVO_SOLVER_SYNC();
            }
            for (int i = lane; i < n; i += 64) y[i] = fabs(M[i * n + i]) > DBL_MIN ? y[i] / M[i * n + i] : 0.0;
            MBAVO_SOLVER_SYNC();
            for (int c = n - 1; c >= 0; --c)
            {
                double part = 0.0;
                for (int r = c + 1 + lane; r < n; r += 64) part += M[c * n + r] * y[r];
                part = wsum(part);
                if (lane == 0) y[c] -= part;
                MBAVO_SOLVER_SYNC();
            }
            for (int i = lane; i < n; i += 64) x[order[i]] = y[i];
            MBAVO_SOLVER_SYNC();
        }
    } // namespace
} // namespace mbavo

#endif
