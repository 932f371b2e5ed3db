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

// The solver entry points are force-inlined: as real (called) device functions they faulted inside k_lm_solve in round 1 --
// a miscompile of the CALLER by this toolchain (profiles/r02_lm_solve_fault.txt; the reproducer build variants were removed
// in round 4, they are in the history up to commit cfcf142).
#define MBAVO_SOLVER_FN __device__ __forceinline__
#define MBAVO_SWEEPS_FN MBAVO_SOLVER_FN
#define MBAVO_SVD_FN MBAVO_SOLVER_FN
#define MBAVO_LDLT_FN MBAVO_SOLVER_FN

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


        // ---- x = pinv(A) b for SYMMETRIC A (the damped normal equations: positive semi-definite) by the two-sided (eigenvalue)
        // Jacobi method, for a whole WORKGROUP of kEigT threads (n even, n <= kEigMaxN).  The one-sided sweeps above spend a
        // round on three dot products per column pair, a dependent chain of two square roots and two divisions and two LDS
        // passes behind barriers (~2 400 cycles per round on one wave whatever n is); here
        //  * a rotation's parameters come from three matrix entries (no dot products);
        //  * the n/2 disjoint rotations of a round are applied as one congruence A <- J^T A J, V <- V J whose (n/2)^2
        //    independent 2 x 2 blocks are the work items: thread (i, j) turns block (i, j) of A and of V and writes the
        //    rotated blocks into the OTHER buffer -- one barrier per round;
        //  * the tournament is run Brent-Luk fashion: the pairs of a round are always the neighbours (2i, 2i + 1), and the
        //    write side moves every row / column to the slot where it meets its next opponent (one fixed permutation for
        //    all rounds), so that every read is an aligned 16-byte pair;
        //  * the DIRECTION of a rotation needs no more than single precision to annihilate an entry to working accuracy over
        //    the sweeps, only its normalisation has to be exact: (c, s) = (u, +-o) / sqrt(u^2 + o^2) with o = 2 a_pq,
        //    u = |a_qq - a_pp| + hypot(a_qq - a_pp, o) -- the hypot from the hardware's approximate square root, the
        //    normalisation from v_rsq_f64 and two Newton steps: two transcendental instructions per rotation, ~100 cycles.
        // What a round costs was measured piece by piece (tools/micro/eig_round.hip, one workgroup alone on a CU, 2.39 GHz):
        // a barrier 40 cycles, a dependent LDS read 72, a dependent v_fma_f64 4.9, a transcendental f64 instruction ~21 -- and
        // the LDS pipe: 128 bytes a cycle for the whole CU whatever the exec mask says (a wave's ds_read_b128 is 8 cycles of
        // it), so four b128 reads and four b64 writes per thread take 300 / 440 / 690 cycles per round with 64 / 256 / 512
        // threads.  The first forms of this solver (every thread reading both diagonal blocks and deriving both rotations, or
        // separate threads for A and V each reading a diagonal block) sat at ~1 100 cycles per round for that reason.  Hence:
        // one thread turns the A block AND the V block (no second set of diagonal reads), it reads three entries of ITS row
        // pair's diagonal block (16 + 8 bytes, issued first: the block reads ride behind the parameter chain), derives that
        // rotation and takes the column pair's from a lane of its own wave that has derived it (ds_bpermute: 4 bytes a lane);
        // waves without live items skip the round; the leading dimension is padded so that the half-wave groups of a b128
        // read fall on complementary banks.
        // Convergence as in the one-sided form: a pair is left alone when |a_pq| <= eps sqrt(a_pp a_qq) (the RELATIVE test:
        // on the graded systems of a cubic spline, cond 1e9, it keeps the small eigenvalues accurate to ~1e-12 where a test
        // against the largest diagonal entry gives 1e-7); the sweeps end when a sweep rotated nothing, or nothing above
        // 1e-7 (quadratic convergence: what is left after such a sweep is ~1e-14 of sqrt(a_pp a_qq), a second-order 1e-28 on the
        // eigenvalues; numpy prototype on the oracle's systems: the same solution errors as with 1e-9, one sweep fewer on some).
        // Threshold of the pseudo-inverse as solve_svd: n eps max|lambda|.  A is PSD, so |lambda_j| are its singular values.
        // Layout: bufs = A0 | A1 | V0 | V1, each n columns of eig_ld(n) doubles; A0 holds A on entry (destroyed).
#ifndef MBAVO_EIG_ABL
#define MBAVO_EIG_ABL 0
#endif
        struct JacobiRot { double c, s; };
        __device__ __forceinline__ JacobiRot jacobi_rot(double app, double aqq, double apq, bool &big)
        {
#pragma clang fp contract(off)
            const double lim2 = fabs(app * aqq), a2 = apq * apq;
            const bool rot = a2 > (DBL_EPSILON * DBL_EPSILON) * lim2;
            big = a2 > 1e-14 * lim2;
            const double d = aqq - app, o = apq + apq, o2 = o * o;
            const double h = __builtin_amdgcn_sqrt(__builtin_fma(d, d, o2)); // ~ hypot(d, o)
            const double u = fabs(d) + h, so = __builtin_copysign(o, __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, o) ^ __builtin_bit_cast(unsigned long long, d)));
            const double w = __builtin_fma(u, u, o2);
            double y = __builtin_amdgcn_rsq(w);
#pragma unroll
            for (int it = 0; it < 2; ++it)
            { // y <- y + y/2 (1 - w y^2)
                const double e = __builtin_fma(-(w * y), y, 1.0);
                y = __builtin_fma(0.5 * y, e, y);
            }
            JacobiRot r; // tan = sign(d) o / u
            r.c = rot ? u * y : 1.0;
            r.s = rot ? so * y : 0.0;
            return r;
        }
        // (u, w) <- (c u - s w, s u + c w)
        __device__ __forceinline__ void jacobi_apply(const JacobiRot &r, double &u, double &w)
        {
#pragma clang fp contract(off)
            const double nu = __builtin_fma(r.c, u, -(r.s * w)), nw = __builtin_fma(r.s, u, r.c * w);
            u = nu;
            w = nw;
        }
        struct __attribute__((aligned(16))) D2 { double x, y; };
        constexpr int kEigMaxN = 48, kEigT = 256;
        // leading dimension: even (16-byte pairs), and = n/2 (mod 16) where that is even, so that the groups of n/2 lanes that
        // read 16-byte pairs of consecutive rows from columns 2 ld doubles apart tile the 64 banks
        __host__ __device__ inline int eig_ld(int n)
        {
            const int h = n / 2;
            if (h & 1) return n;
            int ld = n;
            while ((ld & 15) != (h & 15)) ld += 2;
            return ld;
        }
        __host__ __device__ inline size_t eig_lds_doubles(int n) { return (size_t)4 * n * eig_ld(n); }
        __device__ __forceinline__ double shfl_f64(double v, int src)
        {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
            const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)(unsigned)u);
            const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)(unsigned)(u >> 32));
            return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        }
        // up to MAXI work items per thread: (n/2)^2 <= kEigT * MAXI.  Returns the buffer the result is in.
        template <int MAXI>
        __device__ __forceinline__ int eig_sweeps(double *bufs, int *flags, int n, int ld, int tid)
        {
            const int half = n / 2, hh = half * half, sz = n * ld; // A0 | V0 prepared, flags[0..2] zero (visible after the barrier below)
            // slot -> slot of the next round: top row 2i, bottom row 2i + 1; top_0 stays, the others go round
            auto next_slot = [half](int sl) {
                const int i = sl >> 1;
                if ((sl & 1) == 0)
                {
                    if (i == 0) return 0;
                    return i == half - 1 ? 2 * i + 1 : 2 * (i + 1);
                }
                return i == 0 ? (half > 1 ? 2 : 1) : 2 * (i - 1) + 1;
            };
            const int wave_base = tid & ~63;
            // The items of a thread never change: their offsets are computed once.  Item (i, j) rotates the block (I, J) =
            // (min, max) of A -- the two threads either side of the diagonal compute the SAME block from the same entries, one
            // writes it as it stands and the other transposed, so A stays symmetric bit for bit and no thread branches on its
            // side of the diagonal -- and rows (2i, 2i + 1) x columns (2j, 2j + 1) of V.  i runs fastest over the lanes:
            // neighbouring lanes read neighbouring rows, and the first n/2 lanes of every wave own all the row pairs.
            int o_d[MAXI], o_b[MAXI], o_v[MAXI], w00[MAXI], w01[MAXI], w10[MAXI], w11[MAXI], wv0[MAXI], wv1[MAXI], src[MAXI], kind[MAXI];
#pragma unroll
            for (int q = 0; q < MAXI; ++q)
            {
                const int it = tid + q * kEigT;
                const bool live = it < hh;
                const int j = live ? it / half : 0, i = it % half; // i: valid for every lane (its rotation may be asked for)
                const bool swap = i > j;
                const int I2 = 2 * (swap ? j : i), J2 = 2 * (swap ? i : j);
                o_d[q] = 2 * i * ld + 2 * i; o_b[q] = J2 * ld + I2; o_v[q] = 2 * j * ld + 2 * i;
                const int pI0 = next_slot(I2), pI1 = next_slot(I2 + 1), pJ0 = next_slot(J2), pJ1 = next_slot(J2 + 1);
                // element (row I + a, column J + b) of the rotated block goes to (pI_a, pJ_b), transposed for the lower side
                w00[q] = swap ? pI0 * ld + pJ0 : pJ0 * ld + pI0; w01[q] = swap ? pI1 * ld + pJ0 : pJ0 * ld + pI1;
                w10[q] = swap ? pI0 * ld + pJ1 : pJ1 * ld + pI0; w11[q] = swap ? pI1 * ld + pJ1 : pJ1 * ld + pI1;
                wv0[q] = next_slot(2 * j) * ld + 2 * i; wv1[q] = next_slot(2 * j + 1) * ld + 2 * i;
                src[q] = ((j - wave_base) % half + half) % half; // the lane of this wave whose FIRST item has row pair j
                // bit 0: live, bit 1: the thread's own rotation is the COLUMN pair's (lower side), bit 2: diagonal block,
                // bit 3: reports the state of its row pair
                kind[q] = (live ? 1 : 0) | (swap ? 2 : 0) | (live && i == j ? 4 : 0) | (live && j == 0 ? 8 : 0);
            }
            // a wave takes part in item q as a whole (the shuffle needs the lanes that own the row pairs) or not at all
            bool wave_live[MAXI];
#pragma unroll
            for (int q = 0; q < MAXI; ++q) wave_live[q] = wave_base + q * kEigT < hh;
            // One round, from buffer X into buffer Y.  Every address of a round is a register: the two directions (0 -> 1 and
            // 1 -> 0) keep their own sets and the rounds are unrolled in pairs (with the buffer chosen by a parity inside the loop
            // every access paid a shift and an addition: 16 vector + 8 scalar instructions of ~115 per round, and a lone wave
            // issues one every ~7 cycles).  Rejected variants (tools/micro/eig_round.hip, cycles per round at n = 24 against 990):
            // both roles of a rotation by shuffle instead of the selects 1 068; the column pair's rotation derived locally
            // instead of shuffled 1 109.
            struct Dir
            {
                const double *d[MAXI], *b[MAXI], *v[MAXI];
                double *w00[MAXI], *w01[MAXI], *w10[MAXI], *w11[MAXI], *wv0[MAXI], *wv1[MAXI];
            };
            Dir fw, bw; // reads buffer 0 / writes buffer 1, and the other way round
#pragma unroll
            for (int q = 0; q < MAXI; ++q)
            {
                const double *A0 = bufs, *A1 = bufs + sz, *V0 = bufs + 2 * sz, *V1 = bufs + 3 * sz;
                fw.d[q] = A0 + o_d[q]; fw.b[q] = A0 + o_b[q]; fw.v[q] = V0 + o_v[q];
                bw.d[q] = A1 + o_d[q]; bw.b[q] = A1 + o_b[q]; bw.v[q] = V1 + o_v[q];
                double *W1 = bufs + sz, *W0 = bufs, *X1 = bufs + 3 * sz, *X0 = bufs + 2 * sz;
                fw.w00[q] = W1 + w00[q]; fw.w01[q] = W1 + w01[q]; fw.w10[q] = W1 + w10[q]; fw.w11[q] = W1 + w11[q];
                fw.wv0[q] = X1 + wv0[q]; fw.wv1[q] = X1 + wv1[q];
                bw.w00[q] = W0 + w00[q]; bw.w01[q] = W0 + w01[q]; bw.w10[q] = W0 + w10[q]; bw.w11[q] = W0 + w11[q];
                bw.wv0[q] = X0 + wv0[q]; bw.wv1[q] = X0 + wv1[q];
            }
            auto round = [&](const Dir &a, int *flag) {
                if (!wave_live[0]) return;
                D2 d0[MAXI], b0[MAXI], b1[MAXI], v0[MAXI], v1[MAXI];
                double dqq[MAXI];
#pragma unroll
                for (int q = 0; q < MAXI; ++q)
                { // (a_pp, a_qp) and a_qq of the row pair first: the parameter chain starts as soon as they are in
                    if (!wave_live[q]) continue;
                    d0[q] = *(const D2 *)a.d[q];
                    dqq[q] = a.d[q][ld + 1];
                }
#pragma unroll
                for (int q = 0; q < MAXI; ++q)
                {
                    if (!wave_live[q]) continue;
                    b0[q] = *(const D2 *)a.b[q]; b1[q] = *(const D2 *)(a.b[q] + ld);
                    v0[q] = *(const D2 *)a.v[q]; v1[q] = *(const D2 *)(a.v[q] + ld);
                }
                JacobiRot first;
#pragma unroll
                for (int q = 0; q < MAXI; ++q)
                {
                    if (!wave_live[q]) continue;
                    bool big;
#if MBAVO_EIG_ABL == 2 // ablation (tools/micro/eig_round.hip): no parameter chain
                    JacobiRot ri; ri.c = d0[q].x; ri.s = dqq[q] + d0[q].y; big = true;
#else
                    const JacobiRot ri = jacobi_rot(d0[q].x, dqq[q], d0[q].y, big);
#endif
                    if (q == 0) first = ri;
#if MBAVO_EIG_ABL != 3 // ablation 3: no convergence flag
                    if ((kind[q] & 8) && big) *flag = 1; // every writer stores the same word
#endif
                    JacobiRot rj;
#if MBAVO_EIG_ABL == 1 // ablation: no shuffle
                    rj = first;
#else
                    rj.c = shfl_f64(first.c, src[q]);
                    rj.s = shfl_f64(first.s, src[q]);
#endif
                    const bool lower = kind[q] & 2;
                    JacobiRot rI, rJ;
                    rI.c = lower ? rj.c : ri.c; rI.s = lower ? rj.s : ri.s;
                    rJ.c = lower ? ri.c : rj.c; rJ.s = lower ? ri.s : rj.s;
                    // rows by J_I, then columns by J_J
                    jacobi_apply(rI, b0[q].x, b0[q].y);
                    jacobi_apply(rI, b1[q].x, b1[q].y);
                    jacobi_apply(rJ, b0[q].x, b1[q].x);
                    jacobi_apply(rJ, b0[q].y, b1[q].y);
                    if (kind[q] & 4) b0[q].y = b1[q].x; // the diagonal block stays exactly symmetric
                    jacobi_apply(rj, v0[q].x, v1[q].x);
                    jacobi_apply(rj, v0[q].y, v1[q].y);
#if MBAVO_EIG_ABL == 4 // ablation: one write instead of six
                    if (kind[q] & 1) *a.w00[q] = b0[q].x + b0[q].y + b1[q].x + b1[q].y + v0[q].x + v0[q].y + v1[q].x + v1[q].y;
#else
                    if (kind[q] & 1)
                    {
                        *a.w00[q] = b0[q].x; *a.w01[q] = b0[q].y;
                        *a.w10[q] = b1[q].x; *a.w11[q] = b1[q].y;
                        *(D2 *)a.wv0[q] = v0[q]; // the rows of V stay where they are
                        *(D2 *)a.wv1[q] = v1[q];
                    }
#endif
                }
            };
            __syncthreads();
            int cur = 0; // the buffer the next round reads
            for (int sweep = 0; sweep < 30; ++sweep)
            {
                int *flag = flags + sweep % 3;
                if (tid == 0) flags[(sweep + 1) % 3] = 0; // last read two sweeps ago, many barriers back
                int r = 0;
                for (; r + 1 < n - 1; r += 2)
                {
                    round(fw, flag);
                    __syncthreads();
                    round(bw, flag);
                    __syncthreads();
                }
                if (r < n - 1)
                { // n - 1 is odd: one more round, and the two directions change names for the next sweep
                    round(fw, flag);
                    __syncthreads();
                    const Dir t = fw; fw = bw; bw = t;
                    cur ^= 1;
                }
                if (tid == 0) flags[3] = sweep + 1; // sweeps taken (read by the solver check)
                if (*flag == 0) break; // nothing rotated, or nothing above 1e-7: converged
            }
            return cur;
        }

        // A = L D L^T -> L (see eig_solve): lower triangle of A in place, the strict upper triangle zeroed.  false: a pivot
        // was not positive (A is left half-factored).
        template <int MAXE>
        __device__ __forceinline__ bool eig_factor(double *A1, int n, int ld, int tid)
        {
            int er[MAXE], ec[MAXE];
            bool on_e[MAXE];
#pragma unroll
            for (int q = 0; q < MAXE; ++q)
            { // entry e of the lower triangle, row-major: e = r (r + 1) / 2 + c
                const int e = tid + q * kEigT;
                int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                r += (r + 1) * (r + 2) / 2 <= e ? 1 : 0;
                r -= r * (r + 1) / 2 > e ? 1 : 0;
                on_e[q] = r < n;
                er[q] = on_e[q] ? r : 0;
                ec[q] = on_e[q] ? e - r * (r + 1) / 2 : 0;
            }
            for (int k = 0; k < n; ++k)
            {
                const double d = A1[k * ld + k];
                double own[MAXE], rk[MAXE], ck[MAXE];
#pragma unroll
                for (int q = 0; q < MAXE; ++q)
                { // idle entries (column <= k, or past the matrix) read valid words and are not written
                    own[q] = A1[ec[q] * ld + er[q]];
                    rk[q] = A1[k * ld + er[q]];
                    ck[q] = A1[k * ld + ec[q]];
                }
                if (!(d > 0.0)) return false; // the same word in every thread
                double rd = __builtin_amdgcn_rcp(d);
                rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
                rd = __builtin_fma(__builtin_fma(-d, rd, 1.0), rd, rd);
#pragma unroll
                for (int q = 0; q < MAXE; ++q)
                    if (on_e[q] && ec[q] > k) A1[ec[q] * ld + er[q]] = own[q] - rk[q] * (ck[q] * rd);
                __syncthreads();
            }
            // l_rc = (d_c l_rc) / sqrt(d_c) below the diagonal, sqrt(d_c) on it, zero above
            double lv[MAXE];
#pragma unroll
            for (int q = 0; q < MAXE; ++q)
            {
                const double dc = A1[ec[q] * ld + ec[q]], v = A1[ec[q] * ld + er[q]];
                double y = __builtin_amdgcn_rsq(dc);
#pragma unroll
                for (int it = 0; it < 2; ++it) y = __builtin_fma(0.5 * y, __builtin_fma(-(dc * y), y, 1.0), y);
                lv[q] = er[q] == ec[q] ? dc * y : v * y;
            }
            __syncthreads(); // every diagonal entry has been read
#pragma unroll
            for (int q = 0; q < MAXE; ++q)
                if (on_e[q])
                {
                    A1[ec[q] * ld + er[q]] = lv[q];
                    if (er[q] != ec[q]) A1[er[q] * ld + ec[q]] = 0.0;
                }
            __syncthreads();
            return true;
        }

        // Preconditioning (Veselic / Hari): for positive definite A the sweeps run on L^T L instead of A = L L^T (L the Cholesky
        // factor of A with its rows / columns sorted by falling diagonal: the stand-in for pivoting), accumulating G = L J_1 J_2 ...
        // in V's place.  L^T L is one step of the LR iteration closer to diagonal: 5 sweeps where A itself takes 9 on the
        // cubic spline's systems (cond 1e9), to the same accuracy.  With L^T L = W diag(lambda) W^T the columns of G = L W are
        // sqrt(lambda_j) u_j, u_j the eigenvectors of A and lambda_j its eigenvalues: x = sum_j g_j (g_j . b) / lambda_j^2.
        // A pivot that is not positive (semi-definite or indefinite A) sends the system down the plain path: sweeps on A, V = I.
        // Hs: the system, dense n x n column-major in LDS (kept); bufs: eig_lds_doubles(n) doubles of scratch; iwork: 4 + 2 n ints.
#if defined(MBAVO_EIG_STAMPS) // phase stamps of the LAST solve (tests/harness/solver_check.hip, -DMBAVO_EIG_STAMPS)
        __device__ long long g_eig_stamps[8];
#define MBAVO_EIG_STAMP(k) do { if (tid == 0) g_eig_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MBAVO_EIG_STAMP(k) do { } while (0)
#endif
        // Returns the number of sweeps taken, + 256 when the system went down the preconditioned path.
        __device__ __forceinline__ int eig_solve(double *bufs, const double *Hs, const double *b, double *x, double *tmp, int *iwork, int n, int tid)
        {
            constexpr int T = kEigT;
            const int ld = eig_ld(n), sz = n * ld, nn = n * n;
            int *flags = iwork, *ord = iwork + 4;
            double *A0 = bufs, *A1 = bufs + sz, *V0 = bufs + 2 * sz;
            if (tid < 3) flags[tid] = 0;
            MBAVO_EIG_STAMP(0);
            // sorted position of every index: falling diagonal, ties in index order -- n^2 comparisons over all threads, counted
            // in LDS (one thread per index walking the diagonal took 3 000 cycles)
            int *rank = ord + n;
            for (int i = tid; i < n; i += T) rank[i] = 0;
            __syncthreads();
            for (int e = tid; e < nn; e += T)
            {
                const int i = e / n, k = e % n;
                const double di = Hs[i * n + i], dk = Hs[k * n + k];
                if (dk > di || (dk == di && k < i)) atomicAdd(rank + i, 1);
            }
            __syncthreads();
            for (int i = tid; i < n; i += T) ord[rank[i]] = i;
            __syncthreads();
            MBAVO_EIG_STAMP(1); // sorted
            for (int e = tid; e < nn; e += T) A1[(e / n) * ld + e % n] = Hs[ord[e / n] * n + ord[e % n]];
            __syncthreads();
            MBAVO_EIG_STAMP(2); // gathered
            // A1 = L D L^T in place (lower triangle; column k ends up holding d_k l_rk), right-looking with ONE barrier per
            // column: a thread updates its entries of the trailing triangle straight from the unscaled column and the pivot,
            //     a_rc <- a_rc - a_rk a_ck / d_k        (r >= c > k),
            // so nobody waits for a scaled column to be written first (the textbook form took three barriers a column,
            // ~1 000 cycles each; a single wave walking the columns alone took 2 800 for a 24 x 24 system's 276 dependent
            // LDS reads).  The entries of a thread are fixed: decoded once; idle entries read entry (0, 0) instead of
            // branching round their loads.  Every thread reads the pivot: the verdict (all pivots positive) is uniform.
            const bool spd = n * (n + 1) / 2 <= 2 * T ? eig_factor<2>(A1, n, ld, tid) : eig_factor<(kEigMaxN * (kEigMaxN + 1) / 2 + T - 1) / T>(A1, n, ld, tid);
            MBAVO_EIG_STAMP(4); // factored and scaled
            if (spd)
            { // A0 <- L^T L (upper triangle computed, mirrored: symmetric bit for bit), V0 <- L.  The strict upper triangle of A1
              // is zero, so the sums run over whole columns: six pairs of entries in flight
                for (int e = tid; e < nn; e += T)
                {
                    const int c = e / n, r = e % n;
                    V0[c * ld + r] = A1[c * ld + r];
                    if (r <= c)
                    {
                        double acc = 0.0;
                        for (int m0 = 0; m0 < n; m0 += 6)
                        {
                            D2 u[3], w[3];
#pragma unroll
                            for (int q = 0; q < 3; ++q) { u[q] = *(const D2 *)(A1 + r * ld + m0 + 2 * q); w[q] = *(const D2 *)(A1 + c * ld + m0 + 2 * q); }
#pragma unroll
                            for (int q = 0; q < 3; ++q) acc += u[q].x * w[q].x + u[q].y * w[q].y;
                        }
                        A0[c * ld + r] = acc;
                        A0[r * ld + c] = acc;
                    }
                }
            }
            else
            { // the plain path: sweeps on A itself, V = I
                for (int e = tid; e < nn; e += T)
                {
                    A0[(e / n) * ld + e % n] = Hs[e];
                    V0[(e / n) * ld + e % n] = (e / n == e % n) ? 1.0 : 0.0;
                }
                for (int i = tid; i < n; i += T) ord[i] = i;
            }
            MBAVO_EIG_STAMP(5); // L^T L
            // one work item per thread wherever the blocks allow it
            const int cur = (n / 2) * (n / 2) <= T ? eig_sweeps<1>(bufs, flags, n, ld, tid)
                                                   : eig_sweeps<((kEigMaxN / 2) * (kEigMaxN / 2) + T - 1) / T>(bufs, flags, n, ld, tid);
            MBAVO_EIG_STAMP(6); // sweeps
            const double *Af = bufs + (cur ? sz : 0), *Vf = Af + 2 * sz;
            // coefficients (g_j . b) / lambda_j^p and x = sum_j g_j coef_j: four lanes per dot product, the quarters meet by
            // xor-shuffles (4 n <= 192 threads: whole waves take part)
            const int jj = tid >> 2, part = tid & 3, q4 = n / 4 + (n % 4 ? 1 : 0), lo = part * q4, hi = lo + q4 < n ? lo + q4 : n;
            double lmax = 0.0;
            for (int j0 = 0; j0 < n; j0 += 6)
            {
                double dj[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) dj[u] = Af[(j0 + u) * ld + j0 + u];
#pragma unroll
                for (int u = 0; u < 6; ++u) lmax = fmax(lmax, fabs(dj[u]));
            }
            const double thr = fmax((double)n * DBL_EPSILON * lmax, DBL_MIN);
            if (tid < ((4 * n + 63) & ~63))
            {
                const int j = jj < n ? jj : n - 1;
                double dot = 0.0;
                for (int i = lo; i < hi; ++i) dot += Vf[j * ld + i] * b[ord[i]];
                dot += __shfl_xor(dot, 1, 64);
                dot += __shfl_xor(dot, 2, 64);
                const double lam = Af[j * ld + j];
                const bool keep = fabs(lam) >= thr && lam != 0.0;
                if (part == 0 && jj < n) tmp[j] = keep ? dot / (spd ? lam * lam : lam) : 0.0;
            }
            __syncthreads();
            if (tid < ((4 * n + 63) & ~63))
            {
                const int i = jj < n ? jj : n - 1;
                double acc = 0.0;
                for (int j = lo; j < hi; ++j) acc += Vf[j * ld + i] * tmp[j];
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                if (part == 0 && jj < n) x[ord[i]] = acc;
            }
            const int info = flags[3] + (spd ? 256 : 0);
            __syncthreads();
            MBAVO_EIG_STAMP(7); // solved
            return info;
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
        //
        // REFINE (the batched LM, where one ill-conditioned system of a batch would otherwise hold its whole launch for the
        // eigenvalue Jacobi's 70 us): a system whose pivot ratio lies between max_ratio and max_ratio_refined keeps the
        // factors and takes steps of iterative refinement with the residual b - A x accumulated in double-double (two-product by
        // FMA, two-sum): while cond(A) eps < 1 the iteration contracts by ~cond(A) eps per step onto the correctly rounded
        // solution of the system as given -- the centre of the ball of radius ~cond(A) eps |x| in which the reference's Jacobi
        // SVD (solve_normal_equation.h:10-35; every singular value above Eigen's threshold ~n eps sigma_max: full rank) lands.
        // Accepted when the error LEFT after a correction d_m is below 1e-13 |x|_inf: the iteration contracts by a fixed factor,
        // which the first step shows -- the first correction IS the error of the unrefined solution, so rho ~ |d_1| / |x| -- and
        // the error after step m is ~rho |d_m|; the test is |d_m| min(1, 10 rho) <= 1e-13 |x| (one step suffices up to
        // |d_1| <= 1e-7 |x|, i.e. cond ~1e9: the cubic spline's systems).  Anything else -- a non-positive pivot, a ratio above
        // max_ratio_refined, no convergence in kRefineSteps -- returns false and the caller takes the Jacobi solver.
        constexpr int kRefineSteps = 4;
        // 1 / d for a pivot: the runtime's IEEE division without its range scaling and special-case fix-up (v_rcp_f64, the same
        // Newton steps, the same final correction: the SAME correctly rounded bits for a normal d with a normal reciprocal --
        // pixel_math.h: reciprocal) -- 11 instructions less on each of the 24 dependent steps of the factorisation.  A pivot that
        // is zero, negative or subnormal fails the positivity / ratio test either way.
        __device__ __forceinline__ double pivot_reciprocal(double d)
        {
            double r = __builtin_amdgcn_rcp(d);
            double e = __builtin_fma(-d, r, 1.0);
            r = __builtin_fma(r, e, r);
            e = __builtin_fma(-d, r, 1.0);
            r = __builtin_fma(r, e, r);
            e = __builtin_fma(-d, r, 1.0);
            return __builtin_fma(e, r, r);
        }
        template <int NN, bool REFINE>
        __device__ __forceinline__ bool spd_solve_regs_impl(const double *A, const double *b, double *x, int lane, double max_ratio,
                                                            double max_ratio_refined)
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
                const double rd = pivot_reciprocal(d);
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
            const double rdi = pivot_reciprocal(diag);
            double xv = rhs * rdi;
#pragma unroll
            for (int j = NN - 1; j >= 1; --j)
            {
                const double xj = bcast_lane(xv, j);
                if (lane < j) xv -= (a[j] * rdi) * xj;
            }
            bool ok = pos && dmax <= max_ratio * dmin;
            if constexpr (REFINE)
            {
                if (!ok && pos && dmax <= max_ratio_refined * dmin)
                {
                    const double bi = b[i];
                    double rho = 1.0;
                    for (int step = 0; step < kRefineSteps && !ok; ++step)
                    {
                        // r_i = b_i - sum_j A_ij x_j as an unevaluated sum hi + lo
                        double hi = bi, lo = 0.0;
#pragma unroll
                        for (int j = 0; j < NN; ++j)
                        {
#pragma clang fp contract(off) // the error-free transformations below must not be fused
                            const double aij = A[j * NN + i], xj = bcast_lane(xv, j);
                            const double p = aij * xj, pe = fma(aij, xj, -p);      // aij xj = p + pe exactly
                            const double t = hi - p, bb = t - hi;                  // two-sum of hi and -p
                            const double err = (hi - (t - bb)) + (-p - bb);
                            hi = t;
                            lo += err - pe;
                        }
                        double r = hi + lo;
                        // L y = r, z = D^-1 y, L^T d = z with the factors in the registers
#pragma unroll
                        for (int k = 0; k < NN - 1; ++k)
                        {
                            const double yk = bcast_lane(r, k);
                            if (lane > k) r -= a[k] * yk;
                        }
                        double dv = r * rdi;
#pragma unroll
                        for (int j = NN - 1; j >= 1; --j)
                        {
                            const double dj = bcast_lane(dv, j);
                            if (lane < j) dv -= (a[j] * rdi) * dj;
                        }
                        xv += dv;
                        const double dn = wmax(lane < NN ? fabs(dv) : 0.0), xn = wmax(lane < NN ? fabs(xv) : 0.0);
                        // contraction per step: what the first correction shows, but never below the pivot ratio x eps -- |d1| / |x| can sit far
                        // under cond(A) eps when b lies in well-conditioned directions, and one step would then be accepted with the
                        // true contraction ~1e-3: from a ratio of ~1e10 on, a second correction is always taken
                        if (step == 0) rho = fmin(1.0, fmax(10.0 * dn / xn, (dmax / dmin) * DBL_EPSILON));
                        ok = dn * rho <= 1e-13 * xn; // (NaN compares false)
                    }
                }
            }
            if (lane < NN) x[lane] = xv;
            return ok;
        }
        template <int NN>
        __device__ __forceinline__ bool spd_solve_regs(const double *A, const double *b, double *x, int lane, double max_ratio)
        {
            return spd_solve_regs_impl<NN, false>(A, b, x, lane, max_ratio, 0.0);
        }

        // ---- the same stand-in for the systems the register form does not take (any even n <= 64: 5 .. 10 control knots),
        // by a whole workgroup of T threads.  Factorisation: unpivoted LDL^T, right-looking, the trailing matrix held in
        // REGISTERS in a TD x TD cyclic distribution (thread (ti, tj) owns the entries (ti + a TD, tj + b TD), lower triangle);
        // in step k the owners of column k store it into column k of `F` (LDS, n x (n + 1): it is final, d_k on the diagonal and
        // d_k L[i][k] below), ONE barrier, then every thread reads the pivot, its rows' and its columns' entries of that column
        // (1 + 2 MAXD LDS reads) and updates its MAXD^2 registers -- no read-modify-write of LDS, no index division, the column
        // of step k + 1 goes to another column of F while late threads still read column k.  The substitutions and the
        // double-double refinement by wave 0 with lane i on row i (y_k handed over by v_readlane with a uniform lane index), the
        // other waves waiting at the barrier.  `A`: the system as given (column-major n x n, LDS or global memory: read once for
        // the registers and once per refinement step, lanes on consecutive rows), `rd`: n doubles of LDS (reciprocal pivots).
        // Same acceptance rule and return value as spd_solve_regs_impl<.., true>; uniform over the workgroup.
        // First form (every step a read-modify-write pass over the trailing block in LDS): 100 000 cycles at n = 48 / 256 threads.
        __device__ __forceinline__ double bcast_lane_uniform(double v, int src) // src: the same in every lane (a loop counter)
        {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
            return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        }
        template <int T>
        __device__ __forceinline__ bool spd_solve_coop(double *F, const double *A, const double *b, double *x, double *rd, int *flag, int n,
                                                       int tid, double max_ratio, double max_ratio_refined)
        {
            constexpr int TD = T >= 256 ? 16 : 8, MAXD = 64 / TD;
            const int ld = n + 1, lane = tid & 63, ti = tid % TD, tj = (tid / TD) % TD;
            const bool grid = tid < TD * TD;
            double v[MAXD][MAXD];
#pragma unroll
            for (int bb = 0; bb < MAXD; ++bb)
#pragma unroll
                for (int a = 0; a < MAXD; ++a)
                {
                    const int i = ti + a * TD, j = tj + bb * TD;
                    v[a][bb] = (grid && i < n && j <= i) ? A[j * n + i] : 0.0;
                }
            double dmax = 0.0, dmin = DBL_MAX;
            bool pos = true;
#pragma unroll
            for (int kb = 0; kb < MAXD; ++kb) // the column block of step k, a compile-time index into the registers
                for (int k = kb * TD; k < (kb + 1) * TD && k < n && pos; ++k)
                {
                    if (grid && tj == k - kb * TD)
                    { // column k is final
#pragma unroll
                        for (int a = 0; a < MAXD; ++a)
                        {
                            const int i = ti + a * TD;
                            if (i >= k && i < n) F[k * ld + i] = v[a][kb];
                        }
                    }
                    __syncthreads();
                    const double d = F[k * ld + k]; // (every thread: the state below is uniform)
                    double ci[MAXD], cj[MAXD];
#pragma unroll
                    for (int a = 0; a < MAXD; ++a)
                    {
                        const int i = ti + a * TD, j = tj + a * TD;
                        ci[a] = (i > k && i < n) ? F[k * ld + i] : 0.0;
                        cj[a] = (j > k && j < n) ? F[k * ld + j] : 0.0;
                    }
                    pos = d > 0.0;
                    dmax = fmax(dmax, d);
                    dmin = fmin(dmin, d);
                    const double r = pivot_reciprocal(d);
                    if (tid == 0) rd[k] = r;
#pragma unroll
                    for (int bb = 0; bb < MAXD; ++bb)
                    {
                        const double lj = cj[bb] * r; // L[j][k] (zero outside the trailing block: nothing changes there)
#pragma unroll
                        for (int a = 0; a < MAXD; ++a) v[a][bb] -= ci[a] * lj;
                    }
                }
            __syncthreads();
            bool ok = false;
            if (pos && tid < 64)
            { // wave 0: F holds d_k on the diagonal and d_k L[i][k] below it
                const int i = lane < n ? lane : n - 1; // lanes past the matrix shadow the last row (results unused)
                const double rdi = rd[i], bi = b[i];
                // (the factor entries of kU steps are fetched together, ahead of the dependent chain that uses them: one LDS / memory
                // latency per kU steps instead of one per step -- 300 -> ~40 cycles per step)
                constexpr int kU = 8;
                auto solve = [&](double y) { // L D L^T w = y, row i of y in lane i; returns row i of w
                    for (int k0 = 0; k0 < n - 1; k0 += kU)
                    {
                        double l[kU];
#pragma unroll
                        for (int u = 0; u < kU; ++u)
                        {
                            const int k = k0 + u < n - 1 ? k0 + u : n - 2;
                            l[u] = F[k * ld + i] * rd[k];
                        }
#pragma unroll
                        for (int u = 0; u < kU; ++u)
                        {
                            const int k = k0 + u;
                            if (k < n - 1)
                            {
                                const double yk = bcast_lane_uniform(y, k);
                                if (lane > k && lane < n) y -= l[u] * yk;
                            }
                        }
                    }
                    double w = y * rdi;
                    for (int k0 = n - 1; k0 >= 1; k0 -= kU)
                    {
                        double l[kU];
#pragma unroll
                        for (int u = 0; u < kU; ++u)
                        {
                            const int k = k0 - u >= 1 ? k0 - u : 1;
                            l[u] = F[i * ld + k] * rdi; // d_i L[k][i] sits in column i, row k
                        }
#pragma unroll
                        for (int u = 0; u < kU; ++u)
                        {
                            const int k = k0 - u;
                            if (k >= 1)
                            {
                                const double wk = bcast_lane_uniform(w, k);
                                if (lane < k) w -= l[u] * wk;
                            }
                        }
                    }
                    return w;
                };
                double xv = solve(bi);
                ok = dmax <= max_ratio * dmin;
                double rho = 1.0;
                if (!ok && dmax <= max_ratio_refined * dmin)
                    for (int step = 0; step < kRefineSteps && !ok; ++step)
                    {
                        double hi = bi, lo = 0.0; // r_i = b_i - sum_j A_ij x_j as an unevaluated sum hi + lo
                        for (int j0 = 0; j0 < n; j0 += kU)
                        {
                            double arow[kU];
#pragma unroll
                            for (int u = 0; u < kU; ++u) arow[u] = A[(j0 + u < n ? j0 + u : n - 1) * n + i];
#pragma unroll
                            for (int u = 0; u < kU; ++u)
                            {
#pragma clang fp contract(off) // the error-free transformations below must not be fused
                                const int j = j0 + u;
                                if (j < n)
                                {
                                    const double aij = arow[u], xj = bcast_lane_uniform(xv, j);
                                    const double p = aij * xj, pe = fma(aij, xj, -p);
                                    const double t = hi - p, bb = t - hi;
                                    const double err = (hi - (t - bb)) + (-p - bb);
                                    hi = t;
                                    lo += err - pe;
                                }
                            }
                        }
                        const double dv = solve(hi + lo);
                        xv += dv;
                        const double dn = wmax(lane < n ? fabs(dv) : 0.0), xn = wmax(lane < n ? fabs(xv) : 0.0);
                        // contraction per step: what the first correction shows, but never below the pivot ratio x eps -- |d1| / |x| can sit far
                        // under cond(A) eps when b lies in well-conditioned directions, and one step would then be accepted with the
                        // true contraction ~1e-3: from a ratio of ~1e10 on, a second correction is always taken
                        if (step == 0) rho = fmin(1.0, fmax(10.0 * dn / xn, (dmax / dmin) * DBL_EPSILON));
                        ok = dn * rho <= 1e-13 * xn; // (NaN compares false)
                    }
                if (lane < n) x[lane] = xv;
                if (lane == 0) *flag = ok ? 1 : 0;
            }
            else if (tid == 0)
                *flag = 0;
            __syncthreads();
            ok = *flag != 0;
            __syncthreads(); // (`flag` may be part of the next solver's work area)
            return ok;
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
