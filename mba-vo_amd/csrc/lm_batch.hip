// lm_batch.hip -- device-side Levenberg-Marquardt over a batch of independent alignment problems
// (SURVEY.md 8f row 4): what BlurAwareDirectTracker::optimizePyramidLevel does for ONE problem with a host round trip
// per evaluation (ba_tracker/blur_aware_direct_tracker.cpp:590-924) runs here for B problems with no host
// involvement inside an iteration: the packed blocks never leave the device, the 6N x 6N systems are assembled,
// damped and solved by one wave per problem, the accept / reject decision, the LM radius, the non-monotonic
// step evaluator and the outlier statistics are per-problem device state.  With a host loop the solve of 512
// systems (10 ms on one core) dwarfs the 0.16 ms evaluation; here an iteration is eight launches.
//
//   merge                ba_tracker/merge_hessian_gradient_cost.cpp:39-86
//   solve                ba_tracker/solve_normal_equation.h:10-35 (one-sided Jacobi SVD / pivoted LDL^T, the same
//                        algorithms as host_math.cpp, rows spread over the lanes of a wave)
//   LM radius            ba_tracker/levenberg_marquardt_strategy.cpp:9-45
//   step evaluator       ba_tracker/trust_region_step_evaluator.cpp:45-126
//   loop / outliers      ba_tracker/blur_aware_direct_tracker.cpp:590-699,799-924 (see tracker.cpp for the quirks kept)
//
// Per iteration slot:  k_lm_solve -> cost-only pass on the candidates -> k_lm_decide -> H/g pass on the accepted.
// Problems that are finished, took an invalid step or rejected their step sit the passes out (ProblemDesc::active).
#include "../../include/mbavo.h"
#include "engine.h"
#include "pixel_math.h"
#include "se3_math.h"
// The one-wave solvers of lm_solvers.h (one-sided Jacobi SVD, pivoted LDL^T) synchronise their steps with MBAVO_SOLVER_SYNC: a
// wave-level fence here -- LDS executes a wave's operations in order, so that is all a wave working alone needs, in the one-wave
// workgroups (k_lm_solve<KD, 64>) and as wave 0 of the wide ones (k_lm_solve<KD, kEigT> with solver type 1).
#define MBAVO_SOLVER_SYNC()                                   \
    do                                                        \
    {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#include "lm_solvers.h"
#include "lm_state.h"
#include "host_math.h"
#include "tracker.h"
#include "pose_entries.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

namespace mbavo
{
    // Where the sums of a (problem, frame) slot come from: the frame blocks a finalize kernel wrote, or -- deferred
    // (engine.h: set_defer_finalize) -- the tile partials themselves, added in k_finalize_flat's order ((t0 + t2) + (t1 + t3),
    // absent tiles = +0) and scaled by the same residual scale: the same bits, two launches fewer per LM iteration.
    struct FinSrc
    {
        const double *fb;        // frame blocks (E doubles per slot: [cost | g | H packed])
        const double *partials;  // deferred: tile partials, `stride` doubles each: [valid | g, H sums | cost | spare]
        const int *tile_begin;   // deferred: slot bf owns tiles [tile_begin[bf], tile_begin[bf + 1])
        int stride, deferred;
        int f1;                  // every problem has ONE frame: slot == problem index, so the slot's tile range, start index and residual
                                 // scale can be fetched at kernel entry, beside the descriptor instead of behind it (round 5: the merge
                                 // was a chain descriptor -> tile range -> partials of three memory latencies)
        const double *inv;       // f1: the residual scales, [problem]
    };
    __device__ __forceinline__ double slot_sum_range(const FinSrc &fs, int t0, int n, int e)
    {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) t[i] = fs.partials[(size_t)(t0 + i) * fs.stride + e];
        return (t[0] + t[2]) + (t[1] + t[3]);
    }
    __device__ __forceinline__ double slot_sum(const FinSrc &fs, int bf, int e)
    {
        const int t0 = fs.tile_begin[bf];
        return slot_sum_range(fs, t0, fs.tile_begin[bf + 1] - t0, e);
    }
    template <int E>
    __device__ __forceinline__ double slot_cost(const FinSrc &fs, int bf)
    {
        return fs.deferred ? slot_sum(fs, bf, E) : fs.fb[(size_t)bf * E]; // (patch costs are scaled when they are computed)
    }

    // The host's view of the loop without a copy or an event in the stream: the last workgroup of a k_lm_solve launch stores
    // (slot + 1) << 32 | problems done into a pinned host word.
    // Ticket and done count are ONE 64-bit word (num_done[6..7]; low half: workgroups that have passed, over all launches of the
    // call; high half: problems done), bumped by one atomic -- the workgroup whose ticket completes (slot + 1) B sees every other
    // workgroup's contribution in the value the atomic returns: no fence, no second counter read.
    // (round 4) The completing workgroup also leaves the engine's range-status counter in pinned memory (host_word[4], as an int)
    // before the word: with every problem's final state stored to pinned memory by the workgroup that ended it (k_lm_solve), a
    // finished call needs no copy back and no blocking stream synchronisation (~20 us of a 64-pair call).
    __device__ __forceinline__ void slot_publish(int *num_done, unsigned long long *host_word, int slot, int B, bool done_now,
                                                 const int *status_src = nullptr)
    {
        if (!host_word) return;
        unsigned long long *ticket = reinterpret_cast<unsigned long long *>(num_done + 6);
        const unsigned long long old = atomicAdd(ticket, 1ull + (done_now ? 1ull << 32 : 0ull));
        if ((unsigned)(old & 0xffffffffull) + 1u != (unsigned)(slot + 1) * (unsigned)B) return;
        // The completing workgroup: every other workgroup made its stores (final states in pinned memory, the status counter)
        // visible with a system-scope fence BEFORE its relaxed ticket; this acquire fence after the ticket that read theirs
        // pairs with those fences (fence-to-fence synchronisation), so the release store of the word below publishes all of
        // them -- by the memory model, not only by how gfx9 drains a system fence.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const unsigned nd = (unsigned)(old >> 32) + (done_now ? 1u : 0u);
        if (status_src != nullptr)
            __hip_atomic_store(reinterpret_cast<int *>(host_word + 4), __hip_atomic_load(status_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_word, ((unsigned long long)(slot + 1) << 32) | nd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // The same for the decide launch of a slot, with the LOOK-AHEAD count (round 4): how many problems the NEXT solve launch will
    // find finished (already done, or its loop check `iter + 1 > max_it || abs_dec < min_dec` will end them).  When that is all B,
    // the host -- which has the next solve queued and reads this word before it enqueues that slot's three passes -- enqueues
    // none: no idle launches behind the last solve (4 x ~4.6 us per call before).  Ticket word: num_done[8..9], the low half
    // counts workgroups over all decide launches of the call, the high half is reset by the completing workgroup.
    __device__ __forceinline__ void decide_publish(int *num_done, unsigned long long *host_word2, int slot, int B, bool will_finish)
    {
        if (!host_word2) return;
        unsigned long long *ticket = reinterpret_cast<unsigned long long *>(num_done + 8);
        const unsigned long long old = atomicAdd(ticket, 1ull + (will_finish ? 1ull << 32 : 0ull));
        if ((unsigned)(old & 0xffffffffull) + 1u != (unsigned)(slot + 1) * (unsigned)B) return;
        const unsigned nf = (unsigned)(old >> 32) + (will_finish ? 1u : 0u);
        atomicAdd(ticket, 0ull - ((unsigned long long)nf << 32)); // the count starts over for the next slot's decide launch
        __hip_atomic_store(host_word2, ((unsigned long long)(slot + 1) << 32) | nf, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __device__ __forceinline__ bool lm_will_finish(const LmState &s, const LmOpts &o)
    {
        return s.done != 0 || s.iter + 1 > o.max_it || s.abs_dec < o.min_dec; // k_lm_solve's loop check, one launch ahead
    }

    // One workgroup of T threads per problem: finish the previous accepted step, loop control, damping, solve, model change,
    // candidate.  T = 64 (one wave: the one-sided Jacobi SVD / pivoted LDL^T of lm_solvers.h, any n up to 96) or T = kEigT
    // (solver 0 with n <= kEigMaxN: the workgroup-parallel eigenvalue Jacobi, eig_solve).  The scalar state of the loop is
    // computed redundantly by every thread (same inputs, same arithmetic); thread 0 stores it.
    template <int KD, int T>
    __global__ __launch_bounds__(T) void k_lm_solve(const ProblemDesc *__restrict__ descs, LmState *__restrict__ states, LmOpts o,
                                                    FinSrc fs, const int *__restrict__ start_idx,
                                                    double *__restrict__ Hst, double *__restrict__ gst,
                                                    double *__restrict__ cur_t, double *__restrict__ cur_R,
                                                    int *__restrict__ active, mbavo_trace_rec *__restrict__ trace,
                                                    int *__restrict__ num_done, unsigned long long *host_word, int slot, int B,
                                                    PoseEntry<KD> *pose_table, int *pose_status, LmState *h_final)
    {
        constexpr int M6 = 6 * KD, ND = M6 + 1, E = ND * (ND + 1) / 2;
        extern __shared__ __attribute__((aligned(16))) double lds[];
        const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
        const ProblemDesc &d = descs[b];
        // one-frame batches: the first slot's tile range, start index and scale do not wait for the descriptor
        int e_t0 = 0, e_t1 = 0, e_st0 = 0;
        double e_inv = 0.0;
        if (fs.f1 && fs.deferred)
        {
            e_t0 = fs.tile_begin[b]; e_t1 = fs.tile_begin[b + 1]; e_st0 = start_idx[b];
            e_inv = fs.inv[b];
        }
        const int N = d.N, n = 6 * N, F = d.F;
        LmState s = states[b]; // (slot 0: the initial state, uploaded by the host with the head of the arena)
        if (s.done)
        {
            if (tid == 0) slot_publish(num_done, host_word, slot, B, false, pose_status);
            return;
        }
        if (slot == 0)
        { // the current point starts at the caller's knots (there is no init launch: one kernel less in every call)
            double *Ct0 = cur_t + (size_t)b * 3 * o.max_N, *CR0 = cur_R + (size_t)b * 4 * o.max_N;
            for (int i = tid; i < 3 * N; i += T) Ct0[i] = d.knots_t[i];
            for (int i = tid; i < 4 * N; i += T) CR0[i] = d.knots_R[i];
            __syncthreads();
        }
#if defined(MBAVO_EIG_STAMPS) // development aid: where the kernel's time goes (block 0 prints at its end)
        const long long ts0 = __builtin_amdgcn_s_memtime();
        long long ts1 = 0, ts2 = 0, ts3 = 0;
#endif
        // LDS, T = 64: V and G (n x (n + 1) each) + vectors; the damped system H itself lives in global memory (it persists
        // across iterations anyway), which leaves room for the reference's maximum of 16 control knots (n = 96).
        // T = kEigT: the four n x eig_ld(n) areas of eig_solve first (16-byte aligned), then the vectors.
        // T = kEigT additionally keeps the system itself in LDS for the length of the kernel (merge, damping, model change:
        // ~25 dependent global round trips otherwise); global memory holds it between launches.
        constexpr bool eig = T == kEigT; // the host launches this form when every n <= kEigMaxN
        double *V = lds, *g = V + (eig ? eig_lds_doubles(n) : (size_t)n * (n + 1)), *x = g + n, *tmp = x + n;
        double *Hl = tmp + n;                                        // T = kEigT: n x n
        int *order = eig ? (int *)(Hl + n * n) : (int *)(tmp + n);   // T = kEigT: eig_solve's 4 + 2 n ints
        double *Hg = Hst + (size_t)b * o.max_n * o.max_n, *gg = gst + (size_t)b * o.max_n;
        double *H = eig ? Hl : Hg;
        double *Ct = cur_t + (size_t)b * 3 * o.max_N, *CR = cur_R + (size_t)b * 4 * o.max_N;
        mbavo_trace_rec *tr = trace ? trace + (size_t)b * o.trace_cap : nullptr;

        if (s.fresh)
        { // the H/g pass at the current point has completed: its cost is the evaluation-point cost
            // Frame 0's packed entries (1 .. E - 1: g, then the upper triangle of H) for this thread are FETCHED FIRST, together with the
            // costs, and scattered after the bookkeeping: one memory latency for all of it instead of a dependent chain cost ->
            // bookkeeping -> g -> H (the kernel is a chain of such round trips: 11 700 of its 37 000 cycles were this merge).
            constexpr int kEntries = E - 1, kPer = (kEntries + T - 1) / T;
            const bool early = fs.f1 && fs.deferred;
            const double inv = early ? e_inv : (fs.deferred ? (d.inv_ptr != nullptr ? *d.inv_ptr : d.inv_num_residuals) : 0.0);
            const int bf0 = early ? b : d.bf_base, st0 = early ? e_st0 : start_idx[bf0];
            double val[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q)
            {
                const int p = 1 + tid + q * T;
                val[q] = p <= kEntries ? (early ? slot_sum_range(fs, e_t0, e_t1 - e_t0, p) * inv
                                                : (fs.deferred ? slot_sum(fs, bf0, p) * inv : fs.fb[(size_t)bf0 * E + p])) : 0.0;
            }
            double cost = 0.0;
            if (early) cost = slot_sum_range(fs, e_t0, e_t1 - e_t0, E);
            else
                for (int f = 0; f < F; ++f) cost += slot_cost<E>(fs, d.bf_base + f);
            s.eval_cost = cost;
            if (s.pending_accept)
            { // handleSuccessfulStep (:896-903)
                lm_accepted(s, s.quality);
                tr_accepted(s, s.eval_cost, s.model, o.max_nonmono);
                trace_push(s, tr, o.trace_cap, tid, 0, 1, s.cand_cost, s.model, s.quality);
                ++s.n_accept;
            }
            else
            { // iteration 0 of the level (:590-606)
                s.initial_cost = cost;
                lm_reset(s);
                tr_reset(s, cost);
                trace_push(s, tr, o.trace_cap, tid, 0, 0, 0.0, 0.0, 0.0);
            }
            // merge_hessian_gradient_cost.cpp:39-86, frames in order
            for (int i = tid; i < n * n; i += T) H[i] = 0.0;
            for (int i = tid; i < n; i += T) g[i] = 0.0;
            __syncthreads();
            auto knot_row = [&](int j, int st) { return j < 3 * KD ? 3 * st + j : 3 * (N + st) + (j - 3 * KD); };
#pragma unroll
            for (int q = 0; q < kPer; ++q)
            { // frame 0, from the registers (distinct entries per thread: no two threads add into the same word)
                const int p = 1 + tid + q * T;
                if (p <= M6)
                    g[knot_row(p - 1, st0)] += val[q];
                else if (p <= kEntries)
                {
                    int r, c;
                    tri_decode(p - ND, M6, r, c);
                    const int R = knot_row(r, st0), C = knot_row(c, st0);
                    H[C * n + R] += val[q];
                    if (R != C) H[R * n + C] += val[q];
                }
            }
            __syncthreads();
            for (int f = 1; f < F; ++f)
            {
                const int bf = d.bf_base + f;
                const double *blk = fs.fb + (size_t)bf * E;
                const int st = start_idx[bf];
                for (int j = tid; j < M6; j += T) g[knot_row(j, st)] += fs.deferred ? slot_sum(fs, bf, 1 + j) * inv : blk[1 + j];
                for (int e = tid; e < M6 * (M6 + 1) / 2; e += T)
                {
                    int r, c;
                    tri_decode(e, M6, r, c);
                    const int R = knot_row(r, st), C = knot_row(c, st);
                    const double v = fs.deferred ? slot_sum(fs, bf, ND + e) * inv : blk[ND + e];
                    H[C * n + R] += v;
                    if (R != C) H[R * n + C] += v;
                }
                __syncthreads();
            }
            for (int i = tid; i < n; i += T) gg[i] = g[i];
            s.fresh = 0;
            s.pending_accept = 0;
        }
        else
        {
            for (int i = tid; i < n; i += T) g[i] = gg[i];
            if (eig)
                for (int i = tid; i < n * n; i += T) Hl[i] = Hg[i];
        }
        __syncthreads();

        // finalizeIterationAndCheckIfMinimizerCanContinue (:910-924)
        ++s.iter;
        if (s.iter > o.max_it || s.abs_dec < o.min_dec)
        {
            s.done = 1;
            --s.iter;
            if (tid == 0)
            {
                active[b] = 0;
                states[b] = s;
                if (h_final != nullptr)
                { // the final state straight into pinned host memory, on its way before this workgroup counts as passed
                    h_final[b] = s;
                    __threadfence_system();
                }
                atomicAdd(num_done, 1); // (what the stream-drain scheme and the final check read)
                slot_publish(num_done, host_word, slot, B, true, pose_status);
            }
            // leave the accepted point in the caller's knot buffers
            double *Wt = const_cast<double *>(d.knots_t), *WR = const_cast<double *>(d.knots_R);
            for (int i = tid; i < 3 * N; i += T) Wt[i] = Ct[i];
            for (int i = tid; i < 4 * N; i += T) WR[i] = CR[i];
            return;
        }

#if defined(MBAVO_EIG_STAMPS)
        ts1 = __builtin_amdgcn_s_memtime();
#endif
        // computeTrustRegionStep (:799-831): the damping is applied in place and accumulates over rejected steps
        const double iradius = 1. / s.radius;
        for (int i = tid; i < n; i += T)
        {
            const double v = H[i * n + i] + H[i * n + i] * iradius;
            H[i * n + i] = v;
        }
        __syncthreads();
        if (eig)
            for (int i = tid; i < n * n; i += T) Hg[i] = Hl[i]; // the damped system, for the next launch
        // the solvers destroy their matrix: work on a copy in V's place (LDLT, eigenvalue Jacobi) or keep H in V and rotate a
        // copy (one-sided SVD)
#if defined(MBAVO_EIG_STAMPS)
        ts2 = __builtin_amdgcn_s_memtime();
#endif
        // LDL^T in registers (wave 0) stands in for the Jacobi SVD (solver 0) or the pivoted LDL^T (solver 1) when every pivot is positive and the pivot ratio is
        // at most fast_ratio -- the host loop's rule (host_math.cpp:solve_spd_fast, MBAVO_FAST_SOLVE) -- and, refined in
        // double-double, up to refined_ratio when the refinement converges (lm_solvers.h: spd_solve_regs_impl; MBAVO_LM_REFINE=0
        // switches that off); any other system takes the solver below.  11 000 cycles (22 000 refined) against 138 000 for
        // the eigenvalue Jacobi or the pivoted LDL^T at n = 24 (tests/harness/solver_check.hip), and a launch lasts as long as
        // its slowest system: before the refined form, one system of 64 above the ratio held every launch for 75 us.
        bool have = false;
        if (o.fast_ratio > 0.0 && (n == 12 || n == 18 || n == 24)) // (solver 1 too: its pivoted LDL^T walks 2 000 cycles per column)
        {
            if (tid < 64)
            {
                bool ok;
                if (n == 12) ok = spd_solve_regs_impl<12, true>(H, g, x, lane, o.fast_ratio, o.refined_ratio);
                else if (n == 18) ok = spd_solve_regs_impl<18, true>(H, g, x, lane, o.fast_ratio, o.refined_ratio);
                else ok = spd_solve_regs_impl<24, true>(H, g, x, lane, o.fast_ratio, o.refined_ratio);
                if (tid == 0) order[0] = ok ? 1 : 0;
            }
            __syncthreads();
            have = order[0] != 0;
            __syncthreads(); // `order` is the Jacobi solvers' work area
            if (have && tid == 0) atomicAdd(num_done + 5, 1);
        }
        else if (o.fast_ratio > 0.0 && n <= 64)
        { // 5 .. 10 control knots: the same factorisation by the whole workgroup in LDS (V's place), the system read where it lives
            have = spd_solve_coop<T>(V, H, g, x, tmp, order, n, tid, o.fast_ratio, o.refined_ratio);
            if (have && tid == 0) atomicAdd(num_done + 5, 1);
        }
        if (have) {}
        else if constexpr (T == kEigT)
        {
          if (o.solver == 1)
          { // the reference's LDLT option in the wide workgroup (round 4: so that its candidates get their pose entries here too):
            // the pivoted LDL^T of lm_solvers.h by wave 0 on a copy, the other waves wait
            for (int i = tid; i < n * n; i += T) V[i] = H[i];
            __syncthreads();
            if (tid < 64) ldlt_solve(V, g, x, tmp, order, n, lane);
            __syncthreads();
          }
          else
          {
            const int info = eig_solve(V, Hl, g, x, tmp, order, n, tid);
            if (tid == 0)
            { // solver statistics in the spare words behind num_done (MBAVO_LM_STATS=1 prints them)
                atomicAdd(num_done + 1, 1);
                atomicAdd(num_done + 2, info & 255);
                atomicMax(num_done + 3, info & 255);
                atomicAdd(num_done + 4, info >> 8);
            }
          }
        }
        else if (o.solver == 1)
        {
            for (int i = lane; i < n * n; i += 64) V[i] = H[i];
            __syncthreads();
            ldlt_solve(V, g, x, tmp, order, n, lane);
        }
        else
        {
            double *G = tmp + 2 * n; // third area, n x (n + 1) like V
            const int ld = n + 1;
            for (int i = lane; i < n * n; i += 64) G[(i / n) * ld + i % n] = H[i];
            __syncthreads();
            svd_solve(G, V, g, x, tmp, n, ld, lane);
        }
#if defined(MBAVO_EIG_STAMPS)
        ts3 = __builtin_amdgcn_s_memtime();
#endif
        for (int i = tid; i < n; i += T) x[i] = -x[i];
        __syncthreads();
        double gx = 0.0, xHx = 0.0; // every wave sums the whole vectors
        for (int r = lane; r < n; r += 64)
        {
            gx += g[r] * x[r];
            double a = 0.0;
            for (int c = 0; c < n; ++c) a += H[c * n + r] * x[c];
            xHx += x[r] * a;
        }
        gx = wsum(gx);
        xHx = wsum(xHx);
        s.model = -(gx + 0.5 * xHx);
        if (s.model < 0)
        { // handleInvalidStep
            lm_rejected(s);
            trace_push(s, tr, o.trace_cap, tid, 0, 3, 0.0, s.model, 0.0);
            ++s.n_invalid;
            if (tid == 0) { active[b] = 0; states[b] = s; slot_publish(num_done, host_word, slot, B, false, pose_status); }
            return;
        }
        // computeCandidatePointAndEvaluateCost (:833-883): candidate = current (+) step, into the evaluated buffers
        double *Wt = const_cast<double *>(d.knots_t), *WR = const_cast<double *>(d.knots_R);
        // (a copy in LDS, V's place -- the solvers are done with it: the pose entries below read the candidate from there)
        double *Lt = V, *LR = V + 3 * N;
        for (int i = tid; i < 3 * N; i += T) { const double v = Ct[i] + x[i]; Wt[i] = v; Lt[i] = v; }
        for (int i = tid; i < N; i += T)
        {
            const Quat q = qmul(load_quat(CR + 4 * i), so3_exp(x + 3 * N + 3 * i)); // Spline.h:317-330, not re-normalised
            WR[4 * i] = q.x; WR[4 * i + 1] = q.y; WR[4 * i + 2] = q.z; WR[4 * i + 3] = q.w;
            LR[4 * i] = q.x; LR[4 * i + 1] = q.y; LR[4 * i + 2] = q.z; LR[4 * i + 3] = q.w;
        }
        if (tid == 0) { active[b] = 1; states[b] = s; slot_publish(num_done, host_word, slot, B, false, pose_status); }
        // The blur samples' pose entries of the candidate, WITH the knot Jacobians (round 4): the cost-only pass and -- if the
        // step is accepted -- the H/g pass of this slot evaluate at exactly these knots, so the entries are computed ONCE here,
        // by the workgroup that has the knots in its LDS, instead of by a pose launch / a pose prologue in each of the two passes
        // (Engine::set_external_poses; the arithmetic of k_pose_table: frame_pose_entries, one wave per knot).
        if constexpr (T >= 64 * KD)
        {
            if (pose_table != nullptr)
            {
                __syncthreads(); // the candidate is complete in LDS
                const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                SplineSeg *segs = (SplineSeg *)(V + 8 * N); // (8-byte aligned; S x (KD - 1) segments: sized by the host)
                for (int f = 0; f < F; ++f)
                {
                    frame_pose_entries<KD, true>(d, Lt, LR, f, pose_table + d.pose_base + f * d.S, segs, wave, lane, pose_status, true);
                    __syncthreads(); // the next frame overwrites the segments
                }
            }
        }
#if defined(MBAVO_EIG_STAMPS)
        if (tid == 0 && b == 0)
            printf("k_lm_solve<%d,%d> block 0: merge %lld | damp + store %lld | solve %lld | model + candidate %lld cycles\n", KD, T, ts1 - ts0,
                   ts2 - ts1, ts3 - ts2, (long long)__builtin_amdgcn_s_memtime() - ts3);
#endif
    }

    // One wave per problem, after the cost-only pass on the candidates: step quality, accept / reject, outliers.
    template <int KD>
    __global__ __launch_bounds__(64) void k_lm_decide(const ProblemDesc *__restrict__ descs, LmState *__restrict__ states, LmOpts o,
                                                      FinSrc fs, const double *__restrict__ patch_cost,
                                                      double *__restrict__ inv,
                                                      double *__restrict__ cur_t, double *__restrict__ cur_R,
                                                      int *__restrict__ active, mbavo_trace_rec *__restrict__ trace,
                                                      int *__restrict__ num_done, unsigned long long *host_word2, int slot, int B)
    {
        constexpr int ND = 6 * KD + 1, E = ND * (ND + 1) / 2;
        const int b = blockIdx.x, lane = threadIdx.x;
        const ProblemDesc &d = descs[b];
        int e_t0 = 0, e_t1 = 0;
        const bool early = fs.f1 && fs.deferred; // (see FinSrc::f1)
        if (early) { e_t0 = fs.tile_begin[b]; e_t1 = fs.tile_begin[b + 1]; }
        LmState s = states[b];
        if (s.done || active[b] == 0)
        { // finished, or an invalid step: nothing was evaluated
            if (lane == 0) decide_publish(num_done, host_word2, slot, B, lm_will_finish(s, o));
            return;
        }
        mbavo_trace_rec *tr = trace ? trace + (size_t)b * o.trace_cap : nullptr;
        // the patch costs of frame 0 (up to kPre per lane) are fetched WITH the costs, before the accept test needs either: the
        // three statistics passes below then run from registers (same order of additions: bit-identical flags)
        constexpr int kPre = 8;
        const double *pc = patch_cost + d.patch_base;
        const bool pre = d.K <= 64 * kPre;
        double pcv[kPre];
#pragma unroll
        for (int q = 0; q < kPre; ++q) pcv[q] = (pre && lane + 64 * q < d.K) ? pc[lane + 64 * q] : 0.0;
        double cost = 0.0;
        if (early) cost = slot_sum_range(fs, e_t0, e_t1 - e_t0, E);
        else
            for (int f = 0; f < d.F; ++f) cost += slot_cost<E>(fs, d.bf_base + f);
        s.cand_cost = cost;
        s.abs_dec = s.eval_cost - s.cand_cost; // recorded before the accept test (:624)
        s.quality = tr_quality(s, s.cand_cost, s.model);
        if (s.quality > o.min_q && s.cand_cost < s.eval_cost)
        { // isStepSuccessful (:890-894) -> detectOutliersAndUploadToGpu (:639-699): patch costs of frame 0
            unsigned char *flags = const_cast<unsigned char *>(d.outlier);
            double sum = 0.0, cnt = 0.0, var = 0.0, nbad = 0.0;
            if (pre)
            {
#pragma unroll
                for (int q = 0; q < kPre; ++q)
                    if (lane + 64 * q < d.K && !(pcv[q] < 1e-8)) { sum += pcv[q]; cnt += 1.0; }
            }
            else
                for (int i = lane; i < d.K; i += 64)
                {
                    const double c = pc[i];
                    if (c < 1e-8) continue;
                    sum += c;
                    cnt += 1.0;
                }
            sum = wsum(sum);
            cnt = wsum(cnt);
            const double mu = sum / cnt;
            if (pre)
            {
#pragma unroll
                for (int q = 0; q < kPre; ++q)
                    if (lane + 64 * q < d.K && !(pcv[q] < 1e-8)) var += (pcv[q] - mu) * (pcv[q] - mu);
            }
            else
                for (int i = lane; i < d.K; i += 64)
                {
                    const double c = pc[i];
                    if (c < 1e-8) continue;
                    var += (c - mu) * (c - mu);
                }
            var = wsum(var) / cnt;
            const double bound = o.chi * (double)sqrtf((float)var);
            if (pre)
            {
#pragma unroll
                for (int q = 0; q < kPre; ++q)
                    if (lane + 64 * q < d.K && fabs(pcv[q] - mu) > bound) { flags[lane + 64 * q] = 1; nbad += 1.0; }
            }
            else
                for (int i = lane; i < d.K; i += 64)
                    if (fabs(pc[i] - mu) > bound) { flags[i] = 1; nbad += 1.0; }
            s.num_bad = (int)wsum(nbad);
            const long long num_residuals = (long long)(d.K - s.num_bad) * d.F * d.P;
            // accept: the candidate becomes the current point, the next pass re-evaluates H/g there
            double *Ct = cur_t + (size_t)b * 3 * o.max_N, *CR = cur_R + (size_t)b * 4 * o.max_N;
            for (int i = lane; i < 3 * d.N; i += 64) Ct[i] = d.knots_t[i];
            for (int i = lane; i < 4 * d.N; i += 64) CR[i] = d.knots_R[i];
            s.fresh = 1;
            s.pending_accept = 1;
            if (lane == 0)
            {
                inv[b] = num_residuals > 0 ? 1.0 / (double)num_residuals : 0.0;
                active[b] = 2;
                states[b] = s;
                decide_publish(num_done, host_word2, slot, B, lm_will_finish(s, o));
            }
            return;
        }
        lm_rejected(s); // handleUnsuccessfulStep
        trace_push(s, tr, o.trace_cap, lane, 0, 2, s.cand_cost, s.model, s.quality);
        ++s.n_reject;
        if (lane == 0) { active[b] = 0; states[b] = s; decide_publish(num_done, host_word2, slot, B, lm_will_finish(s, o)); }
    }

#define LM_HIP(expr)                                                                        \
    do                                                                                      \
    {                                                                                       \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
        {                                                                                   \
            fprintf(stderr, "mbavo lm_batch: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
            rc = (int)e_;                                                                   \
            goto done;                                                                      \
        }                                                                                   \
    } while (0)

    int lm_batch(Engine &eng, int B, const mbavo_problem *probs, const mbavo_lm_batch_opts &opt, mbavo_lm_batch_result *results,
                 mbavo_trace_rec *trace, int trace_cap, const LmBatchShared *shared)
    {
        const int k = opt.spline_deg_k;
        if (B < 1 || !probs || (k != 2 && k != 4) || (opt.solver_type != 0 && opt.solver_type != 1) || opt.max_num_iterations < 0)
            return MBAVO_E_ARG;
        int rc = 0;
        hipStream_t st = eng.stream();
        // MBAVO_LM_STAMPS=1: host-side phases of a call on stderr (development aid)
        const bool stamps = getenv("MBAVO_LM_STAMPS") && getenv("MBAVO_LM_STAMPS")[0] == '1';
        std::chrono::steady_clock::time_point tp[6];
        auto stamp = [&](int i) { if (stamps) tp[i] = std::chrono::steady_clock::now(); };
        stamp(0);
        const int E = (6 * k + 1) * (6 * k + 2) / 2;
        int max_N = 0, nbf = 0;
        long long total_K = 0, total_patches = 0;
        for (int b = 0; b < B; ++b)
        {
            if (probs[b].N < k || probs[b].N > 16 || !probs[b].h_start_idx || probs[b].F < 1) return MBAVO_E_ARG;
            max_N = probs[b].N > max_N ? probs[b].N : max_N;
            nbf += probs[b].F;
            total_K += probs[b].K > 0 ? probs[b].K : 1;
            total_patches += (long long)probs[b].F * probs[b].K;
        }
        if (shared && shared->max_N > max_N) max_N = shared->max_N; // (a group of a bigger batch: the batch's strides and kernel form)
        const int max_n = 6 * max_N;
        LmOpts o;
        o.max_it = opt.max_num_iterations; o.max_nonmono = opt.max_consecutive_nonmonotonic_steps; o.solver = opt.solver_type;
        o.trace_cap = trace ? trace_cap : 0; o.max_n = max_n; o.max_N = max_N;
        o.min_q = opt.min_step_quality; o.min_dec = opt.min_abs_cost_decrease; o.chi = opt.max_chi_square_error;
        // the solver forms and the schedule of this call: mbavo_lm_batch_opts' tail under the environment's override layer (options.h)
        const EnvOverrides env = read_env_overrides();
        o.fast_ratio = opt_fast_ratio(opt.fast_solve_ratio, env.fast_solve);                 // default 1e8; 0: the Jacobi solvers / the pivoted LDL^T only
        o.refined_ratio = opt_refined_ratio(opt.refined_ratio, env.lm_refine, o.fast_ratio); // default 1e13; 0: the plain stand-in only
        // every system within the wide workgroup's reach (k_lm_solve<KD, kEigT>: solver 0 falls back to the workgroup-parallel
        // eigenvalue Jacobi there, solver 1 to the pivoted LDL^T on wave 0; both get their candidates' pose entries from it);
        // mbavo_lm_batch_opts.eig = -1 keeps the one-wave one-sided sweeps
        // (solver type 1 takes the wide workgroup as well: its stand-in is the same, its fallback the pivoted LDL^T on wave 0)
        const bool eig = max_n <= kEigMaxN && opt_flag(opt.eig, env.lm_eig, true);
        const size_t lds_solver = eig ? (eig_lds_doubles(max_n) + 3 * max_n + (size_t)max_n * max_n) * sizeof(double) + (size_t)(4 + 2 * max_n) * sizeof(int)
                               : ((size_t)2 * max_n * (max_n + 1) + 6 * max_n) * sizeof(double) + (size_t)max_n * sizeof(int);
        // the solve kernel's pose entries (eigenvalue-Jacobi form only: it has the KD waves): candidate knots + the segments of
        // a frame's samples in the solvers' area
        int max_S = 1;
        for (int b = 0; b < B; ++b) max_S = probs[b].S > max_S ? probs[b].S : max_S;
        if (shared && shared->max_S > max_S) max_S = shared->max_S;
        const bool ext_poses = eig && opt_flag(opt.pose_entries, env.lm_poses, true);
        const size_t lds_pose = ext_poses ? (size_t)8 * max_N * sizeof(double) + (size_t)(max_S < kPoseSPB ? max_S : kPoseSPB) * (k - 1) * sizeof(SplineSeg) : 0;
        const size_t lds = lds_solver > lds_pose ? lds_solver : lds_pose;
        if (lds > 160 * 1024) return MBAVO_E_ARG;

        // one arena for all LM state
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
        // head: what the host initialises, contiguous so that ONE copy from pinned memory uploads it (three pageable copies and
        // a fill of the whole arena cost ~35 us of a 64-pair call: four blit kernels with a ~6 us gap behind each)
        const size_t o_inv = take(sizeof(double) * B), o_act = take(sizeof(int) * B), o_done = take(sizeof(int) * 16),
                     o_start = take(sizeof(int) * nbf), o_flags = take((size_t)total_K), o_state = take(sizeof(LmState) * B),
                     head_bytes = off; // (the initial LM states ride in the head: no init launch)
        const size_t o_H = take(sizeof(double) * (size_t)B * max_n * max_n),
                     o_g = take(sizeof(double) * (size_t)B * max_n), o_ct = take(sizeof(double) * (size_t)B * 3 * max_N),
                     o_cR = take(sizeof(double) * (size_t)B * 4 * max_N), o_fb = take(sizeof(double) * (size_t)nbf * E),
                     o_pc = take(sizeof(double) * (size_t)(total_patches + 1)),
                     o_trace = take(sizeof(mbavo_trace_rec) * (size_t)B * (trace ? trace_cap : 0));
        char *base = nullptr;
        std::vector<mbavo_problem> work(probs, probs + B);
        const LmState *h_states = nullptr;
        char *h_stage = nullptr; // pinned: [done word (64 bytes) | head | final states]
        int h_done = 0;
        LM_HIP(hipSetDevice(eng.device()));
        // the engine's scratch slot 15, grown on demand and kept: no hipMalloc / hipFree per call (~90 us of a 64-pair call)
        base = (char *)eng.named_scratch(15, off);
        if (!base) { rc = (int)hipErrorOutOfMemory; goto done; }
        h_stage = (char *)eng.pinned_scratch(7, 64 + head_bytes + sizeof(LmState) * B);
        if (!h_stage) { rc = (int)hipErrorOutOfMemory; goto done; }
        h_states = (const LmState *)(h_stage + 64 + head_bytes); // (head_bytes is a multiple of 256)
        {
            LmState *states = (LmState *)(base + o_state);
            double *Hst = (double *)(base + o_H), *gst = (double *)(base + o_g), *ct = (double *)(base + o_ct), *cR = (double *)(base + o_cR);
            double *inv = (double *)(base + o_inv), *fb = (double *)(base + o_fb), *pc = (double *)(base + o_pc);
            int *act = (int *)(base + o_act), *num_done = (int *)(base + o_done), *d_start = (int *)(base + o_start);
            unsigned char *flags = (unsigned char *)(base + o_flags);
            mbavo_trace_rec *d_trace = trace ? (mbavo_trace_rec *)(base + o_trace) : nullptr;
            char *head = h_stage + 64;
            memset(head, 0, head_bytes); // done counter, statistics, ticket and outlier flags start at zero
            double *h_inv = (double *)(head + o_inv);
            int *h_act = (int *)(head + o_act), *h_start = (int *)(head + o_start);
            size_t fo = 0;
            int bf = 0;
            for (int b = 0; b < B; ++b)
            { // per-level reset of the outlier flags and count (:600-601): the engine owns them here
                work[b].d_outlier = flags + fo;
                work[b].num_bad = 0;
                fo += work[b].K > 0 ? work[b].K : 1;
                for (int f = 0; f < work[b].F; ++f) h_start[bf++] = work[b].h_start_idx[f];
                // num_bad = 0 at the start of a level (:600); every problem takes part in the first H/g pass
                const long long num_residuals = (long long)work[b].K * work[b].F * work[b].P;
                h_inv[b] = num_residuals > 0 ? 1.0 / (double)num_residuals : 0.0;
                h_act[b] = 2;
                LmState &s0 = ((LmState *)(head + o_state))[b]; // (zero from the memset above)
                s0.radius = 1e4; s0.decrease_factor = 2.0; s0.abs_dec = 1e10; s0.fresh = 1;
            }
            stamp(1);
            LM_HIP(hipMemcpyAsync(base, head, head_bytes, hipMemcpyHostToDevice, st));
            stamp(2);
            // (states: k_lm_init; H, g: the first solve of a problem; patch costs and frame blocks: the passes that are read)
            if (trace) LM_HIP(hipMemsetAsync(d_trace, 0, sizeof(mbavo_trace_rec) * (size_t)B * trace_cap, st));
            // sync_every > 0: the host drains the stream and reads the done-counter every that many iterations (round 1's scheme:
            // up to sync_every - 1 iterations of idle launches after the last problem has finished, and a pipeline bubble at every
            // read).  sync_every <= 0 (default): the last workgroup of every solve launch stores (slot + 1, problems done) into a
            // pinned host word (slot_publish); the host enqueues slot i's solve AND its three passes, then spins on that word until
            // solve i has been published -- nothing but kernels is in the stream (the D2H copy + event of the previous scheme cost a
            // 4.4 us blit kernel and a 5.8 us gap behind it per slot: tools/lm_timeline.py).
            const int sync_every = opt.sync_every;
            volatile unsigned long long *h_word = nullptr, *h_word2 = nullptr;
            int *h_status = (int *)(h_stage + 32); // pinned: the engine's range-status counter lands here with the final copies
            if (sync_every <= 0)
            {
                h_word = (volatile unsigned long long *)h_stage;  // solve launches: (slot + 1, problems done)
                h_word2 = h_word + 1;                             // decide launches: (slot + 1, problems the next solve will end)
                *h_word = 0; // (no kernel that writes them is in flight: every call ends with a stream synchronisation)
                *h_word2 = 0;
            }
            // defer_finalize = -1: the engine's finalize kernels write frame blocks and the LM kernels read those
            eng.set_defer_finalize(opt_flag(opt.defer_finalize, env.lm_defer, true));
            const auto t_sub0 = std::chrono::steady_clock::now();
            // iteration 0 (:604): also builds the layout (device descriptors) the LM kernels read
            if ((rc = eng.evaluate(B, work.data(), k, true, fb, pc, nullptr, nullptr, act, inv)) != 0) goto done;
            const auto t_sub1 = std::chrono::steady_clock::now();
            if (stamps)
                fprintf(stderr, "mbavo lm_batch:   first evaluation: %.1f us before the call (trace memset, words), %.1f us inside Engine::evaluate\n",
                        std::chrono::duration<double, std::micro>(t_sub0 - tp[2]).count(), std::chrono::duration<double, std::micro>(t_sub1 - t_sub0).count());
            const ProblemDesc *descs = eng.device_descs();
            FinSrc fs;
            fs.fb = fb; fs.partials = eng.device_partials(); fs.tile_begin = eng.device_bf_tile_begin();
            fs.stride = E + 2; // engine.hip: Pack<k>::PSTRIDE
            fs.deferred = eng.finalize_deferred() ? 1 : 0;
            fs.f1 = nbf == B ? 1 : 0; // (every F >= 1 was checked above: nbf == B means every F == 1)
            fs.inv = inv;
            // RE-TILING for the late slots of a big batch (round 4).  A batch of more pairs than CUs is tiled one tile per pair -- three
            // rounds of pixels per workgroup, the right grain while most pairs are active -- and a pass then lasts ~38 us as long as ONE
            // pair is active.  A second layout of the same list with four tiles per pair lives in the engine's companion (built while
            // the first evaluation runs); once few enough pairs are left that their fine tiles fit the CUs, both passes of a slot go
            // through it: one round per workgroup.  The LM kernels are told per launch whose partials to sum (FinSrc).
            // retile = -1: one layout.
            Engine *fine = nullptr;
            FinSrc fs_fine = fs;
            if (sync_every <= 0 && fs.deferred && ext_poses && eng.num_tiles() < 4 * nbf && eng.num_tiles() * 2 >= eng.num_cus() &&
                opt_flag(opt.retile, env.lm_retile, true))
            {
                fine = eng.companion();
                fine->set_tile_target(4ll * nbf);
                if ((rc = fine->prepare(B, work.data(), k, act, inv)) != 0) goto done;
                if (fine->layout_flat() && fine->num_tiles() > eng.num_tiles())
                {
                    fs_fine.partials = fine->device_partials();
                    fs_fine.tile_begin = fine->device_bf_tile_begin();
                    fs_fine.deferred = 1;
                }
                else
                    fine = nullptr;
            }
            bool use_fine = false;         // the layout of the passes enqueued last
            FinSrc fs_hg = fs, fs_cost = fs; // whose partials the next solve / decide launch sums
            stamp(3);
            if (lds > 48 * 1024)
            { // more than 8 control knots: the three n x n areas need the large-LDS attribute
                if (k == 4) LM_HIP(hipFuncSetAttribute(eig ? (const void *)k_lm_solve<4, kEigT> : (const void *)k_lm_solve<4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                else LM_HIP(hipFuncSetAttribute(eig ? (const void *)k_lm_solve<2, kEigT> : (const void *)k_lm_solve<2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            unsigned long long *d_word = const_cast<unsigned long long *>(h_word); // pinned host memory is device-visible at its own address
            // (the look-ahead word is only read for batches of up to 128 problems, see below: bigger batches' decide launches skip
            // their B atomics on one address)
            unsigned long long *d_word2 = B <= 128 ? const_cast<unsigned long long *>(h_word2) : nullptr;
            // bounded spin on a pinned word until its slot number reaches `want`; the value, or 0 after a time-out
            auto spin_for = [&](volatile unsigned long long *word, unsigned long long want) -> unsigned long long {
                unsigned long long w = *word;
                if ((w >> 32) >= want) return w;
                const auto t_spin = std::chrono::steady_clock::now();
                for (unsigned long spins = 1; ((w = *word) >> 32) < want; ++spins)
                {
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                    if ((spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t_spin > std::chrono::seconds(10)) return 0;
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                return w;
            };
#define LM_DECIDE_ARGS descs, states, o, fs_cost, pc, inv, ct, cR, act, d_trace, num_done, d_word2, slot, B
            std::vector<double> slot_us; // MBAVO_LM_STAMPS=1: per slot [solve launch | look-ahead wait | passes enqueued | solve word wait]
            auto now_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp[0]).count(); };
            for (int slot = 0; slot <= o.max_it + 1; ++slot)
            {
                const double ts0 = stamps ? now_us() : 0.0;
#define LM_SOLVE_ARGS(KD) descs, states, o, fs_hg, d_start, Hst, gst, ct, cR, act, d_trace, num_done, d_word, slot, B, \
                          (PoseEntry<KD> *)(ext_poses ? eng.device_pose_table() : nullptr), eng.device_status(), \
                          (LmState *)(sync_every <= 0 && !trace ? h_states : nullptr)
                if (k == 4 && eig)
                    hipLaunchKernelGGL((k_lm_solve<4, kEigT>), dim3(B), dim3(kEigT), lds, st, LM_SOLVE_ARGS(4));
                else if (k == 4)
                    hipLaunchKernelGGL((k_lm_solve<4, 64>), dim3(B), dim3(64), lds, st, LM_SOLVE_ARGS(4));
                else if (eig)
                    hipLaunchKernelGGL((k_lm_solve<2, kEigT>), dim3(B), dim3(kEigT), lds, st, LM_SOLVE_ARGS(2));
                else
                    hipLaunchKernelGGL((k_lm_solve<2, 64>), dim3(B), dim3(64), lds, st, LM_SOLVE_ARGS(2));
#undef LM_SOLVE_ARGS
                eng.set_external_poses(ext_poses); // from here on the passes read the entries the solve launches leave in the table
                if (sync_every <= 0)
                {
                    // the three passes of this slot go in behind the solve BEFORE the host looks at the solve's word: the device
                    // has ~45 us of work queued while the host waits.  Look-ahead: the PREVIOUS slot's decide launch has told
                    // how many problems this solve will end (decide_publish; that launch retired at least a whole H/g pass ago
                    // by the time the device gets here, so the wait below is for the host's own benefit, not the device's);
                    // if that is every one, nothing is enqueued behind the solve.
                    const bool last = slot == o.max_it + 1;
                    bool ending = last;
                    const double ts1 = stamps ? now_us() : 0.0;
                    // (batches of up to 128 problems only: a big batch's passes take the host 20-90 us to enqueue, which must overlap
                    // the device's previous slot -- waiting for the decide word first cost the 512-pair call 54 us in bubbles for
                    // 30 us of idle launches saved)
                    if (!last && slot > 0 && B <= 128)
                    {
                        const unsigned long long w2 = spin_for(h_word2, (unsigned long long)slot);
                        if (w2 == 0) { LM_HIP(hipStreamSynchronize(st)); LM_HIP(hipGetLastError()); rc = MBAVO_E_TIMEOUT; goto done; }
                        ending = (int)(unsigned)(w2 & 0xffffffffull) >= B;
                    }
                    auto enqueue_passes = [&]() -> int {
                        // (h_done: the count the previous slot's solve published)
                        if (fine != nullptr && !use_fine && (long long)(B - h_done) * 4 <= eng.num_cus())
                        {
                            use_fine = true;
                            fine->set_defer_finalize(true);
                            fine->set_external_poses(true, eng.device_pose_table());
                        }
                        Engine &E = use_fine ? *fine : eng;
                        fs_cost = use_fine ? fs_fine : fs;
                        int r = E.evaluate(B, work.data(), k, false, fb, pc, nullptr, nullptr, act, inv, false, true);
                        if (r != 0) return r;
                        if (k == 4)
                            hipLaunchKernelGGL((k_lm_decide<4>), dim3(B), dim3(64), 0, st, LM_DECIDE_ARGS);
                        else
                            hipLaunchKernelGGL((k_lm_decide<2>), dim3(B), dim3(64), 0, st, LM_DECIDE_ARGS);
                        r = E.evaluate(B, work.data(), k, true, fb, pc, nullptr, nullptr, act, inv, false, true);
                        fs_hg = fs_cost; // (the next solve launch sums this pass' partials)
                        return r;
                    };
                    const double ts2 = stamps ? now_us() : 0.0;
                    if (!ending && (rc = enqueue_passes()) != 0) goto done;
                    const double ts3 = stamps ? now_us() : 0.0;
                    const unsigned long long w = spin_for(h_word, (unsigned long long)slot + 1);
                    if (stamps) { slot_us.push_back(ts1 - ts0); slot_us.push_back(ts2 - ts1); slot_us.push_back(ts3 - ts2); slot_us.push_back(now_us() - ts3); }
                    if (w == 0)
                    { // a launch failed or the device is wedged: let the runtime say which
                        LM_HIP(hipStreamSynchronize(st));
                        LM_HIP(hipGetLastError());
                        rc = MBAVO_E_TIMEOUT; // (not MBAVO_E_RANGE: a wedged device is not an out-of-range capture time)
                        goto done;
                    }
                    h_done = (int)(unsigned)(w & 0xffffffffull);
                    if (h_done >= B || last) break;
                    if (ending && (rc = enqueue_passes()) != 0) goto done; // (the look-ahead over-counted: cannot happen, but never hang on it)
                    continue;
                }
                else if (slot % sync_every == sync_every - 1 || slot == o.max_it + 1)
                {
                    LM_HIP(hipMemcpyAsync(&h_done, num_done, sizeof(int), hipMemcpyDeviceToHost, st));
                    LM_HIP(hipStreamSynchronize(st));
                    if (h_done >= B) break;
                }
                if ((rc = eng.evaluate(B, work.data(), k, false, fb, pc, nullptr, nullptr, act, inv)) != 0) goto done;
                if (k == 4)
                    hipLaunchKernelGGL((k_lm_decide<4>), dim3(B), dim3(64), 0, st, LM_DECIDE_ARGS);
                else
                    hipLaunchKernelGGL((k_lm_decide<2>), dim3(B), dim3(64), 0, st, LM_DECIDE_ARGS);
                if ((rc = eng.evaluate(B, work.data(), k, true, fb, pc, nullptr, nullptr, act, inv)) != 0) goto done;
            }
            stamp(4);
            LM_HIP(hipGetLastError());
            if (sync_every <= 0 && !trace && h_done >= B)
            { // every problem's final state and the status counter are in pinned memory already (k_lm_solve, slot_publish); what is
              // still in flight is the tail of the last solve launch: poll the stream instead of a blocking wait (~20 us to wake up)
                const auto t_q = std::chrono::steady_clock::now();
                hipError_t q;
                while ((q = hipStreamQuery(st)) == hipErrorNotReady)
                    if (std::chrono::steady_clock::now() - t_q > std::chrono::seconds(10)) break;
                if (q != hipSuccess) LM_HIP(hipStreamSynchronize(st));
                __atomic_thread_fence(__ATOMIC_ACQUIRE); // the device's release store of the word -> the final states and the status below
                *h_status = *reinterpret_cast<volatile int *>(const_cast<unsigned long long *>(h_word) + 4);
            }
            else
            {
                LM_HIP(hipMemcpyAsync(const_cast<LmState *>(h_states), states, sizeof(LmState) * B, hipMemcpyDeviceToHost, st)); // pinned: no staging
                if (trace) LM_HIP(hipMemcpyAsync(trace, d_trace, sizeof(mbavo_trace_rec) * (size_t)B * trace_cap, hipMemcpyDeviceToHost, st));
                if ((rc = eng.fetch_status_enqueue(h_status)) != 0) goto done; // (no blocking copy of its own after the drain: ~10 us)
                LM_HIP(hipStreamSynchronize(st));
            }
            stamp(5);
            if (stamps)
            {
                auto us = [&](int a, int b) { return std::chrono::duration<double, std::micro>(tp[b] - tp[a]).count(); };
                fprintf(stderr, "mbavo lm_batch: [stream %p, %d problems] host set-up %.1f us | head upload call %.1f | first evaluation enqueued %.1f | LM slots %.1f | results + drain %.1f\n",
                        (void *)st, B, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5));
                for (size_t i = 0; i + 3 < slot_us.size(); i += 4)
                    fprintf(stderr, "mbavo lm_batch:   slot %zu: solve launch %.1f us | look-ahead wait %.1f | passes enqueued %.1f | solve word wait %.1f\n", i / 4,
                            slot_us[i], slot_us[i + 1], slot_us[i + 2], slot_us[i + 3]);
            }
            if (eng.fetch_status_take(h_status) != 0) { rc = MBAVO_E_RANGE; goto done; } // a capture time outside the spline somewhere in the call
            if (h_done < B)
            {
                LM_HIP(hipMemcpy(&h_done, num_done, sizeof(int), hipMemcpyDeviceToHost));
                if (h_done < B) { rc = MBAVO_E_RANGE; goto done; }
            }
            if (const char *e = getenv("MBAVO_LM_STATS"); e && e[0] == '1' && eig)
            {
                int st4[6] = {0, 0, 0, 0, 0, 0};
                LM_HIP(hipMemcpy(st4, num_done, sizeof(st4), hipMemcpyDeviceToHost));
                fprintf(stderr, "mbavo lm_batch: %d LDL^T stand-ins; %d eigenvalue-Jacobi solves, %.2f sweeps on average, %d at most, %d preconditioned (L^T L)\n",
                        st4[5], st4[1], st4[1] ? (double)st4[2] / st4[1] : 0.0, st4[3], st4[4]);
            }
            if (results)
                for (int b = 0; b < B; ++b)
                {
                    const LmState &s = h_states[b];
                    mbavo_lm_batch_result &r = results[b];
                    r.iterations = s.iter; r.accepted = s.n_accept; r.rejected = s.n_reject; r.invalid = s.n_invalid;
                    r.num_outliers = s.num_bad; r.num_trace = s.ntrace;
                    r.initial_cost = s.initial_cost; r.final_cost = s.eval_cost; r.radius = s.radius;
                }
        }
    done:
        // an error exit may leave solve launches queued that still store into the pinned done word and the arena: drain them
        // before the next call re-arms the word or regrows either buffer
        if (rc != 0)
        {
            (void)hipStreamSynchronize(st);
            // range-status increments of the failed call must not be reported by the next, successful one: bring
            // the engine's `seen` count up to the device's counter
            (void)eng.fetch_status();
        }
        eng.set_external_poses(false);
        eng.set_defer_finalize(false);
        if (Engine *c = eng.companion_if_any()) { c->set_external_poses(false); c->set_defer_finalize(false); }
        return rc > 0 ? -1000 - rc : rc;
    }
} // namespace mbavo
