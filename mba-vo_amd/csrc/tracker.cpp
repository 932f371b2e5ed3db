// tracker.cpp -- coarse-to-fine Levenberg-Marquardt loop of the blur-aware tracker on the
// fused engine: BlurAwareDirectTracker::optimizeTrajectory / optimizePyramidLevel and the
// helpers they call (ba_tracker/blur_aware_direct_tracker.cpp:544-924).
//
// Control-flow details that decide how often the hot path runs and which step is taken
// are kept as in the reference: the damping H.diag += H.diag / radius is applied in
// place and therefore accumulates over rejected steps (:801-803); the model cost change
// uses the damped H (:821-823); abs_cost_decrease is recorded before the accept test, so
// any non-improving step ends the level (:624, :910-924); outlier statistics come from
// the patch costs of the candidate's cost-only pass, frame 0 only (:639-699); the LM
// radius and outlier flags are reset per level, the spline is warm-started (:590-606).
#include "tracker.h"
#include "host_math.h"
#include "se3_math.h"
#include "timing.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mbavo
{
#define TRK_HIP(expr)                                                                    \
    do                                                                                   \
    {                                                                                    \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
        {                                                                                \
            fprintf(stderr, "mbavo tracker: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
            rc_ = (int)e_;                                                               \
            goto done;                                                                   \
        }                                                                                \
    } while (0)

    // `shadow` (may be null): the flags in ordinary host memory (the persistent kernels' flag bytes live in write-combined device
    // memory, which the host cannot read back at speed); *newly = how many patches this call flagged for the first time
    static int detect_outliers(const double *patch_cost, int K, double chi, unsigned char *flags, unsigned char *shadow = nullptr,
                               int *newly = nullptr)
    { // :639-699
        double sum = 0.0;
        std::vector<double> kept;
        kept.reserve(K);
        for (int i = 0; i < K; ++i)
        {
            const double c = patch_cost[i];
            if (c < 1e-8) continue;
            kept.push_back(c);
            sum += c;
        }
        const double mu = sum / kept.size();
        double var = 0.0;
        for (double c : kept) var += (c - mu) * (c - mu);
        var = var / kept.size();
        const double bound = chi * (double)sqrtf((float)var);
        int n = 0, fresh = 0;
        for (int i = 0; i < K; ++i)
            if (std::fabs(patch_cost[i] - mu) > bound)
            {
                flags[i] = 1;
                ++n;
                if (shadow) { fresh += shadow[i] == 0; shadow[i] = 1; }
            }
        if (newly) *newly = shadow ? fresh : n;
        return n;
    }

    int optimize_trajectory(Engine &eng, const mbavo_track_opts &o, const mbavo_level *levels, int F,
                            const double *h_cap, const double *h_exp, double t0, double dt, double *knots_t,
                            double *knots_R, int N, int *start_idx_out, double *final_cost, mbavo_trace_rec *trace,
                            int trace_cap)
    {
        const int k = o.spline_deg_k;
        if ((k != 2 && k != 4) || F < 1 || N < k || o.num_levels < 1 || o.num_levels > 8) return MBAVO_E_ARG;
        const int n = 6 * N, ndim = 6 * k + 1, E = ndim * (ndim + 1) / 2;
        int rc_ = 0, ntrace = 0;
        hipStream_t st = eng.stream();
        const EnvOverrides env = read_env_overrides(); // (options.h: the A/B tools' override layer over the options below)
        const double fast_ratio = opt_fast_ratio(o.fast_solve_ratio, env.fast_solve); // once per call (mbavo_lm_batch does the same)

        SLAM::Core::SplineSE3 spline(t0, dt);
        spline.setSplineDegK(k);
        for (int i = 0; i < N; ++i) spline.InsertControlKnot(knots_R + 4 * i, knots_t + 3 * i);

        std::vector<int> start_idx(F);
        for (int f = 0; f < F; ++f) start_idx[f] = (int)((h_cap[f] - t0) / dt); // :549-560
        if (start_idx_out) memcpy(start_idx_out, start_idx.data(), sizeof(int) * F);

        std::vector<double> H((size_t)n * n), g(n), step(n), cand_t(3 * N), cand_R(4 * N);
        // Speculation (mbavo_track_opts.speculate): the candidate is evaluated WITH H / g.  An accepted step is followed by an H / g evaluation
        // at the very same knots (:896-903); it differs from the candidate's only through the outlier flags and the residual scale
        // detectOutliers may have changed in between (:639-699) -- where it changed neither, the candidate's H, g and cost ARE that
        // evaluation's, bit for bit, and it is skipped.
        // Worth it where an evaluation is latency-bound and H / g cost little more than the cost alone: the levels that run on
        // persistent kernels (trackFrame 0.330 -> 0.322 ms per frame, 172 -> 154 evaluations of 12.8 / 14.2 us over 8 tracked
        // frames, identical records and poses); dense levels pay twice the cost-only pass for a one-in-three hit.
        // mbavo_track_opts.speculate = -1 / 1: never / on every level.
        const int speculate_env = env.speculate != kEnvUnset ? (env.speculate == 0 ? 0 : (env.speculate == 1 ? 1 : -1))
                                                              : (o.speculate == 0 ? -1 : (o.speculate > 0 ? 1 : 0)); // -1: on the persistent levels
        std::vector<double> Hs(speculate_env != 0 ? (size_t)n * n : 0), gs(speculate_env != 0 ? n : 0);
        std::vector<unsigned char> shadow;
        SLAM::VO::LevenbergMarquardtStrategy lm;
        SLAM::VO::TrustRegionStepEvaluator evaluator(o.max_consecutive_nonmonotonic_steps);
        double eval_cost = 0.0;

        int maxK = 0;
        size_t sumK = 0;
        for (int l = 0; l < o.num_levels; ++l) maxK = levels[l].K > maxK ? levels[l].K : maxK;
        double *d_cap = nullptr, *d_exp = nullptr, *d_kt = nullptr, *d_kR = nullptr, *d_pc = nullptr;
        unsigned char *d_flags = nullptr;
        double *h_pin = nullptr, *h_inv = nullptr; // pinned: frame blocks; 1 / ((K - bad) F P), read by the kernels
        unsigned char *flags = nullptr; // pinned host staging of the outlier flags

        TRK_HIP(hipSetDevice(eng.device()));
        // engine-owned scratch, reused by every call (no hipMalloc / hipFree in the tracking loop)
        d_cap = (double *)eng.named_scratch(0, sizeof(double) * 2 * F);
        d_exp = d_cap ? d_cap + F : nullptr;
        // the knots live in pinned, device-visible host memory [t (3N) | R (4N)]: the host writes them before an
        // evaluation and the kernels' pose prologue reads them over the bus -- no copy, no launch
        d_kt = (double *)eng.pinned_scratch(2, sizeof(double) * 7 * N);
        d_kR = d_kt ? d_kt + 3 * N : nullptr;
        // the per-patch costs land in pinned host memory (written by the fused kernel, read by detect_outliers after the
        // evaluation's synchronisation: no D2H copy); the flags are staged in pinned memory so that their upload is a
        // truly asynchronous copy
        // (sized for ALL levels side by side: one persistent kernel may serve every level of the call, see `joint` below)
        for (int l = 0; l < o.num_levels; ++l) sumK += (size_t)(levels[l].K > 0 ? levels[l].K : 1);
        d_pc = (double *)eng.pinned_scratch(0, sizeof(double) * (size_t)F * sumK);
        flags = (unsigned char *)eng.pinned_scratch(1, maxK > 0 ? maxK : 1);
        d_flags = (unsigned char *)eng.named_scratch(6, maxK > 0 ? maxK : 1);
        h_pin = eng.host_frame_blocks((size_t)F * E * o.num_levels); // device-visible pinned host memory
        if (h_pin) memset(h_pin, 0, sizeof(double) * (size_t)F * E * o.num_levels);
        h_inv = (double *)eng.pinned_scratch(3, sizeof(double));
        if (!h_inv || !d_cap || !d_exp || !d_kt || !d_kR || !d_pc || !d_flags || !flags || !h_pin) { rc_ = (int)hipErrorOutOfMemory; goto done; }
        { // capture / exposure times: one asynchronous copy from pinned staging [cap F | exp F] (:701-719)
            double *stage = (double *)eng.pinned_scratch(4, sizeof(double) * 2 * F);
            if (!stage) { rc_ = (int)hipErrorOutOfMemory; goto done; }
            memcpy(stage, h_cap, sizeof(double) * F);
            memcpy(stage + F, h_exp, sizeof(double) * F);
            TRK_HIP(hipMemcpyAsync(d_cap, stage, sizeof(double) * 2 * F, hipMemcpyHostToDevice, st));
        }

        {
        // One run per pyramid level, coarse to fine (:571-575).  Small levels (everything the reference's semi-dense detector
        // produces) are ONE persistent launch each -- the evaluations are commands to its resident workgroups
        // (Engine::persistent_*).  The per-evaluation inputs of a level's workgroups live in the level's slot of the engine's
        // push block: fine-grained DEVICE memory the CPU writes through the PCIe BAR (write-only for the host): [command 64 B |
        // scale | knots t | knots R | outlier flags].  The kernel of level li + 1 is enqueued while the first evaluation of
        // level li is in flight (its launch cost hides behind that evaluation, and it starts the moment level li's kernel
        // exits).  Without a push block (or for levels too large for the sample-parallel kernel) the inputs stay in pinned
        // host memory / device copies and every evaluation is its own launch.
        struct LevelRun
        {
            mbavo_problem p;
            double *w_inv, *w_kt, *w_kR;
            unsigned char *w_flags;
            bool prepared, launched;
        };
        LevelRun runs[8];
        memset(runs, 0, sizeof(runs));
        const size_t push_bytes = Engine::kPushHeader + 64 + sizeof(double) * 7 * N + (size_t)(maxK > 0 ? maxK : 1) + 64;
        // problem descriptor of level li, its slot's inputs initialised (no outliers, :601); < 0: error code
        auto prepare = [&](int li) -> int {
            LevelRun &R = runs[li];
            if (R.prepared) return 0;
            const int lv = o.num_levels - li - 1;
            const mbavo_level &L = levels[lv];
            const int scale = 1 << lv;
            mbavo_problem &p = R.p;
            memset(&p, 0, sizeof(p));
            p.S = L.S; p.F = F; p.K = L.K; p.P = L.P; p.N = N; p.H = L.H; p.W = L.W;
            p.d_ref_img = L.d_ref_img; p.d_ref_dIxy = L.d_ref_dIxy; p.d_cur_imgs = L.d_cur_imgs;
            p.d_kp_xy = L.d_kp_xy; p.kp_stride = 2; p.d_kp_z = L.d_kp_z; p.d_pattern = L.d_pattern;
            p.d_outlier = d_flags; p.num_bad = 0;
            for (int a = 0; a < 4; ++a) p.intrinsics[a] = o.intrinsics[a] / scale; // :766-770
            p.d_cap_time = d_cap; p.d_exp_time = d_exp; p.t0 = t0; p.dt = dt;
            p.d_knots_t = d_kt; p.d_knots_R = d_kR; p.h_start_idx = start_idx.data(); p.huber_a = o.huber_k;
            // every blur sample must fall on knots that exist (the reference reads past its arrays otherwise,
            // SplineFunctor.h:13-19).  The sample times depend on (cap, exp, S) only: checked here on the host with the
            // kernels' own formula instead of reading the device's status counter back (a synchronous copy per level:
            // 30 us, 12 % of a tracked frame)
            for (int f = 0; f < F; ++f)
                for (int smp = 0; smp < L.S; ++smp)
                {
                    const double ts = h_cap[f] - h_exp[f] * 0.5 + smp * h_exp[f] / (L.S - 1 + 1e-8); // compute_virtual_camera_poses.cu:33
                    int idx;
                    double u;
                    spline_segment(ts, t0, dt, idx, u);
                    if (idx < 0 || idx + k > N) return MBAVO_E_RANGE;
                }
            char *push = (char *)eng.push_block(li, push_bytes);
            if (push)
            {
                double *b = (double *)(push + Engine::kPushHeader);
                R.w_inv = b; R.w_kt = b + 8; R.w_kR = R.w_kt + 3 * N;
                R.w_flags = (unsigned char *)(R.w_kR + 4 * N);
                p.d_knots_t = R.w_kt; p.d_knots_R = R.w_kR; p.d_outlier = R.w_flags;
                memset(R.w_flags, 0, L.K > 0 ? L.K : 1);
                const long long nres = (long long)L.K * F * L.P; // spline_update_step.cpp:116-117, no outliers yet
                *R.w_inv = nres > 0 ? 1.0 / (double)nres : 0.0;
            }
            R.prepared = true;
            return 0;
        };
        // enqueue level li's persistent kernel: 0 = enqueued, 1 = not applicable (now), otherwise an error
        auto launch = [&](int li, bool cached_only) -> int {
            LevelRun &R = runs[li];
            if (R.launched) return 0;
            if (!R.w_inv) return 1;
            PhaseScope ps_level(PhaseTimers::kLevel);
            const int pr = eng.persistent_begin(li, R.p, k, h_pin, d_pc, R.w_inv, cached_only);
            if (pr == 0) R.launched = true;
            return pr;
        };

        // ONE persistent kernel for ALL levels of the call where they fit one list (round 3: a launch and a level set-up cost the
        // host ~14 us each, only partly hidden behind evaluations; trackFrame 0.366 -> see profiles/r03_kfused_experiments.txt 7.):
        // the levels are the problems of one layout, they share the knot buffer of slot 0's push block, every level has its own
        // scale word and flag bytes there, and a command names its level.  mbavo_track_opts.persist_levels = -1 keeps one kernel per level.
        bool joint = false;
        size_t joint_pc_off[8] = {};
        if (o.num_levels > 1 && opt_flag(o.persist_levels, env.persist_levels, true))
        {
            size_t flag_bytes = 0;
            for (int l = 0; l < o.num_levels; ++l) flag_bytes += ((size_t)(levels[l].K > 0 ? levels[l].K : 1) + 63) & ~(size_t)63;
            // [8 scale words | knots t, R (7 N) | SECOND knot area (7 N): the knots of a command's second problem | flag bytes per level]
            char *push = (char *)eng.push_block(0, Engine::kPushHeader + 64 + sizeof(double) * 14 * N + flag_bytes + 64);
            if (push)
            {
                double *b = (double *)(push + Engine::kPushHeader);
                mbavo_problem lp[8];
                unsigned char *fl = (unsigned char *)(b + 8 + 14 * N);
                size_t pc_off = 0;
                bool ok = true;
                for (int li = 0; li < o.num_levels && ok; ++li)
                {
                    if ((rc_ = prepare(li)) != 0) goto done;
                    LevelRun &R = runs[li];
                    const mbavo_level &L = levels[o.num_levels - li - 1];
                    R.w_inv = b + li; R.w_kt = b + 8; R.w_kR = R.w_kt + 3 * N; R.w_flags = fl;
                    R.p.d_knots_t = R.w_kt; R.p.d_knots_R = R.w_kR; R.p.d_outlier = R.w_flags;
                    memset(R.w_flags, 0, L.K > 0 ? L.K : 1);
                    const long long nres = (long long)L.K * F * L.P;
                    *R.w_inv = nres > 0 ? 1.0 / (double)nres : 0.0;
                    fl += ((size_t)(L.K > 0 ? L.K : 1) + 63) & ~(size_t)63;
                    joint_pc_off[li] = pc_off;
                    pc_off += (size_t)F * L.K;
                    lp[li] = R.p;
                }
                PhaseScope ps_level(PhaseTimers::kLevel);
                const int pr = eng.persistent_begin(0, o.num_levels, lp, k, h_pin, d_pc, b, false);
                if (pr == 0) joint = true;
                else if (pr < 0 || pr > 1) { rc_ = pr; goto done; }
                else
                    for (int li = 0; li < o.num_levels; ++li) runs[li].prepared = false; // per-level kernels: their own slots
            }
        }

        // THE NEXT LEVEL'S FIRST EVALUATION RIDES ALONG (round 5; mbavo_track_opts.ride_along, default on, joint persistent kernel
        // only).  A level almost always ends with a rejected candidate (A16: abs_dec goes negative), i.e. at the knots it had BEFORE
        // that candidate -- which are known when the candidate is posted.  So every candidate's command also names the next finer
        // level as its second problem, evaluated with H / g at the CURRENT knots by that level's own, otherwise idle, workgroups.  If
        // the level then ends at those very knots (compared bit for bit), the next level's iteration 0 is already there: one
        // dependent evaluation (~13.5 us) less per pyramid level.  A wasted ride-along costs idle GPU time, never a result.
        const bool ride_along = opt_flag(o.ride_along, env.ride_along, true);
        std::vector<double> ride_kt(3 * (size_t)N), ride_kR(4 * (size_t)N);
        int ride_level = -1;                 // the level (li) the last ride-along evaluated, or -1
        unsigned long long ride_seq = 0;     // ... and the sequence number of its command
        const bool ride_waste = o.ride_along == 2; // (tests: every ride-along counts as taken at other knots -- the wasted path on every level)
        RideAlongStats &ride_stats = RideAlongStats::get();
        // THE ACCEPTED STEP'S EVALUATION AS A RE-SUMMATION (round 5; mbavo_track_opts.resum, default on, persistent kernels).  Two of
        // three accepted steps flag new outliers, so the candidate's H / g (speculation above) are not the accepted point's -- but
        // they differ from it only in which patches' rows count and in the residual scale: the kernel's workgroups still hold the
        // candidate's per-patch sums and add them up again under the new flags (Engine::persistent_post_resum) -- the evaluation's
        // result bit for bit, without pose entries, taps and Jacobians.
        const bool resum = opt_flag(o.resum, env.resum, true);

        for (int li = 0; li < o.num_levels; ++li)
        {
            const int lv = o.num_levels - li - 1; // coarse to fine (:571-575)
            const mbavo_level &L = levels[lv];
            memset(flags, 0, L.K > 0 ? L.K : 1); // :601 (the device copy below, where it is used)
            shadow.assign((size_t)(L.K > 0 ? L.K : 1), 0);
            if (!joint)
            {
                if ((rc_ = prepare(li)) != 0) goto done;
                const int pr = launch(li, false);
                if (pr < 0 || pr > 1) { rc_ = pr; goto done; }
            }
            LevelRun &R = runs[li];
            const bool persistent = joint || R.launched;
            double *const lvl_pin = h_pin + (joint ? (size_t)li * F * E : 0);  // this level's frame blocks / patch costs
            double *const lvl_pc = d_pc + (joint ? joint_pc_off[li] : 0);
            mbavo_problem p = R.p;
            // The outlier count changes with every accepted step; it reaches the kernels through a word (inv_ptr) instead of the
            // problem descriptor, so the engine's cached layout stays valid for the whole level
            int num_bad = 0;
            double *w_inv = persistent ? R.w_inv : h_inv, *w_kt = persistent ? R.w_kt : d_kt, *w_kR = persistent ? R.w_kR : d_kR;
            unsigned char *w_flags = persistent ? R.w_flags : flags;
            double *inv_word = w_inv; // where the kernels read the scale: the level's push slot in persistent mode
            auto set_inv = [&]() {
                const long long nres = (long long)(L.K - num_bad) * F * L.P; // spline_update_step.cpp:116-117
                *inv_word = nres > 0 ? 1.0 / (double)nres : 0.0;
            };
            set_inv();
            if (!persistent)
            {
                p.d_knots_t = d_kt; p.d_knots_R = d_kR; p.d_outlier = d_flags; // per-evaluation launches read the device copy
                TRK_HIP(hipMemsetAsync(d_flags, 0, L.K > 0 ? L.K : 1, st));
            }
            const bool speculate = speculate_env == 1 || (speculate_env < 0 && persistent);
            bool want_prelaunch = persistent && !joint && li + 1 < o.num_levels;
            // one evaluation at the given knots: knots into the pinned buffer, ONE launch for these problem sizes (pose
            // prologue + fused + last-workgroup finalize) whose frame blocks land in pinned host memory (h_pin), then a
            // spin on the kernel's completion word: no copies, no stream synchronisation
            auto evaluate = [&](const double *kt, const double *kR, bool with_h, double *cost, double *Hout = nullptr, double *gout = nullptr,
                                bool with_ride_along = false) -> int {
                if (!Hout) { Hout = H.data(); gout = g.data(); }
                int r;
                {
                    PhaseScope ps(PhaseTimers::kEnqueue);
                    memcpy(w_kt, kt, sizeof(double) * 3 * N);
                    memcpy(w_kR, kR, sizeof(double) * 4 * N);
                    if (!persistent) r = eng.evaluate(1, &p, k, with_h, h_pin, d_pc, nullptr, nullptr, nullptr, h_inv, true);
                    else if (joint && with_ride_along && ride_along && li + 1 < o.num_levels && runs[li + 1].p.K > 0 &&
                             (ride_level < 0 || eng.persistent_second_done(ride_seq)))
                    { // (a new ride-along only behind a finished one: its workgroups must not miss a command that concerns them)
                        memcpy(w_kt + 7 * N, spline.get_knot_data_t(), sizeof(double) * 3 * N); // the CURRENT point, second knot area
                        memcpy(w_kR + 7 * N, spline.get_knot_data_R(), sizeof(double) * 4 * N);
                        memcpy(ride_kt.data(), spline.get_knot_data_t(), sizeof(double) * 3 * N);
                        memcpy(ride_kR.data(), spline.get_knot_data_R(), sizeof(double) * 4 * N);
                        r = eng.persistent_post(0, with_h, li, li + 1);
                        ride_level = li + 1;
                        ride_seq = eng.posted_seq();
                        ride_stats.posts.fetch_add(1, std::memory_order_relaxed);
                    }
                    else r = joint ? eng.persistent_post(0, with_h, li) : eng.persistent_post(li, with_h);
                }
                if (r) return r;
                if (want_prelaunch)
                { // with this level's first evaluation in flight: the next level's kernel goes into the queue (only if its layout
                  // is one of the engine's parked ones -- nothing may be built or uploaded behind a running persistent kernel;
                  // otherwise it is launched when its turn comes)
                    want_prelaunch = false;
                    if (prepare(li + 1) == 0)
                    {
                        const int pr = launch(li + 1, true);
                        if (pr < 0 || pr > 1) return pr;
                    }
                }
                {
                    PhaseScope ps(PhaseTimers::kWait);
                    r = persistent ? eng.persistent_wait() : eng.wait_evaluation();
                }
                if (r) return r;
                PhaseScope ps(PhaseTimers::kMerge);

                merge_blocks_host(F, k, lvl_pin, start_idx.data(), N, cost, with_h ? Hout : nullptr, with_h ? gout : nullptr);
                return 0;
            };
            // the candidate's evaluation (this level's last command, with H / g) summed again under the flags and the scale as they are now
            auto evaluate_again = [&](double *cost) -> int {
                int r = joint ? eng.persistent_post_resum(0, li) : eng.persistent_post_resum(li);
                if (r) return r;
                {
                    PhaseScope ps(PhaseTimers::kWait);
                    r = eng.persistent_wait();
                }
                if (r) return r;
                PhaseScope ps(PhaseTimers::kMerge);
                merge_blocks_host(F, k, lvl_pin, start_idx.data(), N, cost, H.data(), g.data());
                return 0;
            };
            auto record = [&](int iter, int kind, double cc, double model, double q) {
                if (trace && ntrace < trace_cap)
                {
                    mbavo_trace_rec &r = trace[ntrace];
                    r.level = lv; r.iter = iter; r.kind = kind; r.num_outliers = num_bad;
                    r.radius = lm.get_radius(); r.eval_cost = eval_cost; r.candidate_cost = cc;
                    r.model_change = model; r.quality = q;
                }
                ++ntrace;
            };

            // iteration 0 -- already evaluated by the previous level's last ride-along if that was taken at these very knots
            if (joint && ride_level == li && !ride_waste && memcmp(ride_kt.data(), spline.get_knot_data_t(), sizeof(double) * 3 * N) == 0 &&
                memcmp(ride_kR.data(), spline.get_knot_data_R(), sizeof(double) * 4 * N) == 0)
            {
                {
                    PhaseScope ps(PhaseTimers::kWait);
                    if ((rc_ = eng.persistent_wait_second(ride_seq))) goto done;
                }
                PhaseScope ps(PhaseTimers::kMerge);
                merge_blocks_host(F, k, lvl_pin, start_idx.data(), N, &eval_cost, H.data(), g.data());
                ride_stats.hits.fetch_add(1, std::memory_order_relaxed);
            }
            else
            {
                // A ride-along taken at other knots may still be running on this level's workgroups.  Its tiles share the level's
                // ticket counters, tile partials and frame blocks with the command posted next, and the workgroups take commands
                // independently -- one could finish its wasted tile and bump the ticket for the NEW command while a sibling is still
                // on the old one (with two tiles: the first workgroup would count 0->1, 1->2 and sum its new partial with the
                // sibling's stale one).  So the wasted ride-along is waited out (<= one evaluation, ~10 us, on levels that end on an
                // accepted candidate only) before this level's first command goes out.
                if (joint && ride_level == li)
                {
                    PhaseScope ps(PhaseTimers::kWait);
                    if ((rc_ = eng.persistent_wait_second(ride_seq))) goto done;
                    ride_stats.waits.fetch_add(1, std::memory_order_relaxed);
                }
                if ((rc_ = evaluate(spline.get_knot_data_t(), spline.get_knot_data_R(), true, &eval_cost))) goto done;
            }
            if (ride_level == li) ride_level = -1;
            lm.reset();
            evaluator.reset(eval_cost);
            record(0, 0, 0.0, 0.0, 0.0);

            int iter = 0;
            double abs_dec = 1e10;
            for (;;)
            {
                ++iter; // finalizeIterationAndCheckIfMinimizerCanContinue (:910-924)
                if (iter > o.max_num_iterations) break;
                if (abs_dec < o.min_abs_cost_decrease) break;

                // computeTrustRegionStep (:799-831)
                PhaseScope ps_solve(PhaseTimers::kSolve);
                const double iradius = 1. / lm.get_radius();
                for (int i = 0; i < n; ++i) H[(size_t)i * n + i] += H[(size_t)i * n + i] * iradius;
                if (solve_normal_equation_host(H.data(), g.data(), n, o.solver_type, step.data(), fast_ratio) < 0) { rc_ = MBAVO_E_ARG; goto done; }
                double gx = 0.0, xHx = 0.0;
                for (int i = 0; i < n; ++i) gx += g[i] * step[i];
                for (int r = 0; r < n; ++r)
                {
                    double a = 0.0;
                    for (int c = 0; c < n; ++c) a += H[(size_t)c * n + r] * step[c];
                    xHx += step[r] * a;
                }
                const double model = -(gx + 0.5 * xHx);
                ps_solve.stop();
                if (model < 0) { lm.step_rejected(); record(iter, 3, 0.0, model, 0.0); continue; } // handleInvalidStep

                // computeCandidatePointAndEvaluateCost (:833-883)
                spline.Plus_t(step.data(), cand_t.data());
                spline.Plus_R(step.data() + 3 * N, cand_R.data());
                double cand_cost = 0.0;
                if ((rc_ = evaluate(cand_t.data(), cand_R.data(), speculate, &cand_cost, Hs.data(), gs.data(), true))) goto done;

                abs_dec = eval_cost - cand_cost;
                const double quality = evaluator.StepQuality(cand_cost, model);
                if (quality > o.min_step_quality && cand_cost < eval_cost)
                { // isStepSuccessful (:890-894) -> detectOutliers + handleSuccessfulStep (:896-903)
                    bool same_inputs = false;
                    {
                        PhaseScope ps(PhaseTimers::kOutliers);
                        const int bad_before = num_bad;
                        int newly = 0;
                        num_bad = detect_outliers(lvl_pc, L.K, o.max_chi_square_error, w_flags, shadow.data(), &newly); // frame 0's costs, just written
                        same_inputs = newly == 0 && num_bad == bad_before; // flags and residual scale as the candidate's evaluation saw them
                        if (PhaseTimers::get().on) ++PhaseTimers::get().calls[!same_inputs ? PhaseTimers::kFlagsChanged : PhaseTimers::kFlagsSame];
                        set_inv();
                        if (!persistent) TRK_HIP(hipMemcpyAsync(d_flags, w_flags, L.K, hipMemcpyHostToDevice, st));
                    }
                    spline.InvalidParameter(cand_t.data(), cand_R.data());
                    if (speculate && same_inputs)
                    { // the candidate's evaluation IS the accepted point's
                        H.swap(Hs);
                        g.swap(gs);
                        eval_cost = cand_cost;
                    }
                    else if (speculate && resum && persistent && (joint ? eng.persistent_resum_ok(0, li) : eng.persistent_resum_ok(li)))
                    {
                        if ((rc_ = evaluate_again(&eval_cost))) goto done;
                    }
                    else if ((rc_ = evaluate(spline.get_knot_data_t(), spline.get_knot_data_R(), true, &eval_cost))) goto done;
                    lm.step_accepted(quality);
                    evaluator.StepAccepted(eval_cost, model);
                    record(iter, 1, cand_cost, model, quality);
                    continue;
                }
                lm.step_rejected(); // handleUnsuccessfulStep
                record(iter, 2, cand_cost, model, quality);
            }
            if (!joint) (void)eng.persistent_end(li); // the level's resident workgroups exit; the next level's kernel is already queued behind them
        }
        }
        memcpy(knots_t, spline.get_knot_data_t(), sizeof(double) * 3 * N);
        memcpy(knots_R, spline.get_knot_data_R(), sizeof(double) * 4 * N);
        if (final_cost) *final_cost = eval_cost;
    done:
        (void)eng.persistent_end_all();
        return rc_ ? (rc_ > 0 ? -1000 - rc_ : rc_) : ntrace;
    }
} // namespace mbavo
