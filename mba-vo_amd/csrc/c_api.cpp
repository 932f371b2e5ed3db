// c_api.cpp -- the extern "C" boundary declared in include/mbavo.h.
// Thin POD wrappers over the C++ ba_tracker API (ba_tracker.h), the fused engine
// (engine.h) and the host control flow (host_math.h, tracker.h).
#include "../../include/mbavo.h"
#include "ba_tracker.h"
#include "engine.h"
#include "host_math.h"
#include "se3_math.h"
#include "tracker.h"
#include "timing.h"
#include "vo_frontend.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

using namespace SLAM;

// One helper thread per extra group of mbavo_lm_batch, kept between calls: a job is handed over through a flag the helper
// spins on for a short while after its last job (back-to-back calls: no wake-up latency) before it blocks on a condition variable
// (creating a thread per call cost ~60 us of a 1.2 ms call).
struct GroupWorker
{
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    std::atomic<int> state{0}; // 0 idle, 1 job posted, 2 job done, 3 exit
    GroupWorker()
    {
        th = std::thread([this]() {
            for (;;)
            {
                int s = state.load(std::memory_order_acquire);
                if (s != 1 && s != 3)
                { // spin ~200 us, then sleep
                    const auto t0 = std::chrono::steady_clock::now();
                    while ((s = state.load(std::memory_order_acquire)) != 1 && s != 3)
                    {
                        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200))
                        {
                            std::unique_lock<std::mutex> lk(m);
                            cv.wait(lk, [&]() { const int v = state.load(std::memory_order_acquire); return v == 1 || v == 3; });
                            s = state.load(std::memory_order_acquire);
                            break;
                        }
                    }
                }
                if (s == 3) return;
                job();
                state.store(2, std::memory_order_release);
            }
        });
    }
    void post(std::function<void()> f)
    {
        job = std::move(f);
        {
            std::lock_guard<std::mutex> lk(m);
            state.store(1, std::memory_order_release);
        }
        cv.notify_one();
    }
    void wait()
    {
        while (state.load(std::memory_order_acquire) != 2) std::this_thread::yield();
        state.store(0, std::memory_order_release);
    }
    ~GroupWorker()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            state.store(3, std::memory_order_release);
        }
        cv.notify_one();
        if (th.joinable()) th.join();
    }
};

struct mbavo_ctx
{
    mbavo::Engine *engine;
    // mbavo_lm_batch on big batches: a second engine on a stream of its own (created at first use), so that two halves of the
    // batch run as independent chains the GPU interleaves
    std::vector<mbavo::Engine *> extra_engines;
    std::vector<hipStream_t> extra_streams;
    std::vector<GroupWorker *> workers;
    hipEvent_t fork = nullptr;
    mbavo::Engine::Options engine_opts; // (new group engines start with the context's options)
};
struct mbavo_vo
{
    VO::BlurAwareDirectTracker impl;
    std::vector<std::vector<int>> patterns; // owns the copies the options point to
    mbavo_vo(mbavo::Engine &e, const VO::BlurAwareDirectTrackerOptions &o) : impl(e, o) {}
};
struct mbavo_lm
{
    VO::LevenbergMarquardtStrategy impl;
};
struct mbavo_tr
{
    VO::TrustRegionStepEvaluator impl;
    explicit mbavo_tr(int m) : impl(m) {}
};

static int packed_len(int k)
{
    const int nd = 6 * k + 1;
    return nd * (nd + 1) / 2;
}

extern "C"
{
    const char *mbavo_version(void) { return "mbavo-mi355x 0.2 (gfx950)"; }
    int mbavo_abi_version(void) { return MBAVO_ABI_VERSION; }
    int mbavo_sizeof(int which)
    { // what the library was compiled with, for a binding to check its own mirror of the structs against
        switch (which)
        {
        case 0: return (int)sizeof(mbavo_problem);
        case 1: return (int)sizeof(mbavo_track_opts);
        case 2: return (int)sizeof(mbavo_lm_batch_opts);
        case 3: return (int)sizeof(mbavo_vo_options);
        case 4: return (int)sizeof(mbavo_engine_opts);
        case 5: return (int)sizeof(mbavo_vo_state);
        case 6: return (int)sizeof(mbavo_trace_rec);
        case 7: return (int)sizeof(mbavo_level);
        case 8: return (int)sizeof(mbavo_lm_batch_result);
        case 9: return (int)sizeof(mbavo_vo_info);
        default: return MBAVO_E_ARG;
        }
    }

    int mbavo_packed_len(int k) { return packed_len(k); }

    int mbavo_create(mbavo_ctx **out, int device_id)
    {
        if (!out) return MBAVO_E_ARG;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n)
        {
            fprintf(stderr, "mbavo: no usable HIP device (count=%d, requested %d); there is no CPU fallback\n", n, device_id);
            return MBAVO_E_NODEVICE;
        }
        hipError_t e = hipSetDevice(device_id);
        if (e != hipSuccess) return (int)e;
        mbavo_ctx *c = new (std::nothrow) mbavo_ctx;
        if (!c) return MBAVO_E_ARG;
        c->engine = new mbavo::Engine(device_id);
        *out = c;
        return 0;
    }

    int mbavo_destroy(mbavo_ctx *ctx)
    {
        if (!ctx) return MBAVO_E_ARG;
        delete ctx->engine;
        for (GroupWorker *w : ctx->workers) delete w;
        for (mbavo::Engine *e : ctx->extra_engines) delete e;
        for (hipStream_t s : ctx->extra_streams) (void)hipStreamDestroy(s);
        if (ctx->fork) (void)hipEventDestroy(ctx->fork);
        delete ctx;
        return 0;
    }

    int mbavo_set_stream(mbavo_ctx *ctx, void *s)
    {
        if (!ctx) return MBAVO_E_ARG;
        ctx->engine->set_stream((hipStream_t)s);
        return 0;
    }

    int mbavo_set_engine_opts(mbavo_ctx *ctx, const mbavo_engine_opts *o)
    {
        if (!ctx) return MBAVO_E_ARG;
        mbavo::Engine::Options e;
        if (o)
        {
            e.sample_parallel = o->sample_parallel; e.single_launch = o->single_launch; e.fused_pose = o->fused_pose;
            e.fused_pose_max_samples = o->fused_pose_max_samples; e.persistent = o->persistent; e.prelaunch = o->prelaunch;
            e.tiles_per_cu = o->tiles_per_cu; e.min_tile_pixels = o->min_tile_pixels; e.sp_max_slot_tiles = o->sp_max_slot_tiles;
        }
        ctx->engine->set_options(e);
        for (mbavo::Engine *x : ctx->extra_engines) x->set_options(e);
        ctx->engine_opts = e;
        return 0;
    }

    int mbavo_get_engine_opts(mbavo_ctx *ctx, mbavo_engine_opts *o)
    {
        if (!ctx || !o) return MBAVO_E_ARG;
        const mbavo::Engine::Options &e = ctx->engine->options();
        memset(o, 0, sizeof(*o));
        o->sample_parallel = e.sample_parallel; o->single_launch = e.single_launch; o->fused_pose = e.fused_pose;
        o->fused_pose_max_samples = e.fused_pose_max_samples; o->persistent = e.persistent; o->prelaunch = e.prelaunch;
        o->tiles_per_cu = e.tiles_per_cu; o->min_tile_pixels = e.min_tile_pixels; o->sp_max_slot_tiles = e.sp_max_slot_tiles;
        return 0;
    }

    int mbavo_eval_batch(mbavo_ctx *ctx, int B, const mbavo_problem *probs, int k, int with_hessian,
                         double *d_frame_blocks, double *d_patch_cost, double *d_valid)
    {
        if (!ctx) return MBAVO_E_ARG;
        return ctx->engine->evaluate(B, probs, k, with_hessian != 0, d_frame_blocks, d_patch_cost, d_valid, nullptr);
    }

    int mbavo_eval_batch_merged(mbavo_ctx *ctx, int B, const mbavo_problem *probs, int k, double *d_frame_blocks, double *d_systems,
                                double *d_patch_cost, double *d_valid)
    {
        if (!ctx || !d_systems) return MBAVO_E_ARG;
        ctx->engine->set_merge_target(d_systems);
        const int rc = ctx->engine->evaluate(B, probs, k, true, d_frame_blocks, d_patch_cost, d_valid, nullptr);
        ctx->engine->set_merge_target(nullptr); // (an evaluate() that failed before it took the target must not leave it armed)
        return rc;
    }

    int mbavo_eval(mbavo_ctx *ctx, const mbavo_problem *p, int k, double *h_cost, double *h_H, double *h_g,
                   double *d_patch_blocks)
    {
        if (!ctx || !p || !h_cost || (h_H && !h_g) || !p->h_start_idx) return MBAVO_E_ARG;
        mbavo::Engine &eng = *ctx->engine;
        const int E = packed_len(k);
        const size_t n = (size_t)p->F * E;
        double *h_fb = eng.host_frame_blocks(n); // pinned, device-visible: the finalize kernel writes it directly
        if (!h_fb) return (int)hipErrorOutOfMemory;
        int rc = eng.evaluate(1, p, k, h_H != nullptr, h_fb, nullptr, nullptr, d_patch_blocks);
        if (rc) return rc;
        hipError_t e = hipStreamSynchronize(eng.stream());
        if (e != hipSuccess) return (int)e;
        if (eng.fetch_status() != 0) return MBAVO_E_RANGE;
        mbavo::merge_blocks_host(p->F, k, h_fb, p->h_start_idx, p->N, h_cost, h_H, h_g);
        return 0;
    }

    // ---- the five launchers (the C++ versions abort on HIP errors like a device assert would)
    int mbavo_compute_virtual_camera_poses(int S, int F, const double *d_cap, const double *d_exp, int k, double t0,
                                           double dt, const double *d_kt, const double *d_kR, double *d_poses,
                                           double *d_J_t, double *d_J_R)
    {
        if ((k != 2 && k != 4) || !d_cap || !d_exp || !d_kt || !d_kR || !d_poses || ((d_J_t == nullptr) != (d_J_R == nullptr)))
            return MBAVO_E_ARG;
        VO::compute_virtual_camera_poses(S, F, d_cap, d_exp, k, t0, dt, d_kt, d_kR, d_poses, d_J_t, d_J_R);
        return 0;
    }

    static void fill(Core::VectorX<double, 4> &i4, Core::VectorX<int, 2> &hw, const double intr[4], const int HW[2])
    {
        i4.nDim = 4; hw.nDim = 2;
        for (int i = 0; i < 4; ++i) i4.values[i] = intr[i];
        hw.values[0] = HW[0]; hw.values[1] = HW[1];
    }

    int mbavo_compute_local_patches_xy(int S, int F, const double *d_poses, const void *d_kps, const double *d_z, int K,
                                       const double intr[4], const int HW[2], void *d_out)
    {
        if (!d_poses || !d_kps || !d_z || !intr || !HW || !d_out) return MBAVO_E_ARG;
        Core::VectorX<double, 4> i4; Core::VectorX<int, 2> hw;
        fill(i4, hw, intr, HW);
        VO::compute_local_patches_xy(S, F, d_poses, (const Core::Vector2d *)d_kps, d_z, K, i4, hw, (Core::Vector2d *)d_out);
        return 0;
    }

    int mbavo_compute_pixel_jacobian_residual(const unsigned char *d_I_ref, const float *d_dIxy,
                                              const unsigned char *const *d_I_cur, int S, int F, const double *d_poses,
                                              int k, const double *d_J_t, const double *d_J_R, const void *d_centres,
                                              const double *d_z, int K, const int *d_pattern, int P, const double intr[4],
                                              const int HW[2], double *d_res, double *d_jac)
    {
        if ((k != 2 && k != 4) || !d_I_ref || !d_I_cur || !d_poses || !d_centres || !d_z || !d_pattern || !d_res ||
            (d_jac && (!d_J_t || !d_J_R || !d_dIxy)))
            return MBAVO_E_ARG;
        Core::VectorX<double, 4> i4; Core::VectorX<int, 2> hw;
        fill(i4, hw, intr, HW);
        VO::compute_pixel_jacobian_residual(d_I_ref, d_dIxy, d_I_cur, S, F, d_poses, k, d_J_t, d_J_R,
                                            (const Core::Vector2d *)d_centres, d_z, K, d_pattern, P, i4, hw, nullptr,
                                            d_res, d_jac);
        return 0;
    }

    int mbavo_compute_patch_cost_gradient_hessian(int F, int K, int P, int k, const double *d_res, const double *d_jac,
                                                  double huber_a, double inv, double *d_blocks)
    {
        if ((k != 2 && k != 4) || !d_res || !d_blocks) return MBAVO_E_ARG;
        VO::compute_patch_cost_gradient_hessian(F, K, P, k, d_res, d_jac, huber_a, inv, d_blocks);
        return 0;
    }

    int mbavo_compute_frame_cost_gradient_hessian(int F, int K, int k, const double *d_blocks, int eval_gh,
                                                  const unsigned char *d_flags, double *d_frame_blocks)
    {
        if ((k != 2 && k != 4) || !d_blocks || !d_frame_blocks) return MBAVO_E_ARG;
        VO::compute_frame_cost_gradient_hessian(F, K, k, d_blocks, eval_gh != 0, d_flags, d_frame_blocks);
        return 0;
    }

    int mbavo_merge_hessian_gradient_cost(int F, int k, const double *d_fb, const int *h_start, int N, double *h_cost,
                                          double *h_H, double *h_g)
    {
        if (!d_fb || !h_start || !h_cost || (h_H && !h_g)) return MBAVO_E_ARG;
        VO::merge_hessian_gradient_cost(F, k, d_fb, h_start, N, h_cost, h_H, h_g);
        return 0;
    }

    int mbavo_merge_host(int F, int k, const double *h_fb, const int *h_start, int N, double *h_cost, double *h_H,
                         double *h_g)
    {
        if (!h_fb || !h_start || !h_cost || (h_H && !h_g)) return MBAVO_E_ARG;
        mbavo::merge_blocks_host(F, k, h_fb, h_start, N, h_cost, h_H, h_g);
        return 0;
    }

    int mbavo_solve_normal_equation(const double *A, const double *b, int n, int type, double *x)
    {
        if (!A || !b || !x || n < 1) return MBAVO_E_ARG;
        return mbavo::solve_normal_equation_host(A, b, n, type, x) < 0 ? MBAVO_E_ARG : 0;
    }

    // ---- LM / trust region
    mbavo_lm *mbavo_lm_new(void) { return new mbavo_lm(); }
    void mbavo_lm_delete(mbavo_lm *p) { delete p; }
    void mbavo_lm_reset(mbavo_lm *p) { p->impl.reset(); }
    void mbavo_lm_step_accepted(mbavo_lm *p, double q) { p->impl.step_accepted(q); }
    void mbavo_lm_step_rejected(mbavo_lm *p) { p->impl.step_rejected(); }
    double mbavo_lm_get_radius(mbavo_lm *p) { return p->impl.get_radius(); }
    mbavo_tr *mbavo_tr_new(int m) { return new mbavo_tr(m); }
    void mbavo_tr_delete(mbavo_tr *p) { delete p; }
    void mbavo_tr_reset(mbavo_tr *p, double c) { p->impl.reset(c); }
    double mbavo_tr_step_quality(mbavo_tr *p, double c, double m) { return p->impl.StepQuality(c, m); }
    void mbavo_tr_step_accepted(mbavo_tr *p, double c, double m) { p->impl.StepAccepted(c, m); }

    // ---- spline
    int mbavo_spline_get_pose(int k, double t0, double dt, const double *kt, const double *kR, int N, double t,
                              double t_out[3], double q_out[4], double *J_t, double *J_R)
    {
        if ((k != 2 && k != 4) || !kt || !kR || !t_out || !q_out) return MBAVO_E_ARG;
        Core::SplineSE3 s(t0, dt);
        s.setSplineDegK(k);
        for (int i = 0; i < N; ++i) s.InsertControlKnot(kR + 4 * i, kt + 3 * i);
        return s.GetPose(t, q_out, t_out, J_R, J_t) ? 0 : MBAVO_E_RANGE;
    }

    int mbavo_spline_plus(const double *kt, const double *kR, int N, const double *step, double *cand_t, double *cand_R)
    {
        if (!kt || !kR || !step || !cand_t || !cand_R) return MBAVO_E_ARG;
        Core::SplineSE3 s;
        for (int i = 0; i < N; ++i) s.InsertControlKnot(kR + 4 * i, kt + 3 * i);
        s.Plus_t(step, cand_t);
        s.Plus_R(step + 3 * N, cand_R);
        return 0;
    }

    int mbavo_segment_start_index(double t, double t0, double dt)
    {
        int idx;
        double u;
        mbavo::spline_segment(t, t0, dt, idx, u);
        return idx;
    }

    int mbavo_optimize_trajectory(mbavo_ctx *ctx, const mbavo_track_opts *o, const mbavo_level *levels, int F,
                                  const double *h_cap, const double *h_exp, double t0, double dt, double *kt, double *kR,
                                  int N, int *start_idx_out, double *final_cost, mbavo_trace_rec *trace, int cap)
    {
        if (!ctx || !o || !levels || !h_cap || !h_exp || !kt || !kR) return MBAVO_E_ARG;
        return mbavo::optimize_trajectory(*ctx->engine, *o, levels, F, h_cap, h_exp, t0, dt, kt, kR, N, start_idx_out,
                                          final_cost, trace, cap);
    }

    int mbavo_lm_batch(mbavo_ctx *ctx, int B, const mbavo_problem *probs, const mbavo_lm_batch_opts *o,
                       mbavo_lm_batch_result *results, mbavo_trace_rec *trace, int trace_cap)
    {
        if (!ctx || !probs || !o || trace_cap < 0) return MBAVO_E_ARG;
        // Big batches as TWO independent groups (round 4): the pairs share nothing, so the halves run
        // the whole loop side by side -- the second on its own engine, stream and host thread -- and the GPU fills one group's
        // solve / decide launches (one latency-bound workgroup per problem, ~35 us of a 512-pair slot) and the ramps and tails of
        // its passes with the other group's kernels.  Measured (rendered 640x480 pairs, us per LM round, one / two / four groups):
        // 512 pairs 159-163 / 144-145 / 145-149 (packed keyframes 143 / 132 / 132-136), 256 pairs 109 / 115, 128 pairs 86 / 102,
        // 64 pairs 74 / 96 -- below ~400 problems a group's passes no longer fill the machine and every phase is latency-bound
        // either way, so smaller batches stay one group.  Two groups of a 512-pair batch keep the single group's tiling (one
        // tile per pair): identical records.  mbavo_lm_batch_opts.groups = n overrides (1 .. 8).
        const int genv = mbavo::read_env_overrides().lm_groups;
        int groups = genv != mbavo::kEnvUnset ? genv : (o->groups > 0 ? o->groups : (B >= 384 ? 2 : 1));
        if (groups > 8) groups = 8;
        if (groups > B) groups = B;
        if (groups < 2)
        {
            try { return mbavo::lm_batch(*ctx->engine, B, probs, *o, results, trace, trace_cap); }
            catch (...) { (void)hipStreamSynchronize(ctx->engine->stream()); return MBAVO_E_ARG; }
        }
        const int dev = ctx->engine->device();
        if (hipSetDevice(dev) != hipSuccess) return MBAVO_E_NODEVICE;
        if (!ctx->fork && hipEventCreateWithFlags(&ctx->fork, hipEventDisableTiming) != hipSuccess) return MBAVO_E_NODEVICE;
        while ((int)ctx->extra_engines.size() < groups - 1)
        {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return MBAVO_E_NODEVICE;
            ctx->extra_streams.push_back(s);
            ctx->extra_engines.push_back(new mbavo::Engine(dev));
            ctx->extra_engines.back()->set_stream(s);
            ctx->extra_engines.back()->set_options(ctx->engine_opts);
            ctx->workers.push_back(new GroupWorker());
        }
        // whatever the caller enqueued on the context's stream (knot resets, uploads) comes before the other groups' work too
        if (hipEventRecord(ctx->fork, ctx->engine->stream()) != hipSuccess) return MBAVO_E_NODEVICE;
        // What the whole batch decides is decided ONCE and handed to every group: the largest knot count (array strides,
        // the solve kernel's form -- wide workgroup or one wave -- and its LDS) and the largest sample count.  A group's own list is
        // still tiled for itself, so the grouping of the partial sums -- and with it the last bits of costs and knots -- can differ
        // from the single-group run (as with `retile`); the discrete records do not (tests: mixed N, B just under 2 x CUs).
        mbavo::LmBatchShared shared;
        for (int b = 0; b < B; ++b)
        {
            shared.max_N = probs[b].N > shared.max_N ? probs[b].N : shared.max_N;
            shared.max_S = probs[b].S > shared.max_S ? probs[b].S : shared.max_S;
        }
        // shared state of the groups lives on the heap and the workers are ALWAYS waited for, also when this thread's own group throws
        // (std::bad_alloc from a vector): no exception crosses the C boundary and no worker writes into a dead frame
        struct Rcs { std::vector<int> v; };
        auto rcs = std::make_shared<Rcs>();
        rcs->v.assign(groups, 0);
        std::vector<int> posted;
        auto first_of = [B, groups](int g) { return (int)((long long)B * g / groups); }; // contiguous, near-equal shares
        const mbavo_lm_batch_opts opts = *o;
        for (int g = 1; g < groups; ++g)
        {
            if (hipStreamWaitEvent(ctx->extra_streams[g - 1], ctx->fork, 0) != hipSuccess) { rcs->v[g] = MBAVO_E_NODEVICE; continue; }
            mbavo::Engine *eng = ctx->extra_engines[g - 1];
            ctx->workers[g - 1]->post([=]() {
                const int b0 = first_of(g), n = first_of(g + 1) - b0;
                int rc;
                try
                {
                    rc = mbavo::lm_batch(*eng, n, probs + b0, opts, results ? results + b0 : nullptr,
                                         trace ? trace + (size_t)b0 * trace_cap : nullptr, trace_cap, &shared);
                }
                catch (...) { rc = MBAVO_E_ARG; (void)hipStreamSynchronize(eng->stream()); }
                rcs->v[g] = rc;
            });
            posted.push_back(g - 1);
        }
        try
        {
            rcs->v[0] = mbavo::lm_batch(*ctx->engine, first_of(1), probs, opts, results, trace, trace_cap, &shared);
        }
        catch (...) { rcs->v[0] = MBAVO_E_ARG; (void)hipStreamSynchronize(ctx->engine->stream()); }
        for (int w : posted) ctx->workers[w]->wait(); // (every call returns synchronised with its streams, failed or not)
        for (int g = 0; g < groups; ++g)
            if (rcs->v[g] != 0) return rcs->v[g];
        return 0;
    }

    // ---- trackFrame front end
    int mbavo_detect_semidense(mbavo_ctx *ctx, const unsigned char *d_img, int H, int W, int level, int H0, int W0, int cell_H,
                               int cell_W, float thr, const float *d_depth_z, double *d_kp_xy, double *d_kp_z, int cap, int *h_count)
    {
        if (!ctx) return MBAVO_E_ARG;
        return mbavo::detect_semidense(*ctx->engine, d_img, H, W, level, H0, W0, cell_H, cell_W, thr, d_depth_z, d_kp_xy, d_kp_z,
                                       cap, h_count);
    }

    int mbavo_pyramid_levels_u8(mbavo_ctx *ctx, unsigned char *const *h_level_ptrs, int H0, int W0, int num_levels)
    {
        if (!ctx || !h_level_ptrs || num_levels < 1 || num_levels > 8) return MBAVO_E_ARG;
        for (int l = 0; l < num_levels; ++l)
            if (!h_level_ptrs[l]) return MBAVO_E_ARG;
        return mbavo::pyramid_enqueue(*ctx->engine, h_level_ptrs, H0, W0, num_levels);
    }

    int mbavo_se3_exp(const double a[6], double pose[7])
    {
        if (!a || !pose) return MBAVO_E_ARG;
        memcpy(pose, Core::Transformation::exp(a).getData(), sizeof(double) * 7);
        return 0;
    }
    int mbavo_se3_log(const double pose[7], double out[6])
    {
        if (!pose || !out) return MBAVO_E_ARG;
        Core::Transformation::log(Core::Transformation(pose + 3, pose), out);
        return 0;
    }
    int mbavo_transform_mul(const double A[7], const double B[7], double out[7])
    {
        if (!A || !B || !out) return MBAVO_E_ARG;
        const Core::Transformation r = Core::Transformation(A + 3, A) * Core::Transformation(B + 3, B);
        memcpy(out, r.getData(), sizeof(double) * 7);
        return 0;
    }
    int mbavo_transform_inverse(const double A[7], double out[7])
    {
        if (!A || !out) return MBAVO_E_ARG;
        memcpy(out, Core::Transformation(A + 3, A).inverse().getData(), sizeof(double) * 7);
        return 0;
    }
    int mbavo_spline_transform_to(int k, double t0, double dt, double *kt, double *kR, int N, double t, const double q[4],
                                  const double p[3])
    {
        if ((k != 2 && k != 4) || !kt || !kR || !q || !p || N < k) return MBAVO_E_ARG;
        Core::SplineSE3 s(t0, dt);
        s.setSplineDegK(k);
        for (int i = 0; i < N; ++i) s.InsertControlKnot(kR + 4 * i, kt + 3 * i);
        if (!s.TransformTo(t, q, p)) return MBAVO_E_RANGE;
        memcpy(kt, s.get_knot_data_t(), sizeof(double) * 3 * N);
        memcpy(kR, s.get_knot_data_R(), sizeof(double) * 4 * N);
        return 0;
    }

    int mbavo_vo_create(mbavo_ctx *ctx, const mbavo_vo_options *o, mbavo_vo **out)
    {
        if (!ctx || !o || !out || o->num_pyramid_levels < 1 || o->num_pyramid_levels > 8) return MBAVO_E_ARG;
        VO::BlurAwareDirectTrackerOptions v;
        for (int i = 0; i < 4; ++i) v.intrinsics[i] = o->intrinsics[i];
        v.im_size_HW[0] = o->H; v.im_size_HW[1] = o->W;
        v.num_pyramid_levels = o->num_pyramid_levels;
        std::vector<std::vector<int>> pats(8);
        for (int l = 0; l < 8; ++l)
        {
            v.num_virtual_poses_per_frame[l] = o->num_virtual_poses_per_frame[l];
            v.patch_size[l] = o->patch_size[l];
            v.local_patch_pattern_xy[l] = nullptr;
            if (l < o->num_pyramid_levels)
            {
                if (!o->local_patch_pattern_xy[l] || o->patch_size[l] < 1) return MBAVO_E_ARG;
                pats[l].assign(o->local_patch_pattern_xy[l], o->local_patch_pattern_xy[l] + 2 * o->patch_size[l]);
                v.local_patch_pattern_xy[l] = pats[l].data();
            }
        }
        v.huber_k = o->huber_k;
        v.max_consecutive_nonmonotonic_steps = o->max_consecutive_nonmonotonic_steps;
        v.max_num_iterations = o->max_num_iterations;
        v.min_step_quality = o->min_step_quality; v.min_abs_cost_decrease = o->min_abs_cost_decrease;
        v.solver_type = o->solver_type; v.spline_deg_k = o->spline_deg_k;
        v.dt_frame = o->dt_frame; v.dt_ctrl_knot = o->dt_ctrl_knot; v.max_chi_square_error = o->max_chi_square_error;
        v.keyframe_max_flow_mag0 = o->keyframe_max_flow_mag0; v.keyframe_max_flow_mag1 = o->keyframe_max_flow_mag1;
        v.keyframe_max_flow_mag2 = o->keyframe_max_flow_mag2; v.keyframe_max_blur_kernel_mag = o->keyframe_max_blur_kernel_mag;
        v.score_threshold = o->score_threshold;
        v.grid_selection_cell_H = o->grid_selection_cell_H; v.grid_selection_cell_W = o->grid_selection_cell_W;
        v.fast_solve_ratio = o->fast_solve_ratio; v.speculate = o->speculate; v.persist_levels = o->persist_levels;
        v.keyframe_levels_at_once = o->keyframe_levels_at_once; v.speculate_keyframe = o->speculate_keyframe; v.ride_along = o->ride_along; v.resum = o->resum;
        if (v.spline_deg_k != 2 && v.spline_deg_k != 4) return MBAVO_E_ARG;
        mbavo_vo *h = new (std::nothrow) mbavo_vo(*ctx->engine, v);
        if (!h) return MBAVO_E_ARG;
        const int st = h->impl.status();
        if (st != 0) { delete h; return st; }
        h->patterns.swap(pats);
        *out = h;
        return 0;
    }

    int mbavo_vo_destroy(mbavo_vo *vo)
    {
        delete vo;
        return 0;
    }

    int mbavo_vo_set_spline(mbavo_vo *vo, double t0, double dt, int N, const double *kt, const double *kR)
    {
        if (!vo || N < 0 || N > 16 || (N > 0 && (!kt || !kR))) return MBAVO_E_ARG;
        Core::SplineSE3 *s = vo->impl.getSplineTrajectory();
        s->Clear();
        s->setStartTime(t0); s->setSamplingFreq(dt); s->setSplineDegK(vo->impl.getOptions().spline_deg_k);
        for (int i = 0; i < N; ++i) s->InsertControlKnot(kR + 4 * i, kt + 3 * i);
        return 0;
    }

    int mbavo_vo_get_spline(mbavo_vo *vo, double *t0, double *dt, int *N, double *kt, double *kR)
    {
        if (!vo) return MBAVO_E_ARG;
        Core::SplineSE3 *s = vo->impl.getSplineTrajectory();
        const int n = (int)s->get_num_knots();
        if (t0) *t0 = s->getStartTime();
        if (dt) *dt = s->getSamplingFreq();
        if (N) *N = n;
        if (kt && n) memcpy(kt, s->get_knot_data_t(), sizeof(double) * 3 * n);
        if (kR && n) memcpy(kR, s->get_knot_data_R(), sizeof(double) * 4 * n);
        return 0;
    }

    int mbavo_vo_last_trace(mbavo_vo *vo, mbavo_trace_rec *trace, int trace_cap)
    {
        if (!vo || trace_cap < 0 || (trace_cap > 0 && !trace)) return MBAVO_E_ARG;
        return vo->impl.lastTrace(trace, trace_cap);
    }

    static_assert(sizeof(mbavo_vo_state) == sizeof(VO::TrackerState), "mbavo_vo_state mirrors VO::TrackerState");

    int mbavo_vo_get_state(mbavo_vo *vo, mbavo_vo_state *st)
    {
        if (!vo || !st) return MBAVO_E_ARG;
        VO::TrackerState s;
        vo->impl.getState(s);
        memcpy(st, &s, sizeof(s));
        return 0;
    }

    int mbavo_vo_set_state(mbavo_vo *vo, const mbavo_vo_state *st)
    {
        if (!vo || !st) return MBAVO_E_ARG;
        VO::TrackerState s;
        memcpy(&s, st, sizeof(s));
        return vo->impl.setState(s);
    }

    int mbavo_vo_set_keyframe(mbavo_vo *vo, const unsigned char *sharp, const float *depth_z, double sharp_cap)
    {
        if (!vo || !sharp || !depth_z) return MBAVO_E_ARG;
        VO::FrameView s{sharp, sharp_cap, 0.0};
        return vo->impl.setKeyframe(s, depth_z);
    }

    int mbavo_vo_num_keypoints(mbavo_vo *vo, int level)
    {
        if (!vo || level < 0 || level >= vo->impl.getOptions().num_pyramid_levels) return MBAVO_E_ARG;
        return vo->impl.numKeypoints(level);
    }

    int mbavo_vo_get_keypoints(mbavo_vo *vo, int level, double *xy, double *z)
    {
        if (!vo || level < 0 || level >= vo->impl.getOptions().num_pyramid_levels || !xy || !z) return MBAVO_E_ARG;
        const int K = vo->impl.numKeypoints(level);
        if (K == 0) return 0;
        hipError_t e = hipMemcpy(xy, vo->impl.deviceKeypointsXY(level), sizeof(double) * 2 * K, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(z, vo->impl.deviceKeypointsZ(level), sizeof(double) * K, hipMemcpyDeviceToHost);
        return (int)e;
    }

    int mbavo_vo_track_frame(mbavo_vo *vo, const unsigned char *sharp, const float *depth_z, double sharp_cap,
                             const unsigned char *blur, double blur_cap, double blur_exp, double T_out[7], mbavo_vo_info *info)
    {
        if (!vo || !T_out) return MBAVO_E_ARG;
        VO::FrameView s{sharp, sharp_cap, 0.0}, b{blur, blur_cap, blur_exp};
        Core::Transformation T;
        VO::TrackInfo ti;
        const int rc = vo->impl.trackFrame(s, b, depth_z, &T, &ti);
        if (rc != 0) return rc;
        memcpy(T_out, T.getData(), sizeof(double) * 7);
        if (info)
        {
            info->is_keyframe = ti.is_keyframe; info->num_keypoints0 = ti.num_keypoints0; info->num_trace = ti.num_trace;
            info->start_idx = ti.start_idx; info->avg_flow = ti.avg_flow; info->avg_kernel = ti.avg_kernel;
            info->final_cost = ti.final_cost;
        }
        return 0;
    }

    int mbavo_p2p_create(mbavo_ctx *ctx, int rank, int world, long long max_doubles_per_slot, unsigned char *handle_out)
    {
        if (!ctx) return MBAVO_E_ARG;
        return ctx->engine->p2p_create(rank, world, max_doubles_per_slot, handle_out);
    }
    int mbavo_p2p_connect(mbavo_ctx *ctx, const unsigned char *all_handles) { return ctx ? ctx->engine->p2p_connect(all_handles) : MBAVO_E_ARG; }
    int mbavo_p2p_ranks(mbavo_ctx *ctx) { return ctx ? ctx->engine->p2p_ranks() : 0; }
    int mbavo_allgather_blocks_p2p(mbavo_ctx *ctx, double *d, long long count_per_rank)
    {
        return ctx ? ctx->engine->p2p_collective(0, d, count_per_rank) : MBAVO_E_ARG;
    }
    int mbavo_allreduce_blocks_p2p(mbavo_ctx *ctx, double *d, long long count) { return ctx ? ctx->engine->p2p_collective(1, d, count) : MBAVO_E_ARG; }
    int mbavo_p2p_status(mbavo_ctx *ctx) { return ctx ? ctx->engine->p2p_status() : MBAVO_E_ARG; }
    int mbavo_p2p_set_timeout(mbavo_ctx *ctx, double seconds) { return ctx ? ctx->engine->p2p_set_timeout(seconds) : MBAVO_E_ARG; }
    int mbavo_p2p_disconnect(mbavo_ctx *ctx) { return ctx ? ctx->engine->p2p_disconnect() : MBAVO_E_ARG; }
    int mbavo_p2p_destroy(mbavo_ctx *ctx) { return ctx ? ctx->engine->p2p_destroy() : MBAVO_E_ARG; }

    int mbavo_profile(mbavo_ctx *ctx, int enable)
    {
        if (!ctx) return MBAVO_E_ARG;
        ctx->engine->profile_enable(enable);
        return 0;
    }

    int mbavo_profile_read(mbavo_ctx *ctx, double *ms, int *n)
    {
        if (!ctx) return MBAVO_E_ARG;
        return ctx->engine->profile_read(ms, n);
    }

    void mbavo_timing_report(void) { mbavo::PhaseTimers::get().report(); }
    void mbavo_reload_env(void) { mbavo::reload_env_overrides(); }
    void mbavo_ride_along_stats(long long out[3])
    {
        mbavo::RideAlongStats &s = mbavo::RideAlongStats::get();
        out[0] = s.posts.exchange(0);
        out[1] = s.hits.exchange(0);
        out[2] = s.waits.exchange(0);
    }

    const char *mbavo_last_kernel(mbavo_ctx *ctx) { return ctx ? ctx->engine->last_kernel() : ""; }

    // ---- multi-GPU (multi_gpu.hip)
    int mbavo_shard_keypoints(const mbavo_problem *whole, int rank, int world, mbavo_problem *shard, int *first)
    {
        return mbavo::shard_keypoints(whole, rank, world, shard, first);
    }

    int mbavo_shard_frames(const mbavo_problem *whole, int rank, int world, mbavo_problem *shard, int *first)
    {
        return mbavo::shard_frames(whole, rank, world, shard, first);
    }

    int mbavo_system_len(int N) { return 1 + 6 * N + 36 * N * N; }

    int mbavo_merge_device(mbavo_ctx *ctx, int B, const mbavo_problem *probs, int k, const double *d_fb, double *d_systems)
    {
        if (!ctx) return MBAVO_E_ARG;
        return ctx->engine->merge_device(B, probs, k, d_fb, d_systems);
    }

    int mbavo_comm_unique_id(unsigned char *id) { return id ? mbavo::comm_unique_id(id) : MBAVO_E_ARG; }

    int mbavo_comm_init(mbavo_ctx *ctx, const unsigned char *id, int rank, int world)
    {
        if (!ctx || !id || world < 1 || rank < 0 || rank >= world) return MBAVO_E_ARG;
        return ctx->engine->comm_init(id, rank, world);
    }

    int mbavo_comm_ranks(mbavo_ctx *ctx) { return ctx ? ctx->engine->comm_ranks() : 0; }

    int mbavo_comm_destroy(mbavo_ctx *ctx) { return ctx ? ctx->engine->comm_destroy() : MBAVO_E_ARG; }

    int mbavo_allreduce_blocks(mbavo_ctx *ctx, void *comm, double *d_blocks, long long count)
    {
        if (!ctx || !d_blocks || count < 0) return MBAVO_E_ARG;
        return ctx->engine->allreduce(comm, d_blocks, d_blocks, count);
    }

    int mbavo_allreduce_blocks_to(mbavo_ctx *ctx, void *comm, const double *d_send, double *d_recv, long long count)
    {
        if (!ctx || !d_send || !d_recv || count < 0) return MBAVO_E_ARG;
        return ctx->engine->allreduce(comm, d_send, d_recv, count);
    }

    int mbavo_allgather_blocks(mbavo_ctx *ctx, void *comm, double *d_blocks, long long count_per_rank)
    {
        if (!ctx || !d_blocks || count_per_rank < 0) return MBAVO_E_ARG;
        return ctx->engine->allgather(comm, d_blocks, count_per_rank);
    }
}
