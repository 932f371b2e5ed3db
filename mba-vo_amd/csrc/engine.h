// engine.h -- fused batch evaluation engine (HIP).  Internal to the library.
//
// One call evaluates B independent alignment problems (pyramid levels of one
// pair, or many keyframe pairs); small problems in ONE launch (k_fused_sp<.., ONE>: pose
// entries in the prologue, finalize by the last workgroup of a slot), large ones with
// three launches on one stream:
//   k_pose_table  : F*S blur-sample poses + pose-to-knot Jacobians per problem
//                   (the work of compute_virtual_camera_poses.cu:9-110); when every tile has a CU to itself and
//                   S <= 8 this is the fused kernel's prologue instead (k_fused<.., POSE = true>: two launches)
//   k_fused       : patch centres, per-pixel residual / 1x6k Jacobian over the S
//                   samples, Huber, packed outer products, per-tile partial sums
//                   (compute_local_patches_xy.cu, compute_hessian_gradients_cost.cu:23-239)
//   k_finalize    : fixed-order sum of the tile partials into the per-frame packed
//                   blocks (compute_hessian_gradients_cost.cu:247-283)
// No intermediate of the reference pipeline (per-sample Jacobians, per-pixel rows,
// per-patch blocks) is materialised in HBM.
#ifndef MBAVO_ENGINE_H
#define MBAVO_ENGINE_H

#include "options.h"
#include "../../include/mbavo.h"
#include <hip/hip_runtime.h>
#include <map>
#include <vector>

namespace mbavo
{
    // device-visible descriptor of one problem
    struct ProblemDesc
    {
        const unsigned char *ref_img;
        const float *ref_dIxy;
        const unsigned char *const *cur_imgs;
        const double *kp_xy;
        const double *kp_z;
        const int *pattern;
        const unsigned char *outlier;
        const double *cap, *exp_t;
        const double *knots_t, *knots_R;
        double fx, fy, cx, cy;
        double t0, dt;
        double huber_a;
        double inv_num_residuals;
        int S, F, K, P, N, H, W, kp_stride, grad_fp16;
        int pose_base;        // first PoseEntry of this problem (entry = f*S + s)
        int bf_base;          // first (problem, frame) slot
        long long pixel_base; // first pixel of this problem in the rho scratch (f*K*P + kp*P + p)
        long long patch_base; // first patch of this problem in the patch-cost output (f*K + kp)
        // device-side LM (lm_batch.hip): bit 0 = take part in cost-only passes, bit 1 = in H/g passes (null: always);
        // 1/((K - bad)*F*P) kept on the device because the outlier count changes there (null: the field above)
        const int *active;
        const double *inv_ptr;
    };

    // a tile = a contiguous keypoint range of one (problem, frame), handled by one workgroup
    struct TileDesc
    {
        int prob, frame, kp_begin, kp_count;
    };

    struct P2PState;

    class Engine
    {
    public:
        explicit Engine(int device);
        ~Engine();
        void set_stream(hipStream_t s) { stream_ = s; }
        hipStream_t stream() const { return stream_; }
        int device() const { return device_; }

        // asynchronous; see mbavo_eval_batch
        int evaluate(int B, const mbavo_problem *probs, int kdeg, bool with_hessian,
                     double *d_frame_blocks, double *d_patch_cost, double *d_valid,
                     double *d_patch_blocks_strided /* B == 1 only, stride E, may be null */,
                     const int *d_active_mask = nullptr /* [B] */, const double *d_inv = nullptr /* [B] */,
                     bool signal_host = false /* arm the pinned completion word (see wait_evaluation) */,
                     bool same_list = false /* the caller vouches: the very problem list (every field, every pointer) of the previous
                                               evaluate() of this engine -- the layout is not rebuilt or compared (the batched LM's
                                               passes: building and comparing 512 descriptors cost ~25 us of host time per pass) */);
        // blocks until the evaluation just enqueued has completed (completion word, or the stream)
        int wait_evaluation();

        // Persistent evaluation of ONE small problem (host-driven LM loop, tracker.cpp): one launch per pyramid level, the
        // resident workgroups take commands from a CPU-writable device block (see k_sp_persist).  There are kPushSlots
        // independent command blocks ("slots"): the kernel of the NEXT level can be enqueued behind the running one while
        // the current level is still being evaluated (it starts the moment its predecessor exits, with its launch cost
        // hidden behind an evaluation in flight); commands may only be posted to the oldest kernel that has not been ended.
        // begin: 0 = started, 1 = not applicable (the problem does not take the single-launch sample-parallel kernel, or
        // cached_only and its layout would have to be built and uploaded: use evaluate() / try again when its turn comes),
        // < 0 / > 0 = error.  The problem's knots, outlier flags and residual scale live in the slot's push block, the
        // patch-cost output and the frame blocks in pinned host memory.  post + wait (= eval): one evaluation.  end: tells
        // the slot's workgroups to exit (asynchronous).  No other work may be enqueued on the engine's stream in between.
        static constexpr int kPushSlots = 8;
        int persistent_begin(int slot, const mbavo_problem &p, int kdeg, double *h_frame_blocks, double *h_patch_cost, const double *h_inv,
                             bool cached_only = false);
        // ONE kernel for a list of B problems that share their knot buffer (round 3: the pyramid levels of a tracked frame -- one
        // launch per frame instead of one per level): a command names the problem it evaluates (persistent_post's `prob`), the
        // workgroups of the other problems' tiles skip it.  h_inv: B consecutive words (problem b's scale at h_inv[b]); frame
        // blocks / patch costs of problem b at its rows of the list (frames / patches of the problems before it first).
        int persistent_begin(int slot, int B, const mbavo_problem *probs, int kdeg, double *h_frame_blocks, double *h_patch_cost,
                             const double *h_inv, bool cached_only = false);
        // Fine-grained DEVICE memory the CPU writes directly through the PCIe BAR (write-combining: stores + sfence; never
        // read it from the CPU): a slot's command block (its first 64 bytes, owned by the engine) and the per-evaluation
        // inputs the caller lays out behind it (knots, residual scale, outlier flags).  Pushing the inputs costs the GPU
        // nothing; PULLING them from pinned host memory is bounded by the bus' small-read rate (~10 M/s: 35 us per
        // evaluation for a dozen words per workgroup; tools/micro/host_push_probe.hip: 1.9 us round trip pushed).
        // nullptr when the platform cannot do it (or the blocks would have to grow while a persistent kernel runs).
        void *push_block(int slot, size_t bytes);
        static constexpr size_t kPushHeader = 64;
        // prob2 >= 0 (round 5): the command ALSO evaluates problem prob2 of the kernel's list, with H / g, at the knots the caller put
        // into the second knot area (7 N doubles behind the first); persistent_wait() waits for `prob` only, the second problem's
        // completion is asked for by the sequence number of its post (posted_seq())
        int persistent_post(int slot, bool with_hessian, int prob = 0, int prob2 = -1);
        // (round 5) The H / g evaluation that was problem `prob`'s LAST command, summed again under the outlier flags and the residual
        // scale as the push block holds them NOW -- what an H / g evaluation at the same knots would return, bit for bit, without
        // the pixel work (engine.hip: sp_resum_body).  persistent_resum_ok: the kernel of `slot` can do that for `prob` (one patch
        // per wave, tiles of one round); the caller guarantees that no other command went to that problem in between.
        bool persistent_resum_ok(int slot, int prob = 0) const { return persistent_active(slot) && prob >= 0 && prob < 15 && (persist_resum_[slot] >> prob & 1u) != 0; }
        int persistent_post_resum(int slot, int prob = 0);
        int persistent_wait();
        unsigned long long posted_seq() const { return pending_seq_; }
        bool persistent_second_done(unsigned long long seq) const;
        int persistent_wait_second(unsigned long long seq);
        int persistent_eval(int slot, bool with_hessian, int prob = 0) { const int r = persistent_post(slot, with_hessian, prob); return r ? r : persistent_wait(); }
        int persistent_end(int slot);
        int persistent_end_all();
        bool persistent_active(int slot) const { return slot >= 0 && slot < kPushSlots && (persist_mask_ >> slot & 1u) != 0; }
        const ProblemDesc *device_descs() const { return (const ProblemDesc *)d_descs_; }
        // The batched LM (lm_batch.hip) sums the tile partials of a (problem, frame) slot inside its own kernels instead of
        // reading frame blocks: with set_defer_finalize(true) an evaluate() whose list takes the flat finalize (>= 64 slots of
        // <= 4 tiles each) launches NO finalize kernel and leaves d_frame_blocks untouched; finalize_deferred() says whether the
        // last evaluate() did so.  Partial of tile t: doubles [t * stride, (t + 1) * stride) = [valid | g, H sums 1 .. E-1 | cost |
        // spare] (unscaled: the residual scale is applied by whoever sums); slot bf owns tiles [begin[bf], begin[bf + 1]).
        // The batched LM's solve kernel writes the pose entries of the candidate it produces into the engine's table itself
        // (entry of (problem, frame f, sample s) at pose_base + f S + s, the k_pose_table layout): with set_external_poses(true)
        // an evaluate() launches neither k_pose_table nor the fused kernel's pose prologue (the single-launch sample-parallel
        // kernels keep computing their own in LDS).
        void set_external_poses(bool on, void *table = nullptr) { external_poses_ = on; external_table_ = on ? table : nullptr; }
        void *device_pose_table() const { return d_poses_; }
        // A second tiling of the same problem list (batched LM, round 4): `tiles` > 0 makes the layout aim at that many tiles instead
        // of one per CU (e.g. 4 per pair: a pass over a FEW active pairs then takes one round instead of a pair's three); prepare()
        // builds and uploads the layout of a list without launching anything (so that the first pass through it finds it ready).
        void set_tile_target(long long tiles) { tile_target_ = tiles; }
        int prepare(int B, const mbavo_problem *probs, int kdeg, const int *d_active_mask, const double *d_inv);
        // the engine's companion for that second tiling (created at first use on the same device and stream, owned by this engine)
        Engine *companion();
        Engine *companion_if_any() const { return companion_; }
        int *device_status() const { return (int *)d_status_; }
        // scheduling options (include/mbavo.h: mbavo_engine_opts; all zero = the defaults) and what they resolve to under the
        // environment's override layer (options.h)
        struct Options { int sample_parallel = 0, single_launch = 0, fused_pose = 0, fused_pose_max_samples = 0, persistent = 0, prelaunch = 0,
                             tiles_per_cu = 0, min_tile_pixels = 0, sp_max_slot_tiles = 0; };
        void set_options(const Options &o) { opts_ = o; layout_uploaded_ = false; h_descs_.clear(); }
        const Options &options() const { return opts_; }
        EngineTuning tuning() const;
        void set_defer_finalize(bool on) { defer_finalize_ = on; }
        // The NEXT evaluate() with H/g also leaves every problem's merged system [cost | g (6N) | H (6N x 6N, column-major)]
        // (merge_hessian_gradient_cost.cpp:39-86; mbavo_system_len(N) doubles each, back to back) in d_systems: written by the
        // finalize step itself where every problem has one frame and N == k (no extra launch), by the merge kernel behind it
        // otherwise.  One-shot: cleared by that evaluate().
        void set_merge_target(double *d_systems) { merge_target_ = d_systems; }
        bool last_merge_fused() const { return merge_fused_last_; }
        bool finalize_deferred() const { return deferred_last_; }
        const double *device_partials() const { return (const double *)d_partials_; }
        const int *device_bf_tile_begin() const { return (const int *)d_bf_tile_begin_; }
        // (for the launch sequence, a free function template in engine.hip) leaves the finalize to the caller if asked to and possible
        bool take_deferral(bool flat_finalize) { return deferred_last_ = flat_finalize && defer_finalize_; }

        // range status since the previous fetch (call after a stream sync): non-zero if a blur
        // sample's knot segment had to be clamped into [0, N-k]
        int fetch_status();
        // the same without a blocking copy of its own: enqueue the counter's copy into pinned host memory on the engine's stream
        // (fetch_status_enqueue), synchronise the stream with whatever else is pending, then take the delta (fetch_status_take)
        int fetch_status_enqueue(int *h_pinned);
        int fetch_status_take(const int *h_pinned);

        int total_bf() const { return total_bf_; }
        int num_tiles() const { return (int)h_tiles_.size(); }
        int num_cus() const { return num_cus_; }
        bool layout_flat() const { return flat_finalize_; } // the cached layout takes the one-block-per-slot finalize (deferrable)
        // name of the dominant kernel the last evaluate() dispatched, e.g. "k_fused<4,true,false>" (bench labels)
        const char *last_kernel();

        // optional per-launch timing of the dominant kernel (k_fused) with HIP events on the
        // engine's stream; read back after a stream sync (bench.py roofline leg)
        void profile_enable(int every); // 0 = off, n > 0 = time every n-th launch
        int profile_read(double *fused_ms_sum, int *launches);

        // named device scratch that persists across calls (grown on demand, freed with the engine): the LM loop
        // keeps its knots / flags / patch-cost buffers here instead of hipMalloc'ing per call
        void *named_scratch(int slot, size_t bytes);

        // persistent staging owned by the context (used by mbavo_eval / tracker)
        double *host_frame_blocks(size_t n_doubles);
        // further pinned, device-visible host buffers (grown on demand, freed with the engine): the LM loop has the
        // fused kernel write the per-patch costs straight into one (no D2H copy for the outlier statistics) and stages
        // the outlier flags in another (an H2D copy from pinned memory is asynchronous, from pageable memory it is not)
        void *pinned_scratch(int slot, size_t bytes);

        // multi-GPU (multi_gpu.hip): the context's own RCCL communicator, the in-place sum over ranks on this engine's
        // stream, and merge_hessian_gradient_cost on the device (packed frame blocks -> [cost | g | H] systems)
        int comm_init(const unsigned char *unique_id, int rank, int world);
        int comm_ranks() const;
        int comm_destroy();
        // One-shot collectives over peer-mapped receive regions (p2p_comm.hip): no RCCL, one kernel per collective
        int p2p_create(int rank, int world, long long max_doubles_per_slot, unsigned char *handle_out /* MBAVO_P2P_HANDLE_BYTES */);
        int p2p_connect(const unsigned char *all_handles /* world x MBAVO_P2P_HANDLE_BYTES, rank order */);
        int p2p_ranks() const;
        int p2p_collective(int mode /* 0 all-gather in place, 1 all-reduce in place */, double *d_buf, long long count);
        int p2p_status();
        int p2p_set_timeout(double seconds);
        int p2p_disconnect();
        int p2p_destroy();
        int allreduce(void *caller_comm_or_null, const double *d_send, double *d_recv, long long count);
        int allgather(void *caller_comm_or_null, double *d, long long count_per_rank);
        int merge_device(int B, const mbavo_problem *probs, int kdeg, const double *d_frame_blocks, double *d_systems);

    private:
        int ensure(void **ptr, size_t *cap, size_t bytes);
        int rebuild_layout(int B, const mbavo_problem *probs, int kdeg, const int *d_active, const double *d_inv, bool cached_only = false);

        int device_;
        hipStream_t stream_ = nullptr;
        int num_cus_ = 256;

        // cached layout of the last problem list
        std::vector<ProblemDesc> h_descs_;
        std::vector<TileDesc> h_tiles_;
        std::vector<int> h_bf_tile_begin_; // nBF + 1
        std::vector<int> h_bf_prob_;       // nBF
        std::vector<int> h_entry_prob_;    // problem of every pose-table entry (saves the pose kernel a search)
        int cached_kdeg_ = 0;
        int total_bf_ = 0, total_entries_ = 0;
        long long total_pixels_ = 0, total_patches_ = 0;
        bool layout_uploaded_ = false;
        int sp_logs_ = 0; // > 0: the cached layout is tiled for the sample-parallel kernel with S = 2^sp_logs_
        bool flat_finalize_ = false; // many (problem, frame) slots of <= 4 tiles each: k_finalize_flat
        bool defer_finalize_ = false, deferred_last_ = false; // set_defer_finalize / what the last evaluate() did
        Options opts_;
        double *merge_target_ = nullptr;                       // set_merge_target
        bool merge_fused_last_ = false;
        bool external_poses_ = false;                          // set_external_poses
        void *external_table_ = nullptr;                       // ... reading another engine's table
        long long tile_target_ = 0;                            // set_tile_target
        Engine *companion_ = nullptr;
        bool empty_slots_ = false;   // some (problem, frame) slot has no tile (K == 0): no workgroup would finalize it in the single-launch form

        void *d_layout_ = nullptr; size_t cap_layout_ = 0; // one arena: descs | tiles | bf_tile_begin | bf_prob | entry_prob
        std::vector<char> h_layout_;
        void *d_descs_ = nullptr, *d_tiles_ = nullptr, *d_bf_tile_begin_ = nullptr, *d_bf_prob_ = nullptr, *d_entry_prob_ = nullptr;
        // The trackers cycle through a few layouts (one per pyramid level, the same ones frame after frame until the
        // keyframe changes): the layouts that are not active wait here with their device arenas, and a problem list that
        // matches one is a pointer swap instead of a tiling pass and an upload (4 uploads per tracked frame otherwise).
        // what a layout depends on besides the problem list: part of the cache key (a changed tile target or tiling option
        // must not find a stale layout)
        struct LayoutKey
        {
            long long tile_target = 0;
            int tiles_per_cu = 1, sample_parallel = -1, min_tile_pixels = 256, sp_max_slot_tiles = 64;
            bool operator==(const LayoutKey &o) const
            {
                return tile_target == o.tile_target && tiles_per_cu == o.tiles_per_cu && sample_parallel == o.sample_parallel &&
                       min_tile_pixels == o.min_tile_pixels && sp_max_slot_tiles == o.sp_max_slot_tiles;
            }
        };
        struct ParkedLayout
        {
            LayoutKey key;
            std::vector<ProblemDesc> descs;
            std::vector<TileDesc> tiles;
            std::vector<int> bf_tile_begin, bf_prob, entry_prob;
            int kdeg = 0, total_bf = 0, total_entries = 0, sp_logs = 0;
            long long total_pixels = 0, total_patches = 0;
            bool uploaded = false, flat_finalize = false, empty_slots = false;
            void *d_layout = nullptr; size_t cap_layout = 0;
            void *d_descs = nullptr, *d_tiles = nullptr, *d_bf_tile_begin = nullptr, *d_bf_prob = nullptr, *d_entry_prob = nullptr;
        };
        static constexpr int kParkedLayouts = 4;
        ParkedLayout parked_[kParkedLayouts];
        int parked_victim_ = 0;
        void swap_layout(ParkedLayout &s);
        LayoutKey layout_key_;              // of the active layout
        unsigned long long layout_gen_ = 0; // bumped whenever the active layout changes (rebuild, swap): evaluate(same_list) checks it
        unsigned long long same_list_gen_ = 0;
        const mbavo_problem *same_list_probs_ = nullptr;
        void *d_poses_ = nullptr; size_t cap_poses_ = 0;
        void *d_rho_ = nullptr; size_t cap_rho_ = 0;
        void *d_partials_ = nullptr; size_t cap_partials_ = 0;
        void *d_status_ = nullptr;
        void *d_tickets_ = nullptr; size_t cap_tickets_ = 0;
        void *h_flag_ = nullptr;            // pinned completion word of the single-launch kernels
        unsigned long long flag_seq_ = 0;
        bool flag_pending_ = false;
        void *d_push_ = nullptr; size_t cap_push_ = 0; // fine-grained device memory, CPU-writable: kPushSlots blocks of push_stride_
        size_t push_stride_ = 0;                       // bytes, each starting with its PersistCmd
        int push_probe_ = 0;                           // 0 = not probed, 1 = the CPU can store into device memory (large BAR), -1 = it cannot
        unsigned persist_mask_ = 0;                    // slots with a persistent kernel enqueued and not ended
        unsigned long long pending_seq_ = 0;           // sequence number of the evaluation posted last
        int pending_mode_ = 0;                         // ... and its mode (1 cost-only, 2 H / g, 3 summed again): the timing aid's key
        int persist_gen_ = 0;
        int persist_gen_of_[kPushSlots] = {};          // generation of the kernel enqueued on each slot
        unsigned persist_resum_[kPushSlots] = {};      // per slot: the problems of its kernel's list that can be summed again (persistent_resum_ok)
        int status_seen_ = 0;
        void *h_fb_ = nullptr; size_t cap_hfb_ = 0;
        static constexpr int kPinnedSlots = 8; // 0-4 host-driven LM loop (tracker.cpp), 7 lm_batch
        void *pinned_[kPinnedSlots] = {};
        size_t pinned_cap_[kPinnedSlots] = {};

        static constexpr int kSlots = 16; // 0-6 LM loop (tracker.cpp), 8-10 keyframe detection (keyframe_ops.hip), 7 / 11 merge_device, 15 lm_batch
        void *slots_[kSlots] = {};
        size_t slot_cap_[kSlots] = {};

        char last_kernel_[64] = "";
        int last_kernel_id_[6] = {0, 0, 0, 0, 0, 0};
        std::vector<ProblemDesc> scratch_descs_;
        void *comm_ = nullptr;          // ncclComm_t owned by this context (comm_init)
        struct P2PState *p2p_ = nullptr; // peer-mapped receive regions (p2p_create)
        std::vector<char> merge_descs_; // what merge_device last uploaded (re-uploaded only when it changes)
        std::vector<int> merge_start_;
        int merge_kdeg_ = 0;
        void *merge_dev_[2] = {nullptr, nullptr};

        int prof_every_ = 0, prof_seen_ = 0;
        std::map<const void *, size_t> lds_attr_;
        std::vector<hipEvent_t> prof_ev_; // pairs (start, stop)
        int prof_used_ = 0;

    public:
        // called by the launch helper of k_fused
        bool prof_events(hipEvent_t *e0, hipEvent_t *e1);
        // hipFuncAttributeMaxDynamicSharedMemorySize, set once per kernel on this engine's device
        hipError_t ensure_lds(const void *kernel, size_t bytes);
    };

    // multi_gpu.hip
    int comm_unique_id(unsigned char *id);
    int shard_keypoints(const mbavo_problem *whole, int rank, int world, mbavo_problem *out, int *first);
    int shard_frames(const mbavo_problem *whole, int rank, int world, mbavo_problem *out, int *first);
} // namespace mbavo

#endif
