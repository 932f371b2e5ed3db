/*
 * mbavo.h -- C ABI of the MI355X-native blur-aware photometric tracking path.
 *
 * Drop-in boundary for the hot path of ethliup/MBA-VO (src/ba_tracker).  Every
 * entry point names the reference interface it replaces (paths relative to the
 * reference's src/).  Plain pointers and sizes only; `d_` = device (HIP) memory,
 * `h_` = host memory.  All functions return 0 on success, a positive hipError_t
 * value on a HIP failure, or a negative MBAVO_E_* code; none of them falls back
 * to a CPU path -- without a usable HIP device the compute entry points fail.
 *
 * The C++ API with the reference's exact signatures (namespace SLAM::VO) is in
 * mba-vo_amd/csrc/ba_tracker.h and is exported by the same shared library.
 */
#ifndef MBAVO_H
#define MBAVO_H

#ifdef __cplusplus
extern "C" {
#endif

#define MBAVO_E_ARG (-1)      /* bad argument (null pointer, unsupported spline degree, ...) */
#define MBAVO_E_RANGE (-2)    /* a blur sample fell outside the spline's knot range (SplineFunctor.h:13-19 has no check) */
#define MBAVO_E_NODEVICE (-3) /* no HIP device: the product has no CPU fallback */
#define MBAVO_E_TIMEOUT (-4)  /* a bounded device-side wait ran out (a peer rank never arrived; a wedged device) */

typedef struct mbavo_ctx mbavo_ctx;

/* One alignment problem = the argument list of evaluate_cost_hessian_gradient
 * (ba_tracker/spline_update_step.h:70-87) plus the CudaSharedStorages fields it
 * reads (spline_update_step.h:18-58), as POD.  All pointers are DEVICE pointers
 * except h_start_idx.
 * ZERO-INITIALISE the struct (memset / `= {0}`) before filling it in: fields appended by later library versions
 * (grad_fp16, num_residuals -- ABI 2; the scheduling / solver-form tails of the option structs -- ABI 3) select optional
 * behaviour and must read 0 when a caller does not know them.
 * mbavo_abi_version() returns the struct revision the loaded library expects (MBAVO_ABI_VERSION of this header). */
#define MBAVO_ABI_VERSION 3
typedef struct mbavo_problem {
    int S;                                /* n_vir_poses_per_frame */
    int F;                                /* n_frames */
    int K;                                /* num_keypoints */
    int P;                                /* patch_size */
    int N;                                /* num_ctrl_knots */
    int H, W;                             /* im_size_HW; H*W <= 2^29 (32-bit tap offsets), else MBAVO_E_ARG */
    const unsigned char *d_ref_img;       /* cuda_ref_img, H*W u8 */
    const float *d_ref_dIxy;              /* cuda_dIxy_ref, H*W*2 interleaved [dx,dy] */
    const unsigned char *const *d_cur_imgs; /* storages.cuda_cur_images: device array of F device pointers */
    const double *d_kp_xy;                /* keypoint i at d_kp_xy[i*kp_stride + {0,1}] */
    int kp_stride;                        /* 2 = packed xy; 3 = Core::Vector2d array, pointer at .values */
    const double *d_kp_z;                 /* storages.cuda_keypoint_depth_z, K */
    const int *d_pattern;                 /* storages.cuda_local_patch_pattern_xy, P*2 (dx,dy) */
    const unsigned char *d_outlier;       /* storages.cuda_keypoints_outlier_flags, K, or NULL */
    int num_bad;                          /* storages.num_bad_keypoints */
    double intrinsics[4];                 /* fx fy cx cy at this pyramid level */
    const double *d_cap_time;             /* storages.cuda_img_cap_time, F */
    const double *d_exp_time;             /* storages.cuda_img_exp_time, F */
    double t0, dt;                        /* spline_start_time, spline_sample_dt */
    const double *d_knots_t;              /* storages.cuda_spline_ctrl_knots_data_t, 3N */
    const double *d_knots_R;              /* storages.cuda_spline_ctrl_knots_data_R, 4N xyzw */
    const int *h_start_idx;               /* cpu_ctrl_knot_start_indices, F (host; only read by host merges) */
    double huber_a;
    int grad_fp16;                        /* 0: d_ref_dIxy is float [dx,dy] per pixel (reference layout, Gradient.h:16-75);
                                             1: d_ref_dIxy is IEEE half [dx,dy] per pixel (4 B/pixel, BASELINE configs[4]).
                                             Central differences of an 8-bit image are multiples of 0.5 in [-127.5, 127.5],
                                             exactly representable in fp16, so both formats see identical tap values (results agree to rounding).
                                             2: d_ref_dIxy is the packed keyframe of mbavo_pack_keyframe_u8 (4 B/pixel: intensity and both
                                             differences in one word; H/g passes read nothing else of the keyframe, cost-only passes d_ref_img).
                                             All problems of one mbavo_eval_batch call must use the same format. */
    long long num_residuals;              /* 0: the blocks are scaled by 1/((K - num_bad)*F*P) of THIS problem
                                             (inv_num_residuals, spline_update_step.cpp:116-117).  > 0: by 1/num_residuals --
                                             the residual count of the WHOLE problem a keypoint / frame shard belongs to, so
                                             that the shards' packed blocks simply add up to the whole problem's (multi-GPU:
                                             mbavo_shard_keypoints / mbavo_shard_frames fill it in) */
} mbavo_problem;

/* ---- context: owns all device scratch (replaces initialize/free_shared_cuda_storages,
 * ba_tracker/spline_update_step.cpp:9-95, for the fused path) */
int mbavo_create(mbavo_ctx **out, int device_id);
int mbavo_destroy(mbavo_ctx *ctx);
int mbavo_set_stream(mbavo_ctx *ctx, void *hip_stream); /* NULL = default stream */
/* sizeof of the library's own structs, for a foreign-language binding to check its mirror against: 0 mbavo_problem,
 * 1 mbavo_track_opts, 2 mbavo_lm_batch_opts, 3 mbavo_vo_options, 4 mbavo_engine_opts, 5 mbavo_vo_state, 6 mbavo_trace_rec,
 * 7 mbavo_level, 8 mbavo_lm_batch_result, 9 mbavo_vo_info */
int mbavo_sizeof(int which);

/* ---- options instead of environment variables (ABI 3).  The reference configures everything through one plain struct
 * (BlurAwareDirectTrackerOptions, blur_aware_direct_tracker.h:15-67); so does this library: every switch that changes results or
 * scheduling is a field -- here for the evaluation engine (all callers of the context), in the tails of mbavo_track_opts /
 * mbavo_vo_options / mbavo_lm_batch_opts for the LM loops.  A ZEROED struct is the default everywhere: flags are tri-state
 * (0 default, 1 on, -1 off), numbers 0 = default.  The MBAVO_* environment variables of the A/B tools still override an option
 * (one reader: csrc/options.h read_env_overrides); only the diagnostics MBAVO_TIMING / MBAVO_LM_STAMPS / MBAVO_LM_STATS are
 * environment-only. */
typedef struct mbavo_engine_opts {
    int sample_parallel;        /* small lists on the sample-parallel kernel: 0 by size, 1 wherever the list allows, -1 never [MBAVO_SP] */
    int single_launch;          /* small lists in ONE launch (pose prologue + ticket epilogue); default on            [MBAVO_ONE] */
    int fused_pose;             /* pose entries as the fused kernel's prologue; default on                            [MBAVO_FUSED_POSE] */
    int fused_pose_max_samples; /* ... up to this many blur samples; default 8                                        [MBAVO_FUSED_POSE_MAX_S] */
    int persistent;             /* persistent evaluation kernels under the host-driven LM loop; default on            [MBAVO_PERSIST] */
    int prelaunch;              /* next pyramid level's persistent kernel enqueued behind the running one; default on [MBAVO_PRELAUNCH] */
    int tiles_per_cu;           /* default 1                                                                          [MBAVO_TILES_PER_CU] */
    int min_tile_pixels;        /* default 256                                                                        [MBAVO_MIN_TILE_PX] */
    int sp_max_slot_tiles;      /* default 64                                                                         [MBAVO_SP_MAX_SLOT_TILES] */
    int reserved[7];
} mbavo_engine_opts;
int mbavo_set_engine_opts(mbavo_ctx *ctx, const mbavo_engine_opts *opts); /* NULL = all defaults */
int mbavo_get_engine_opts(mbavo_ctx *ctx, mbavo_engine_opts *opts_out);   /* what was set (not the environment's overrides) */
int mbavo_packed_len(int spline_deg_k);                 /* E = (6k+1)(6k+2)/2 */

/* ---- fused evaluation of B independent problems in one pass (poses -> residual /
 * Jacobian -> packed normal-equation blocks).  Replaces steps (1)-(5) of
 * evaluate_cost_hessian_gradient (spline_update_step.cpp:127-227) for every problem.
 * Asynchronous on the context's stream.
 *   d_frame_blocks: sum_b F_b rows of E doubles, problem-major then frame:
 *                   [cost | g_local (6k) | upper(H_local)] == cuda_frame_cost_gradient_hessian_tR
 *   d_patch_cost  : sum_b F_b*K_b doubles (slot 0 of every patch block), or NULL
 *   d_valid       : sum_b F_b doubles, number of in-bounds pixels per frame, or NULL
 * with_hessian = 0 is the reference's cost-only mode (H/g slots left untouched). */
int mbavo_eval_batch(mbavo_ctx *ctx, int B, const mbavo_problem *h_problems, int spline_deg_k,
                     int with_hessian, double *d_frame_blocks, double *d_patch_cost, double *d_valid);

/* The same evaluation ENDING IN THE REFERENCE'S UNIT: evaluate_cost_hessian_gradient with H / g outputs
 * (spline_update_step.cpp:97-241) ends in merge_hessian_gradient_cost (:232-239, merge_hessian_gradient_cost.cpp:39-86),
 * i.e. in [cost | g (6N) | H (6N x 6N, column-major, both triangles)] -- mbavo_eval_batch alone leaves the PACKED per-frame
 * blocks.  d_systems receives mbavo_system_len(N_b) doubles per problem, back to back in problem order (the layout of
 * mbavo_merge_device), d_frame_blocks the packed blocks as before.  Where every problem has one frame and N == k control
 * knots (start index 0: the scatter is a plain unpack) the finalize step stores the system itself -- no merge launch, the
 * evaluation costs what mbavo_eval_batch costs; otherwise mbavo_merge_device's kernel runs behind it (h_start_idx required).
 * Asynchronous on the context's stream. */
int mbavo_eval_batch_merged(mbavo_ctx *ctx, int B, const mbavo_problem *h_problems, int spline_deg_k,
                            double *d_frame_blocks, double *d_systems, double *d_patch_cost, double *d_valid);

/* synchronous single problem, host outputs == evaluate_cost_hessian_gradient
 * (spline_update_step.h:70-87): h_H is the 6N x 6N column-major system, h_g 6N;
 * pass NULL, NULL for cost-only.  d_patch_blocks (F*K*E, may be NULL) receives slot 0
 * of every patch block at stride E like cuda_patch_cost_gradient_hessian_tR. */
int mbavo_eval(mbavo_ctx *ctx, const mbavo_problem *h_problem, int spline_deg_k,
               double *h_total_cost, double *h_H, double *h_g, double *d_patch_blocks);

/* ---- the reference's five launchers, one call each (device buffers caller-owned,
 * synchronous like the reference). */
/* compute_virtual_camera_poses (ba_tracker/compute_virtual_camera_poses.h:18-33) */
int mbavo_compute_virtual_camera_poses(int S, int F, const double *d_cap, const double *d_exp, int spline_deg_k,
                                       double t0, double dt, const double *d_knots_t, const double *d_knots_R,
                                       double *d_poses, double *d_J_t, double *d_J_R);
/* compute_local_patches_xy (ba_tracker/compute_local_patches_xy.h:10-18); xy arrays are Core::Vector2d (24 B) */
int mbavo_compute_local_patches_xy(int S, int F, const double *d_poses, const void *d_keypoints_vec2d,
                                   const double *d_kp_z, int K, const double intrinsics[4], const int HW[2],
                                   void *d_local_patches_vec2d);
/* compute_pixel_jacobian_residual (ba_tracker/compute_hessian_gradients_cost.h:11-29) */
int mbavo_compute_pixel_jacobian_residual(const unsigned char *d_I_ref, const float *d_dIxy_ref,
                                          const unsigned char *const *d_I_cur_imgs, int S, int F,
                                          const double *d_poses, int spline_deg_k, const double *d_J_t,
                                          const double *d_J_R, const void *d_local_patches_vec2d,
                                          const double *d_kp_z, int K, const int *d_pattern, int P,
                                          const double intrinsics[4], const int HW[2],
                                          double *d_pixel_residuals, double *d_pixel_jacobians_or_null);
/* compute_patch_cost_gradient_hessian (compute_hessian_gradients_cost.h:52-60) */
int mbavo_compute_patch_cost_gradient_hessian(int F, int K, int P, int spline_deg_k, const double *d_residuals,
                                              const double *d_jacobians_or_null, double huber_a,
                                              double inv_num_residuals, double *d_patch_blocks);
/* compute_frame_cost_gradient_hessian (compute_hessian_gradients_cost.h:62-68) */
int mbavo_compute_frame_cost_gradient_hessian(int F, int K, int spline_deg_k, const double *d_patch_blocks,
                                              int eval_gradient_hessian, const unsigned char *d_outlier_or_null,
                                              double *d_frame_blocks);
/* merge_hessian_gradient_cost (ba_tracker/merge_hessian_gradient_cost.h:8-15): D2H + scatter */
int mbavo_merge_hessian_gradient_cost(int F, int spline_deg_k, const double *d_frame_blocks, const int *h_start_idx,
                                      int N, double *h_total_cost, double *h_H, double *h_g);
/* the scatter alone, on host blocks (merge_hessian_gradient_cost.cpp:39-86) */
int mbavo_merge_host(int F, int spline_deg_k, const double *h_frame_blocks, const int *h_start_idx, int N,
                     double *h_total_cost, double *h_H, double *h_g);
/* solve_normal_equation (ba_tracker/solve_normal_equation.h:10-35): x = -A^+ b; 0 = Jacobi SVD, 1 = LDLT.
 * Deviation from the reference, on by default: for solver_type 0 a positive definite system whose LDL^T pivot ratio is at
 * most 1e8 is solved by LDL^T instead of the Jacobi SVD (the same x to rounding x cond(A); the batched LM additionally
 * refines in double-double up to a ratio of 1e13).  MBAVO_FAST_SOLVE=0 in the environment (read once per
 * mbavo_solve_normal_equation / mbavo_optimize_trajectory / mbavo_lm_batch call) reproduces solve_normal_equation.h case 0
 * -- JacobiSVD::solve, minimum-norm least squares -- for every system; rank-deficient systems always take it. */
int mbavo_solve_normal_equation(const double *h_A_colmajor, const double *h_b, int n, int solver_type, double *h_x);

/* ---- host control flow that decides how often the path runs */
/* LevenbergMarquardtStrategy (ba_tracker/levenberg_marquardt_strategy.h:8-27) */
typedef struct mbavo_lm mbavo_lm;
mbavo_lm *mbavo_lm_new(void);
void mbavo_lm_delete(mbavo_lm *);
void mbavo_lm_reset(mbavo_lm *);
void mbavo_lm_step_accepted(mbavo_lm *, double step_quality);
void mbavo_lm_step_rejected(mbavo_lm *);
double mbavo_lm_get_radius(mbavo_lm *);
/* TrustRegionStepEvaluator (ba_tracker/trust_region_step_evaluator.h:37-80) */
typedef struct mbavo_tr mbavo_tr;
mbavo_tr *mbavo_tr_new(int max_consecutive_nonmonotonic_steps);
void mbavo_tr_delete(mbavo_tr *);
void mbavo_tr_reset(mbavo_tr *, double initial_cost);
double mbavo_tr_step_quality(mbavo_tr *, double cost, double model_cost_change);
void mbavo_tr_step_accepted(mbavo_tr *, double cost, double model_cost_change);

/* SplineSE3 pieces (core/common/Spline.h:222-330) on flat knot arrays */
int mbavo_spline_get_pose(int spline_deg_k, double t0, double dt, const double *h_knots_t, const double *h_knots_R,
                          int N, double t, double h_t_out[3], double h_q_out_xyzw[4],
                          double *h_J_t_or_null /*3x3k*/, double *h_J_R_or_null /*4x3k*/);
int mbavo_spline_plus(const double *h_knots_t, const double *h_knots_R, int N, const double *h_step /*6N*/,
                      double *h_cand_t, double *h_cand_R);
int mbavo_segment_start_index(double t, double t0, double dt); /* SplineFunctor.h:13-19 */

/* ---- LM loop over the pyramid: BlurAwareDirectTracker::optimizeTrajectory
 * (ba_tracker/blur_aware_direct_tracker.cpp:544-924) on device-resident levels */
typedef struct mbavo_level {
    int H, W, K, P, S;
    const unsigned char *d_ref_img;
    const float *d_ref_dIxy;
    const unsigned char *const *d_cur_imgs; /* device array of F device pointers */
    const double *d_kp_xy;                  /* packed K*2 */
    const double *d_kp_z;
    const int *d_pattern;
} mbavo_level;
typedef struct mbavo_track_opts {
    int num_levels, spline_deg_k, max_num_iterations, max_consecutive_nonmonotonic_steps, solver_type;
    double intrinsics[4]; /* level 0 */
    double huber_k, min_step_quality, min_abs_cost_decrease, max_chi_square_error;
    /* ABI 3 -- zero = default.  Solver form: the pivot ratio up to which LDL^T stands in for solve_normal_equation.h's
     * solvers (0: 1e8; < 0: never -- the Jacobi SVD / pivoted LDL^T for every system; > 1: that ratio)        [MBAVO_FAST_SOLVE] */
    double fast_solve_ratio;
    int speculate;      /* candidates evaluated WITH H / g: 0 on the persistent levels, 1 every level, -1 never [MBAVO_SPECULATE] */
    int persist_levels; /* one persistent kernel for all levels of a call; default on                          [MBAVO_PERSIST_LEVELS] */
    int ride_along;     /* the next pyramid level's first evaluation rides along with this level's candidates (same
                           results, one dependent evaluation less per level); default on; 2 (tests): on, and every
                           ride-along is treated as taken at other knots, i.e. waited out and redone           [MBAVO_RIDE_ALONG] */
    int resum;          /* the H / g evaluation behind an accepted step whose outlier flags changed is the candidate's,
                           summed again on the device under the new flags and scale (bit-identical results, no pixel
                           work); default on                                                                   [MBAVO_RESUM] */
    int reserved[2];
} mbavo_track_opts;
typedef struct mbavo_trace_rec {
    int level, iter, kind; /* 0 initial evaluation, 1 accepted, 2 rejected, 3 invalid step */
    int num_outliers;
    double radius, eval_cost, candidate_cost, model_change, quality;
} mbavo_trace_rec;
/* knots are updated in place; returns the number of trace records (>= 0) or an error (< 0) */
int mbavo_optimize_trajectory(mbavo_ctx *ctx, const mbavo_track_opts *opts, const mbavo_level *levels, int F,
                              const double *h_cap, const double *h_exp, double t0, double dt,
                              double *h_knots_t, double *h_knots_R, int N, int *h_start_idx_out,
                              double *h_final_cost, mbavo_trace_rec *trace, int trace_cap);

/* ---- device-side LM over a batch of independent problems (one pyramid level each): optimizePyramidLevel
 * (ba_tracker/blur_aware_direct_tracker.cpp:590-924) for B problems at once with the packed blocks, the 6N x 6N
 * assembly / damping / solve (solve_normal_equation.h:10-35), the radius and step-evaluator state and the outlier
 * statistics all on the device; the host only reads a done-counter: behind an event, one iteration late, while the next
 * iteration is already queued (sync_every <= 0, the default), or after draining the stream every `sync_every` iterations.
 * Each problem's knots (d_knots_t / d_knots_R, device) are updated in place; d_outlier / num_bad of the input are
 * ignored (flags start cleared, as at the start of a level).  Trace records as mbavo_optimize_trajectory writes
 * them (level = 0), `trace_cap` per problem.  At most 16 control knots per problem (the reference's
 * max_num_ctrl_knots: two 6N x 6N work areas in the 160 KB of LDS); more returns MBAVO_E_ARG.  Returns 0 or an error. */
typedef struct mbavo_lm_batch_opts {
    int spline_deg_k, max_num_iterations, max_consecutive_nonmonotonic_steps, solver_type, sync_every;
    double min_step_quality, min_abs_cost_decrease, max_chi_square_error;
    /* ABI 3 -- zero = default */
    double fast_solve_ratio; /* as mbavo_track_opts.fast_solve_ratio                                                [MBAVO_FAST_SOLVE] */
    double refined_ratio;    /* pivot ratio up to which the stand-in, refined with double-double residuals, is admitted:
                                0: 1e13; < 0: never (plain stand-in only); > 1: that ratio                          [MBAVO_LM_REFINE] */
    int eig;                 /* the wide workgroup (eigenvalue Jacobi as the fallback) for n <= 48; default on       [MBAVO_LM_EIG] */
    int pose_entries;        /* the solve launch writes the candidate's pose entries; default on                     [MBAVO_LM_POSES] */
    int defer_finalize;      /* the LM kernels sum the tile partials themselves; default on                          [MBAVO_LM_DEFER] */
    int retile;              /* a second, finer tiling for the late slots of big batches; default on                 [MBAVO_LM_RETILE] */
    int groups;              /* independent groups on their own streams: 0 = 2 from 384 problems, else 1; 1 .. 8     [MBAVO_LM_GROUPS] */
    int reserved[5];
} mbavo_lm_batch_opts;
typedef struct mbavo_lm_batch_result {
    int iterations, accepted, rejected, invalid, num_outliers, num_trace;
    double initial_cost, final_cost, radius;
} mbavo_lm_batch_result;
int mbavo_lm_batch(mbavo_ctx *ctx, int B, const mbavo_problem *h_problems, const mbavo_lm_batch_opts *opts,
                   mbavo_lm_batch_result *h_results_or_null /*B*/, mbavo_trace_rec *h_trace_or_null /*B x trace_cap*/,
                   int trace_cap);

/* ---- keyframe input producers on device (core/measurements/ImagePyramid.h:59-99,
 * core/image_proc/Gradient.h:16-75) */
int mbavo_pyramid_down_u8(const unsigned char *d_src, int H, int W, unsigned char *d_dst, void *hip_stream);
/* levels 1 .. num_levels-1 (level l: (H0 >> l) x (W0 >> l)) below h_level_ptrs[0] on the context's stream, three levels per launch:
 * the same 2 x 2 box with truncation per level (ImagePyramid.h:59-99); h_level_ptrs: host array of device pointers, num_levels <= 8 */
int mbavo_pyramid_levels_u8(mbavo_ctx *ctx, unsigned char *const *h_level_ptrs, int H0, int W0, int num_levels);
int mbavo_image_gradients_u8(const unsigned char *d_src, int H, int W, float *d_dIxy, void *hip_stream);
/* same gradient image stored as IEEE half pairs (fp16 pyramid, mbavo_problem.grad_fp16 = 1) */
int mbavo_image_gradients_u8_half(const unsigned char *d_src, int H, int W, void *d_dIxy_half, void *hip_stream);
/* packed keyframe (mbavo_problem.grad_fp16 = 2): ONE 32-bit word per pixel -- bits 0-7 the intensity, bits 8-16 and 23-31 the
 * doubled central differences 2 dI/dx, 2 dI/dy of Gradient.h:16-75 as 9-bit two's complement (zero on the 1-pixel border).  Holds
 * exactly what the u8 image and its float gradient image hold (4 bytes per pixel instead of 9); d_ref_dIxy then points at it. */
int mbavo_pack_keyframe_u8(const unsigned char *d_src, int H, int W, void *d_packed /* H*W uint32 */, void *hip_stream);

/* gradient magnitude image (Gradient.h:56-71, the detector's score) */
int mbavo_gradient_magnitude_u8(const unsigned char *d_src, int H, int W, float *d_mag, void *hip_stream);
/* semi-dense keypoints of pyramid level `level` with their depths: candidates = magnitude > score_threshold
 * (core/feature_detectors/FeatureDetectorSemiDense.cpp:27-43), one per grid cell (FeatureDetectorBase.cpp:49-91;
 * cell_H/cell_W <= 0: every candidate, row-major), depth at the level-0 position of the H0 x W0 float map
 * d_depth_z, z < 1e-2 dropped (ba_tracker/blur_aware_direct_tracker.cpp:389-415).  Device in, device out (packed
 * K x 2 doubles + K doubles, at most `cap` written); *h_count = number found.  Synchronous on the context's stream. */
int mbavo_detect_semidense(mbavo_ctx *ctx, const unsigned char *d_img, int H, int W, int level, int H0, int W0,
                           int cell_H, int cell_W, float score_threshold, const float *d_depth_z,
                           double *d_kp_xy, double *d_kp_z, int cap, int *h_count);

/* ---- synthetic blurred frame: synthesize_motion_blurred_img (ba_tracker/generate_synthetic_data.cpp:182-214):
 * mean of `num_samples` warps of the sharp image along the spline over the exposure, on a fronto-parallel plane.
 * Knots are host arrays; d_ref / d_out are device u8 images.  Synchronous. */
int mbavo_synthesize_blur(const unsigned char *d_ref, int H, int W, double plane_depth, const double intrinsics[4],
                          int spline_deg_k, double t0, double dt, const double *h_knots_t, const double *h_knots_R,
                          int N, double cap_time, double exp_time, int num_samples, unsigned char *d_out,
                          void *hip_stream);

/* ---- the caller of the path: BlurAwareDirectTracker (ba_tracker/blur_aware_direct_tracker.{h,cpp}).
 * Poses are 7 doubles: translation, then unit quaternion x,y,z,w (Core::Transformation's layout). */
int mbavo_se3_exp(const double h_tangent[6] /*upsilon, omega*/, double h_pose[7]);   /* Transformation::exp (.cpp:171-177) */
int mbavo_se3_log(const double h_pose[7], double h_tangent[6]);                      /* Transformation::log (.cpp:164-169) */
int mbavo_transform_mul(const double h_A[7], const double h_B[7], double h_out[7]);  /* operator* (.cpp:109-119) */
int mbavo_transform_inverse(const double h_A[7], double h_out[7]);                   /* inverse (.cpp:83-90) */
int mbavo_spline_transform_to(int spline_deg_k, double t0, double dt, double *h_knots_t, double *h_knots_R, int N,
                              double t, const double h_q_xyzw[4], const double h_t[3]); /* Spline.h:183-200 */

typedef struct mbavo_vo_options { /* BlurAwareDirectTrackerOptions (blur_aware_direct_tracker.h:15-67) */
    int H, W, num_pyramid_levels;
    double intrinsics[4];
    int num_virtual_poses_per_frame[8], patch_size[8];
    const int *local_patch_pattern_xy[8]; /* host, (dx,dy) pairs; copied at creation */
    double huber_k;
    int max_consecutive_nonmonotonic_steps, max_num_iterations, solver_type, spline_deg_k;
    double min_step_quality, min_abs_cost_decrease;
    double dt_frame, dt_ctrl_knot, max_chi_square_error;
    double keyframe_max_flow_mag0, keyframe_max_flow_mag1, keyframe_max_flow_mag2, keyframe_max_blur_kernel_mag;
    float score_threshold; int grid_selection_cell_H, grid_selection_cell_W; /* reference: 25 / 30 / 30 (.cpp:353-358) */
    /* ABI 3 -- zero = default; the first three as in mbavo_track_opts */
    double fast_solve_ratio;
    int speculate, persist_levels;
    int keyframe_levels_at_once; /* pyramid, gradients and grid selection of ALL levels in three launches; default on [MBAVO_KF_MULTI] */
    int speculate_keyframe;      /* keyframe pre-processing started under the LM loop, on a second stream, when the predicted motion
                                    already passes the keyframe test (identical results); default on              [MBAVO_KF_SPECULATE] */
    int ride_along;              /* as mbavo_track_opts.ride_along */
    int resum;                   /* as mbavo_track_opts.resum */
    int reserved[2];
} mbavo_vo_options;
typedef struct mbavo_vo_info {
    int is_keyframe, num_keypoints0, num_trace, start_idx;
    double avg_flow, avg_kernel, final_cost;
} mbavo_vo_info;
typedef struct mbavo_vo mbavo_vo;
int mbavo_vo_create(mbavo_ctx *ctx, const mbavo_vo_options *opts, mbavo_vo **out);
int mbavo_vo_destroy(mbavo_vo *vo);
/* getSplineTrajectory()->InsertControlKnot(...) before the first frame (the `get_num_knots() == 0` test at .cpp:99) */
int mbavo_vo_set_spline(mbavo_vo *vo, double t0, double dt, int N, const double *h_knots_t, const double *h_knots_R);
int mbavo_vo_get_spline(mbavo_vo *vo, double *h_t0, double *h_dt, int *h_N, double *h_knots_t /*3*16*/, double *h_knots_R /*4*16*/);
/* LM records of the last mbavo_vo_track_frame's optimizeTrajectory (blur_aware_direct_tracker.cpp:590-924 logs them per
 * iteration), as mbavo_optimize_trajectory writes them, at most 512; returns the number copied (>= 0) or an error (< 0) */
int mbavo_vo_last_trace(mbavo_vo *vo, mbavo_trace_rec *trace, int trace_cap);
/* Checkpoint / resume of a tracker (the reference keeps this state in BlurAwareDirectTracker's members,
 * blur_aware_direct_tracker.h:69-125: mSplineTrajectory, mTKeyframe, mTprevB2W, mNeighFrameVelocity, mPrevTimestamp,
 * mIsFirstFrame): everything trackFrame carries from one frame to the next besides the keyframe's own data; the keyframe
 * (pyramid, gradients, keypoints) is re-made from its sharp frame and depth map by mbavo_vo_set_keyframe =
 * tmpProcessKeyframe (blur_aware_direct_tracker.cpp:342-415).  Poses are 7 doubles t[3], q[4] xyzw, restored bit for bit. */
typedef struct mbavo_vo_state {
    double t0, dt; int N, is_first;
    double knots_t[3 * 16], knots_R[4 * 16];
    double T_keyframe[7], T_prev_b2w[7], velocity[6], prev_timestamp;
} mbavo_vo_state;
int mbavo_vo_get_state(mbavo_vo *vo, mbavo_vo_state *h_state);
int mbavo_vo_set_state(mbavo_vo *vo, const mbavo_vo_state *h_state);
int mbavo_vo_set_keyframe(mbavo_vo *vo, const unsigned char *h_sharp, const float *h_depth_z, double sharp_cap_time);
int mbavo_vo_num_keypoints(mbavo_vo *vo, int level);
int mbavo_vo_get_keypoints(mbavo_vo *vo, int level, double *h_xy /*K*2*/, double *h_z /*K*/);
/* trackFrame (.cpp:88-203): images and the z-depth map of the sharp frame are HOST buffers (H x W), as the
 * reference's Core::Frame holds them; everything after the upload stays on the device. */
int mbavo_vo_track_frame(mbavo_vo *vo, const unsigned char *h_sharp, const float *h_depth_z, double sharp_cap_time,
                         const unsigned char *h_blur, double blur_cap_time, double blur_exp_time,
                         double h_T_out[7], mbavo_vo_info *info_or_null);

/* ---- multi-GPU: one process per GPU; the path shards with no exchange except the final sum of the normal equations
 * (the reference's reduction point: merge_hessian_gradient_cost, ba_tracker/merge_hessian_gradient_cost.cpp:39-86, called
 * at spline_update_step.cpp:232-239).  RCCL is resolved at run time (dlopen), so the library loads on hosts without it. */
/* Shards of one problem (host-side pointer arithmetic on the DEVICE pointers of `whole`; nothing is copied): contiguous
 * keypoint range [K*rank/world, K*(rank+1)/world) -- a band of the keyframe, since the detector emits keypoints row-major --
 * or contiguous frame range [F*rank/world, F*(rank+1)/world).  `shard->num_residuals` is set to the whole problem's count,
 * so the packed blocks of all shards ADD UP to the blocks of the whole problem (keypoint shards: every frame block is a
 * partial sum; frame shards: every rank owns its frames' blocks).  A frame shard may come out empty (F == 0 when
 * world > F): the caller skips its evaluation.  *h_first = first keypoint / frame of the shard (may be NULL). */
int mbavo_shard_keypoints(const mbavo_problem *h_whole, int rank, int world, mbavo_problem *h_shard, int *h_first);
int mbavo_shard_frames(const mbavo_problem *h_whole, int rank, int world, mbavo_problem *h_shard, int *h_first);
/* merge_hessian_gradient_cost (merge_hessian_gradient_cost.cpp:39-86) ON THE DEVICE for B problems: the F_b packed frame
 * blocks of problem b (d_frame_blocks, the layout mbavo_eval_batch writes; h_problems[b] supplies F, N, h_start_idx) are
 * scattered into its normal-equation system [cost | g (6N) | H (6N x 6N, column-major, symmetric)] = mbavo_system_len(N)
 * doubles, systems back to back in problem order.  Frames are added in ascending order (bit-reproducible).  For a
 * keypoint / frame shard (mbavo_shard_*) the result is that shard's PARTIAL system; the sum over ranks
 * (mbavo_allreduce_blocks) is the whole problem's system.  Asynchronous on the context's stream. */
int mbavo_system_len(int N); /* 1 + 6N + 36N^2 */
int mbavo_merge_device(mbavo_ctx *ctx, int B, const mbavo_problem *h_problems, int spline_deg_k,
                       const double *d_frame_blocks, double *d_systems);
/* The context's own RCCL communicator: rank 0 calls mbavo_comm_unique_id (ncclGetUniqueId) and hands the bytes to the
 * other ranks through any side channel (bench.py: a torch.distributed broadcast); every rank then calls mbavo_comm_init
 * (ncclCommInitRank on the context's device; collective).  mbavo_comm_ranks = ncclCommCount (0: no communicator). */
#define MBAVO_COMM_ID_BYTES 128
int mbavo_comm_unique_id(unsigned char h_id[MBAVO_COMM_ID_BYTES]);
int mbavo_comm_init(mbavo_ctx *ctx, const unsigned char h_id[MBAVO_COMM_ID_BYTES], int rank, int world);
int mbavo_comm_ranks(mbavo_ctx *ctx);
int mbavo_comm_destroy(mbavo_ctx *ctx);
/* In-place sum over all ranks of `count` doubles (packed blocks or merged systems) with ONE ncclAllReduce on the context's
 * stream, ordered after the kernels that wrote them (RCCL over xGMI).  `rccl_comm` = a caller-owned ncclComm_t, or NULL
 * for the context's own communicator (mbavo_comm_init); with neither: MBAVO_E_ARG. */
int mbavo_allreduce_blocks(mbavo_ctx *ctx, void *rccl_comm, double *d_blocks, long long count);
/* The same sum out of place (d_recv = sum over ranks of d_send).  Independent keyframe pairs sharded pair -> rank
 * (SURVEY.md 8e(1)): every rank evaluates its pairs straight into ITS slice of a send buffer whose other slices stay zero
 * -- nothing else ever writes them -- so the slices are disjoint, the sum is exact (x + 0 + ... + 0) and no buffer has to
 * be cleared between iterations. */
int mbavo_allreduce_blocks_to(mbavo_ctx *ctx, void *rccl_comm, const double *d_send, double *d_recv, long long count);
/* In-place ncclAllGather of equal slices: rank r's `count_per_rank` doubles sit at d_blocks + r * count_per_rank on entry,
 * every rank holds all slices on return (half the all-reduce's traffic for disjoint slices; needs equal slice lengths). */
int mbavo_allgather_blocks(mbavo_ctx *ctx, void *rccl_comm, double *d_blocks, long long count_per_rank);

/* ---- the same two collectives WITHOUT RCCL, sized for this path's messages (19 KB of one joint system ... 1.3 MB of 512 packed
 * blocks: latency-bound, where a ring pays a hop per rank): every rank maps every peer's receive region (hipIpcGetMemHandle /
 * hipIpcOpenMemHandle) and a collective is ONE kernel on the context's stream -- store this rank's slice into every peer's
 * region, raise a per-step flag there, wait for the peers' flags in the own region, then copy the slots out (all-gather) or
 * add them in rank order (all-reduce: every rank gets the same bits).  Replaces the exchange at the reference's reduction
 * point, merge_hessian_gradient_cost.cpp:39-86 / spline_update_step.cpp:232-239, like mbavo_allreduce_blocks above.
 *   mbavo_p2p_create   allocates this rank's region (2 x world slots of max_doubles_per_slot doubles: the largest
 *                      count_per_rank of an all-gather / count of an all-reduce that will be asked for) and returns its
 *                      64-byte IPC handle; the caller hands every rank's handle to every rank (all-gather of 64 bytes, as
 *                      with the RCCL id) and calls
 *   mbavo_p2p_connect  with the world x 64 bytes in rank order.  At most 16 ranks, all on GPUs of one node (ranks may share a GPU).
 *   Every rank must issue the same sequence of p2p collectives.  A peer that does not arrive within 20 s (mbavo_p2p_set_timeout:
 *   any time in (0, 3600] s, for the collectives enqueued after it) ends the kernel: what it would have written -- the peers'
 *   slices of an all-gather, the vector of an all-reduce -- is filled with NaN, so that an unreduced buffer cannot pass for a
 *   result, and mbavo_p2p_status (synchronises the stream) returns MBAVO_E_TIMEOUT from then on (sticky: the ranks' sequence
 *   numbers no longer agree; tear the regions down and create them again).
 *   Tear-down is two-phase, like any shared mapping: every rank finishes its last collective and calls mbavo_p2p_disconnect
 *   (unmaps the PEERS' regions); after a barrier of the caller -- nobody maps anybody any more -- every rank calls
 *   mbavo_p2p_destroy (frees its OWN region; also what mbavo_destroy does).  Freeing a region a peer still maps makes a later
 *   mbavo_p2p_create fail in hipIpcGetMemHandle (seen intermittently with three ranks when tear-down was one call). */
#define MBAVO_P2P_HANDLE_BYTES 64
int mbavo_p2p_create(mbavo_ctx *ctx, int rank, int world, long long max_doubles_per_slot, unsigned char *h_handle_out /*64*/);
int mbavo_p2p_connect(mbavo_ctx *ctx, const unsigned char *h_all_handles /* world x 64, rank order */);
int mbavo_p2p_ranks(mbavo_ctx *ctx); /* world once connected, else 0 */
int mbavo_allgather_blocks_p2p(mbavo_ctx *ctx, double *d_blocks, long long count_per_rank); /* in place, as mbavo_allgather_blocks */
int mbavo_allreduce_blocks_p2p(mbavo_ctx *ctx, double *d_blocks, long long count);          /* in place, as mbavo_allreduce_blocks */
int mbavo_p2p_status(mbavo_ctx *ctx);
int mbavo_p2p_set_timeout(mbavo_ctx *ctx, double seconds);
int mbavo_p2p_disconnect(mbavo_ctx *ctx);
int mbavo_p2p_destroy(mbavo_ctx *ctx);

/* ---- measurement: HIP-event timing of the dominant kernel (the fused residual/Jacobian/JtJ
 * kernel) on the context's stream, attached to the kernel's own dispatch (hipExtLaunchKernelGGL: begin / end
 * timestamps of the kernel).  enable = n > 0 starts a fresh collection that times every n-th launch (a timed
 * launch costs a few us of launch gap, so timing every launch would slow the measured region itself);
 * 0 stops.  read returns the summed duration in ms and the number of timed launches (blocks until the recorded
 * events completed). */
int mbavo_profile(mbavo_ctx *ctx, int enable);
int mbavo_profile_read(mbavo_ctx *ctx, double *h_fused_ms_sum, int *h_launches);
/* name of the dominant kernel the context's last evaluation dispatched, e.g. "k_fused<4,true,false>" (labels the timing) */
const char *mbavo_last_kernel(mbavo_ctx *ctx);
/* host-side phase timers of the tracking loop (enabled by MBAVO_TIMING=1 in the environment): print the totals since
 * the last report to stderr and reset them.  Development aid; a no-op when the timers are off. */
void mbavo_timing_report(void);
/* The MBAVO_* override variables of the A/B tools (csrc/options.h) are read ONCE per process; a tool that changes one inside a
 * running process calls this afterwards.  Not for concurrent use with running API calls. */
void mbavo_reload_env(void);
/* process-wide counters of the host LM loop's ride-along evaluations (mbavo_track_opts.ride_along) since the last call:
 * out[0] commands that carried one, out[1] pyramid levels that started on theirs, out[2] levels that waited a wasted one
 * out before their first command (it shares the level's ticket counters and partials with that command).  Reset on read. */
void mbavo_ride_along_stats(long long out[3]);

const char *mbavo_version(void);
/* revision of the POD structs above (MBAVO_ABI_VERSION of the header the library was built with): compare before use */
int mbavo_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
