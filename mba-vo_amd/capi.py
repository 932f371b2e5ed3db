"""ctypes binding of libmbavo.so (include/mbavo.h).  No CPU fallback: if the library is
missing or no HIP device is usable, the compute entry points raise."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmbavo.so")
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
vp = C.c_void_p


class Problem(C.Structure):
    """struct mbavo_problem"""
    _fields_ = [
        ("S", C.c_int), ("F", C.c_int), ("K", C.c_int), ("P", C.c_int), ("N", C.c_int),
        ("H", C.c_int), ("W", C.c_int),
        ("d_ref_img", vp), ("d_ref_dIxy", vp), ("d_cur_imgs", vp),
        ("d_kp_xy", vp), ("kp_stride", C.c_int), ("d_kp_z", vp), ("d_pattern", vp),
        ("d_outlier", vp), ("num_bad", C.c_int), ("intrinsics", C.c_double * 4),
        ("d_cap_time", vp), ("d_exp_time", vp), ("t0", C.c_double), ("dt", C.c_double),
        ("d_knots_t", vp), ("d_knots_R", vp), ("h_start_idx", c_ip), ("huber_a", C.c_double),
        ("grad_fp16", C.c_int), ("num_residuals", C.c_longlong),
    ]


class Level(C.Structure):
    """struct mbavo_level"""
    _fields_ = [
        ("H", C.c_int), ("W", C.c_int), ("K", C.c_int), ("P", C.c_int), ("S", C.c_int),
        ("d_ref_img", vp), ("d_ref_dIxy", vp), ("d_cur_imgs", vp),
        ("d_kp_xy", vp), ("d_kp_z", vp), ("d_pattern", vp),
    ]


class TrackOpts(C.Structure):
    """struct mbavo_track_opts"""
    _fields_ = [
        ("num_levels", C.c_int), ("spline_deg_k", C.c_int), ("max_num_iterations", C.c_int),
        ("max_consecutive_nonmonotonic_steps", C.c_int), ("solver_type", C.c_int),
        ("intrinsics", C.c_double * 4), ("huber_k", C.c_double), ("min_step_quality", C.c_double),
        ("min_abs_cost_decrease", C.c_double), ("max_chi_square_error", C.c_double),
        # ABI 3: zero = default (include/mbavo.h)
        ("fast_solve_ratio", C.c_double), ("speculate", C.c_int), ("persist_levels", C.c_int), ("ride_along", C.c_int),
        ("resum", C.c_int), ("reserved", C.c_int * 2),
    ]


class EngineOpts(C.Structure):
    """struct mbavo_engine_opts (tri-state flags: 0 default, 1 on, -1 off; numbers: 0 default)"""
    _fields_ = [("sample_parallel", C.c_int), ("single_launch", C.c_int), ("fused_pose", C.c_int), ("fused_pose_max_samples", C.c_int),
                ("persistent", C.c_int), ("prelaunch", C.c_int), ("tiles_per_cu", C.c_int), ("min_tile_pixels", C.c_int),
                ("sp_max_slot_tiles", C.c_int), ("reserved", C.c_int * 7)]


class VoState(C.Structure):
    """struct mbavo_vo_state"""
    _fields_ = [("t0", C.c_double), ("dt", C.c_double), ("N", C.c_int), ("is_first", C.c_int),
                ("knots_t", C.c_double * 48), ("knots_R", C.c_double * 64),
                ("T_keyframe", C.c_double * 7), ("T_prev_b2w", C.c_double * 7), ("velocity", C.c_double * 6),
                ("prev_timestamp", C.c_double)]


class TraceRec(C.Structure):
    """struct mbavo_trace_rec"""
    _fields_ = [
        ("level", C.c_int), ("iter", C.c_int), ("kind", C.c_int), ("num_outliers", C.c_int),
        ("radius", C.c_double), ("eval_cost", C.c_double), ("candidate_cost", C.c_double),
        ("model_change", C.c_double), ("quality", C.c_double),
    ]


class LmBatchOpts(C.Structure):
    """struct mbavo_lm_batch_opts"""
    _fields_ = [("spline_deg_k", C.c_int), ("max_num_iterations", C.c_int), ("max_consecutive_nonmonotonic_steps", C.c_int),
                ("solver_type", C.c_int), ("sync_every", C.c_int), ("min_step_quality", C.c_double),
                ("min_abs_cost_decrease", C.c_double), ("max_chi_square_error", C.c_double),
                # ABI 3: zero = default (include/mbavo.h)
                ("fast_solve_ratio", C.c_double), ("refined_ratio", C.c_double), ("eig", C.c_int), ("pose_entries", C.c_int),
                ("defer_finalize", C.c_int), ("retile", C.c_int), ("groups", C.c_int), ("reserved", C.c_int * 5)]


class LmBatchResult(C.Structure):
    """struct mbavo_lm_batch_result"""
    _fields_ = [("iterations", C.c_int), ("accepted", C.c_int), ("rejected", C.c_int), ("invalid", C.c_int),
                ("num_outliers", C.c_int), ("num_trace", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("radius", C.c_double)]


class VoOptions(C.Structure):
    """struct mbavo_vo_options"""
    _fields_ = [
        ("H", C.c_int), ("W", C.c_int), ("num_pyramid_levels", C.c_int), ("intrinsics", C.c_double * 4),
        ("num_virtual_poses_per_frame", C.c_int * 8), ("patch_size", C.c_int * 8),
        ("local_patch_pattern_xy", c_ip * 8), ("huber_k", C.c_double),
        ("max_consecutive_nonmonotonic_steps", C.c_int), ("max_num_iterations", C.c_int), ("solver_type", C.c_int),
        ("spline_deg_k", C.c_int), ("min_step_quality", C.c_double), ("min_abs_cost_decrease", C.c_double),
        ("dt_frame", C.c_double), ("dt_ctrl_knot", C.c_double), ("max_chi_square_error", C.c_double),
        ("keyframe_max_flow_mag0", C.c_double), ("keyframe_max_flow_mag1", C.c_double),
        ("keyframe_max_flow_mag2", C.c_double), ("keyframe_max_blur_kernel_mag", C.c_double),
        ("score_threshold", C.c_float), ("grid_selection_cell_H", C.c_int), ("grid_selection_cell_W", C.c_int),
        # ABI 3: zero = default (include/mbavo.h)
        ("fast_solve_ratio", C.c_double), ("speculate", C.c_int), ("persist_levels", C.c_int), ("keyframe_levels_at_once", C.c_int),
        ("speculate_keyframe", C.c_int), ("ride_along", C.c_int), ("resum", C.c_int), ("reserved", C.c_int * 2),
    ]


class VoInfo(C.Structure):
    """struct mbavo_vo_info"""
    _fields_ = [("is_keyframe", C.c_int), ("num_keypoints0", C.c_int), ("num_trace", C.c_int), ("start_idx", C.c_int),
                ("avg_flow", C.c_double), ("avg_kernel", C.c_double), ("final_cost", C.c_double)]


# every symbol include/mbavo.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "mbavo_create", "mbavo_destroy", "mbavo_set_stream", "mbavo_packed_len", "mbavo_eval_batch", "mbavo_eval",
    "mbavo_compute_virtual_camera_poses", "mbavo_compute_local_patches_xy", "mbavo_compute_pixel_jacobian_residual",
    "mbavo_compute_patch_cost_gradient_hessian", "mbavo_compute_frame_cost_gradient_hessian",
    "mbavo_merge_hessian_gradient_cost", "mbavo_merge_host", "mbavo_solve_normal_equation",
    "mbavo_lm_new", "mbavo_lm_delete", "mbavo_lm_reset", "mbavo_lm_step_accepted", "mbavo_lm_step_rejected",
    "mbavo_lm_get_radius", "mbavo_tr_new", "mbavo_tr_delete", "mbavo_tr_reset", "mbavo_tr_step_quality",
    "mbavo_tr_step_accepted", "mbavo_spline_get_pose", "mbavo_spline_plus", "mbavo_segment_start_index",
    "mbavo_optimize_trajectory", "mbavo_pyramid_down_u8", "mbavo_pyramid_levels_u8", "mbavo_image_gradients_u8", "mbavo_image_gradients_u8_half", "mbavo_pack_keyframe_u8", "mbavo_synthesize_blur", "mbavo_allreduce_blocks", "mbavo_allreduce_blocks_to", "mbavo_allgather_blocks",
    "mbavo_profile", "mbavo_profile_read", "mbavo_version", "mbavo_abi_version",
    "mbavo_gradient_magnitude_u8", "mbavo_detect_semidense", "mbavo_se3_exp", "mbavo_se3_log", "mbavo_transform_mul",
    "mbavo_transform_inverse", "mbavo_spline_transform_to", "mbavo_vo_create", "mbavo_vo_destroy", "mbavo_vo_set_spline",
    "mbavo_vo_get_spline", "mbavo_sizeof", "mbavo_set_engine_opts", "mbavo_get_engine_opts", "mbavo_eval_batch_merged", "mbavo_p2p_create", "mbavo_p2p_connect", "mbavo_p2p_ranks", "mbavo_allgather_blocks_p2p",
    "mbavo_allreduce_blocks_p2p", "mbavo_p2p_status", "mbavo_p2p_disconnect", "mbavo_p2p_destroy", "mbavo_vo_last_trace", "mbavo_vo_get_state", "mbavo_vo_set_state", "mbavo_vo_set_keyframe", "mbavo_vo_num_keypoints", "mbavo_vo_get_keypoints", "mbavo_vo_track_frame", "mbavo_lm_batch",
    "mbavo_shard_keypoints", "mbavo_shard_frames", "mbavo_system_len", "mbavo_merge_device", "mbavo_comm_unique_id",
    "mbavo_comm_init", "mbavo_comm_ranks", "mbavo_comm_destroy", "mbavo_last_kernel", "mbavo_timing_report",
    "mbavo_ride_along_stats", "mbavo_p2p_set_timeout", "mbavo_reload_env",
]


KERNEL_SOURCES = ("csrc/engine.hip", "csrc/pixel_math.h", "csrc/se3_math.h", "csrc/pose_entries.h")


def kernel_source_sha():
    """sha256[:16] over the sources of the evaluation kernels: stamped into the committed counter extracts
    (profiles/*_pmc_*.json, *_hbm_counters*.json) when they are collected and compared by bench.py, which flags figures
    read from an extract of another kernel revision as stale."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(_HERE, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(quiet=True):
    """hipcc --offload-arch=gfx950 build of libmbavo.so (cross-compiles without a GPU)."""
    subprocess.run(["bash", os.path.join(_HERE, "build.sh")], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


def load():
    """Load libmbavo.so; raises if it has not been built (the product has no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmbavo.so is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  It must be the first HIP runtime in
    # the process, so that libmbavo.so binds to the SAME runtime instance as the torch tensors / streams it is
    # handed; two runtimes in one process do not see each other's devices or allocations.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.mbavo_version.restype = C.c_char_p
    L.mbavo_ride_along_stats.argtypes = [C.POINTER(C.c_longlong)]
    L.mbavo_ride_along_stats.restype = None
    L.mbavo_reload_env.restype = None
    L.mbavo_p2p_set_timeout.argtypes = [vp, C.c_double]
    L.mbavo_create.argtypes = [C.POINTER(vp), C.c_int]
    L.mbavo_destroy.argtypes = [vp]
    L.mbavo_set_stream.argtypes = [vp, vp]
    L.mbavo_packed_len.argtypes = [C.c_int]
    L.mbavo_eval_batch.argtypes = [vp, C.c_int, C.POINTER(Problem), C.c_int, C.c_int, vp, vp, vp]
    L.mbavo_set_engine_opts.argtypes = [vp, C.POINTER(EngineOpts)]
    L.mbavo_get_engine_opts.argtypes = [vp, C.POINTER(EngineOpts)]
    L.mbavo_eval_batch_merged.argtypes = [vp, C.c_int, C.POINTER(Problem), C.c_int, vp, vp, vp, vp]
    L.mbavo_eval.argtypes = [vp, C.POINTER(Problem), C.c_int, c_dp, c_dp, c_dp, vp]
    L.mbavo_compute_virtual_camera_poses.argtypes = [C.c_int, C.c_int, vp, vp, C.c_int, C.c_double, C.c_double,
                                                     vp, vp, vp, vp, vp]
    L.mbavo_compute_local_patches_xy.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, c_dp, c_ip, vp]
    L.mbavo_compute_pixel_jacobian_residual.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp,
                                                        C.c_int, vp, C.c_int, c_dp, c_ip, vp, vp]
    L.mbavo_compute_patch_cost_gradient_hessian.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_double,
                                                            C.c_double, vp]
    L.mbavo_compute_frame_cost_gradient_hessian.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp]
    L.mbavo_merge_hessian_gradient_cost.argtypes = [C.c_int, C.c_int, vp, c_ip, C.c_int, c_dp, c_dp, c_dp]
    L.mbavo_merge_host.argtypes = [C.c_int, C.c_int, c_dp, c_ip, C.c_int, c_dp, c_dp, c_dp]
    L.mbavo_solve_normal_equation.argtypes = [c_dp, c_dp, C.c_int, C.c_int, c_dp]
    L.mbavo_lm_new.restype = vp
    for n in ("mbavo_lm_delete", "mbavo_lm_reset", "mbavo_lm_step_rejected"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = None
    L.mbavo_lm_step_accepted.argtypes = [vp, C.c_double]
    L.mbavo_lm_step_accepted.restype = None
    L.mbavo_lm_get_radius.argtypes = [vp]
    L.mbavo_lm_get_radius.restype = C.c_double
    L.mbavo_tr_new.restype = vp
    L.mbavo_tr_new.argtypes = [C.c_int]
    L.mbavo_tr_delete.argtypes = [vp]
    L.mbavo_tr_delete.restype = None
    L.mbavo_tr_reset.argtypes = [vp, C.c_double]
    L.mbavo_tr_reset.restype = None
    L.mbavo_tr_step_quality.argtypes = [vp, C.c_double, C.c_double]
    L.mbavo_tr_step_quality.restype = C.c_double
    L.mbavo_tr_step_accepted.argtypes = [vp, C.c_double, C.c_double]
    L.mbavo_tr_step_accepted.restype = None
    L.mbavo_spline_get_pose.argtypes = [C.c_int, C.c_double, C.c_double, c_dp, c_dp, C.c_int, C.c_double,
                                        c_dp, c_dp, c_dp, c_dp]
    L.mbavo_spline_plus.argtypes = [c_dp, c_dp, C.c_int, c_dp, c_dp, c_dp]
    L.mbavo_segment_start_index.argtypes = [C.c_double, C.c_double, C.c_double]
    L.mbavo_optimize_trajectory.argtypes = [vp, C.POINTER(TrackOpts), C.POINTER(Level), C.c_int, c_dp, c_dp,
                                            C.c_double, C.c_double, c_dp, c_dp, C.c_int, c_ip, c_dp,
                                            C.POINTER(TraceRec), C.c_int]
    L.mbavo_pyramid_down_u8.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.mbavo_image_gradients_u8.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.mbavo_image_gradients_u8_half.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.mbavo_pack_keyframe_u8.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.mbavo_pyramid_levels_u8.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int]
    L.mbavo_synthesize_blur.argtypes = [vp, C.c_int, C.c_int, C.c_double, c_dp, C.c_int, C.c_double, C.c_double, c_dp,
                                        c_dp, C.c_int, C.c_double, C.c_double, C.c_int, vp, vp]
    L.mbavo_allreduce_blocks.argtypes = [vp, vp, vp, C.c_longlong]
    L.mbavo_allreduce_blocks_to.argtypes = [vp, vp, vp, vp, C.c_longlong]
    L.mbavo_allgather_blocks.argtypes = [vp, vp, vp, C.c_longlong]
    L.mbavo_shard_keypoints.argtypes = [C.POINTER(Problem), C.c_int, C.c_int, C.POINTER(Problem), c_ip]
    L.mbavo_shard_frames.argtypes = [C.POINTER(Problem), C.c_int, C.c_int, C.POINTER(Problem), c_ip]
    L.mbavo_system_len.argtypes = [C.c_int]
    L.mbavo_merge_device.argtypes = [vp, C.c_int, C.POINTER(Problem), C.c_int, vp, vp]
    L.mbavo_comm_unique_id.argtypes = [C.c_char_p]
    L.mbavo_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.mbavo_comm_ranks.argtypes = [vp]
    L.mbavo_comm_destroy.argtypes = [vp]
    L.mbavo_last_kernel.argtypes = [vp]
    L.mbavo_last_kernel.restype = C.c_char_p
    L.mbavo_timing_report.restype = None
    L.mbavo_gradient_magnitude_u8.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.mbavo_detect_semidense.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                         vp, vp, vp, C.c_int, c_ip]
    L.mbavo_se3_exp.argtypes = [c_dp, c_dp]
    L.mbavo_se3_log.argtypes = [c_dp, c_dp]
    L.mbavo_transform_mul.argtypes = [c_dp, c_dp, c_dp]
    L.mbavo_transform_inverse.argtypes = [c_dp, c_dp]
    L.mbavo_spline_transform_to.argtypes = [C.c_int, C.c_double, C.c_double, c_dp, c_dp, C.c_int, C.c_double, c_dp, c_dp]
    L.mbavo_vo_create.argtypes = [vp, C.POINTER(VoOptions), C.POINTER(vp)]
    L.mbavo_vo_destroy.argtypes = [vp]
    L.mbavo_vo_set_spline.argtypes = [vp, C.c_double, C.c_double, C.c_int, c_dp, c_dp]
    L.mbavo_vo_get_spline.argtypes = [vp, c_dp, c_dp, c_ip, c_dp, c_dp]
    L.mbavo_vo_num_keypoints.argtypes = [vp, C.c_int]
    L.mbavo_vo_last_trace.argtypes = [vp, C.POINTER(TraceRec), C.c_int]
    L.mbavo_vo_get_state.argtypes = [vp, C.POINTER(VoState)]
    L.mbavo_vo_set_state.argtypes = [vp, C.POINTER(VoState)]
    L.mbavo_vo_set_keyframe.argtypes = [vp, vp, vp, C.c_double]
    L.mbavo_vo_get_keypoints.argtypes = [vp, C.c_int, c_dp, c_dp]
    L.mbavo_vo_track_frame.argtypes = [vp, vp, vp, C.c_double, vp, C.c_double, C.c_double, c_dp, C.POINTER(VoInfo)]
    L.mbavo_lm_batch.argtypes = [vp, C.c_int, C.POINTER(Problem), C.POINTER(LmBatchOpts), C.POINTER(LmBatchResult),
                                 C.POINTER(TraceRec), C.c_int]
    L.mbavo_p2p_create.argtypes = [vp, C.c_int, C.c_int, C.c_longlong, C.c_char_p]
    L.mbavo_p2p_connect.argtypes = [vp, C.c_char_p]
    L.mbavo_p2p_ranks.argtypes = [vp]
    L.mbavo_allgather_blocks_p2p.argtypes = [vp, vp, C.c_longlong]
    L.mbavo_allreduce_blocks_p2p.argtypes = [vp, vp, C.c_longlong]
    L.mbavo_p2p_status.argtypes = [vp]
    L.mbavo_p2p_disconnect.argtypes = [vp]
    L.mbavo_p2p_destroy.argtypes = [vp]
    L.mbavo_profile.argtypes = [vp, C.c_int]
    L.mbavo_profile_read.argtypes = [vp, c_dp, c_ip]
    _LIB = L
    return L


def dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def ip(a):
    return None if a is None else a.ctypes.data_as(c_ip)


def check(rc, what="mbavo call"):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


class Context:
    """mbavo_ctx wrapper."""

    def __init__(self, device_id=0, stream=None):
        self.lib = load()
        self.handle = vp()
        check(self.lib.mbavo_create(C.byref(self.handle), int(device_id)), "mbavo_create")
        self.device_id, self.stream = int(device_id), stream  # (None / 0: the null stream)
        if stream is not None:
            check(self.lib.mbavo_set_stream(self.handle, vp(stream)), "mbavo_set_stream")

    def engine_opts(self, **kw):
        """mbavo_set_engine_opts: the named fields (tri-state flags 1 / -1, numbers), every other field at its default."""
        o = EngineOpts()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, int(v))
        check(self.lib.mbavo_set_engine_opts(self.handle, C.byref(o)), "mbavo_set_engine_opts")

    def close(self):
        if self.handle:
            self.lib.mbavo_destroy(self.handle)
            self.handle = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
