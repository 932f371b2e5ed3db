"""ctypes binding of the CPU oracle (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  `lib()` is the plain-C restatement (oracle/mbavo_oracle.c);
`ref()` is the reference's own compilable sources (oracle/_ref, built in the
build container only; None when absent).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_fp = C.POINTER(C.c_float)
c_u8p = C.POINTER(C.c_ubyte)


def dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def ip(a):
    return None if a is None else a.ctypes.data_as(c_ip)


def fp(a):
    return None if a is None else a.ctypes.data_as(c_fp)


def u8p(a):
    return None if a is None else a.ctypes.data_as(c_u8p)


class OrcProblem(C.Structure):
    _fields_ = [
        ("S", C.c_int), ("F", C.c_int), ("K", C.c_int), ("P", C.c_int),
        ("k", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("ref_img", c_u8p), ("ref_dIxy", c_fp), ("cur_imgs", C.POINTER(c_u8p)),
        ("kp_xy", c_dp), ("kp_z", c_dp), ("pattern", c_ip), ("outlier", c_u8p),
        ("num_bad", C.c_int), ("intr", C.c_double * 4),
        ("cap", c_dp), ("exp_t", c_dp), ("t0", C.c_double), ("dt", C.c_double),
        ("knots_t", c_dp), ("knots_R", c_dp), ("start_idx", c_ip), ("huber_a", C.c_double),
    ]


class OrcLevel(C.Structure):
    _fields_ = [
        ("H", C.c_int), ("W", C.c_int), ("K", C.c_int), ("P", C.c_int), ("S", C.c_int),
        ("ref_img", c_u8p), ("ref_dIxy", c_fp), ("cur_imgs", C.POINTER(c_u8p)),
        ("kp_xy", c_dp), ("kp_z", c_dp), ("pattern", c_ip),
    ]


class OrcVoOpts(C.Structure):
    _fields_ = [
        ("H", C.c_int), ("W", C.c_int), ("num_levels", C.c_int), ("intr", C.c_double * 4),
        ("num_virtual_poses", C.c_int * 8), ("patch_size", C.c_int * 8), ("pattern_xy", c_ip * 8),
        ("huber_k", C.c_double),
        ("max_nonmono", C.c_int), ("max_num_iterations", C.c_int), ("solver_type", C.c_int), ("spline_deg_k", C.c_int),
        ("min_step_quality", C.c_double), ("min_abs_cost_decrease", C.c_double),
        ("dt_frame", C.c_double), ("dt_ctrl_knot", C.c_double), ("max_chi_square_error", C.c_double),
        ("keyframe_max_flow_mag0", C.c_double), ("keyframe_max_flow_mag1", C.c_double),
        ("keyframe_max_flow_mag2", C.c_double), ("keyframe_max_blur_kernel_mag", C.c_double),
        ("score_threshold", C.c_float), ("grid_cell_H", C.c_int), ("grid_cell_W", C.c_int),
    ]


class OrcVoInfo(C.Structure):
    _fields_ = [("is_keyframe", C.c_int), ("num_keypoints0", C.c_int), ("num_trace", C.c_int), ("start_idx", C.c_int),
                ("avg_flow", C.c_double), ("avg_kernel", C.c_double), ("final_cost", C.c_double)]


class OrcVoState(C.Structure):
    _fields_ = [("t0", C.c_double), ("dt", C.c_double), ("N", C.c_int), ("is_first", C.c_int),
                ("knots_t", C.c_double * 48), ("knots_R", C.c_double * 64),
                ("T_keyframe", C.c_double * 7), ("T_prev_b2w", C.c_double * 7), ("velocity", C.c_double * 6),
                ("prev_timestamp", C.c_double)]


class OrcTrackOpts(C.Structure):
    _fields_ = [
        ("num_levels", C.c_int), ("k", C.c_int), ("max_num_iterations", C.c_int),
        ("max_nonmono", C.c_int), ("solver_type", C.c_int), ("intr", C.c_double * 4),
        ("huber_k", C.c_double), ("min_step_quality", C.c_double),
        ("min_abs_cost_decrease", C.c_double), ("max_chi_square_error", C.c_double),
    ]


class OrcTraceRec(C.Structure):
    _fields_ = [
        ("level", C.c_int), ("iter", C.c_int), ("kind", C.c_int), ("num_outliers", C.c_int),
        ("radius", C.c_double), ("eval_cost", C.c_double), ("candidate_cost", C.c_double),
        ("model_change", C.c_double), ("quality", C.c_double),
    ]


class OrcLm(C.Structure):
    _fields_ = [("radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
                ("decrease_factor", C.c_double)]


class OrcTr(C.Structure):
    _fields_ = [("max_nonmono", C.c_int), ("minimum_cost", C.c_double), ("current_cost", C.c_double),
                ("reference_cost", C.c_double), ("candidate_cost", C.c_double),
                ("acc_ref", C.c_double), ("acc_cand", C.c_double), ("num_nonmono", C.c_int)]


def build(quiet=True):
    """Compile the C restatement and, when /root/reference is present, oracle/_ref."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _bind(path):
    if True:
        L = C.CDLL(path)
        L.orc_tr_quality.restype = C.c_double
        L.orc_optimize_trajectory.restype = C.c_int
        L.orc_solve_normal_equation.restype = C.c_int
        L.orc_bilinear.restype = C.c_int
        L.orc_pixel_intensity.restype = C.c_int
        L.orc_bilinear.argtypes = [c_u8p, c_fp, C.c_int, C.c_int, C.c_double, C.c_double, c_dp]
        L.orc_pixel_intensity.argtypes = [c_u8p, c_fp, C.c_int, C.c_int, c_dp, c_dp, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_double, c_dp, c_dp]
        L.orc_spline_segment.argtypes = [C.c_double, C.c_double, C.c_double, c_ip, c_dp]
        for n in ("orc_c2_vec3", "orc_c4_vec3", "orc_c2_rot3", "orc_c4_rot3"):
            getattr(L, n).argtypes = [c_dp, C.c_double, c_dp, c_dp]
        L.orc_compute_virtual_camera_poses.argtypes = [
            C.c_int, C.c_int, c_dp, c_dp, C.c_int, C.c_double, C.c_double, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip]
        L.orc_compute_local_patches_xy.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_int, c_dp, c_dp]
        L.orc_compute_pixel_jacobian_residual.argtypes = [
            c_u8p, c_fp, C.POINTER(c_u8p), C.c_int, C.c_int, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp, C.c_int,
            c_ip, C.c_int, c_dp, C.c_int, C.c_int, c_dp, c_dp]
        L.orc_compute_patch_cost_gradient_hessian.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, C.c_double, C.c_double, c_dp]
        L.orc_compute_frame_cost_gradient_hessian.argtypes = [
            C.c_int, C.c_int, C.c_int, c_dp, C.c_int, c_u8p, c_dp]
        L.orc_merge_hessian_gradient_cost.argtypes = [C.c_int, C.c_int, c_dp, c_ip, C.c_int, c_dp, c_dp, c_dp]
        L.orc_evaluate.argtypes = [C.POINTER(OrcProblem), c_dp, c_dp, c_dp, c_dp, c_dp]
        L.orc_evaluate_fast.argtypes = [C.POINTER(OrcProblem), C.c_int, c_dp, c_dp, c_dp, c_dp]
        L.orc_count_valid.argtypes = [C.POINTER(OrcProblem), C.c_int, c_dp]
        L.orc_solve_normal_equation.argtypes = [c_dp, c_dp, C.c_int, C.c_int, c_dp]
        L.orc_tr_quality.argtypes = [C.POINTER(OrcTr), C.c_double, C.c_double]
        L.orc_tr_reset.argtypes = [C.POINTER(OrcTr), C.c_double]
        L.orc_tr_accepted.argtypes = [C.POINTER(OrcTr), C.c_double, C.c_double]
        L.orc_tr_init.argtypes = [C.POINTER(OrcTr), C.c_int]
        L.orc_lm_accepted.argtypes = [C.POINTER(OrcLm), C.c_double]
        L.orc_plus_t.argtypes = [c_dp, c_dp, C.c_int, c_dp]
        L.orc_plus_R.argtypes = [c_dp, c_dp, C.c_int, c_dp]
        L.orc_pyramid_down_u8.argtypes = [c_u8p, C.c_int, C.c_int, c_u8p]
        L.orc_image_gradients_u8.argtypes = [c_u8p, C.c_int, C.c_int, c_fp, c_fp]
        L.orc_warp_image.argtypes = [c_u8p, C.c_int, C.c_int, c_dp, c_dp, C.c_double, c_dp, c_u8p]
        L.orc_synthesize_blur.argtypes = [c_u8p, C.c_int, C.c_int, C.c_double, c_dp, C.c_int, C.c_double,
                                          C.c_double, c_dp, c_dp, C.c_double, C.c_double, C.c_int, c_u8p]
        L.orc_optimize_trajectory.argtypes = [
            C.POINTER(OrcTrackOpts), C.POINTER(OrcLevel), C.c_int, c_dp, c_dp, C.c_double, C.c_double,
            c_dp, c_dp, C.c_int, c_ip, c_dp, C.POINTER(OrcTraceRec), C.c_int]
        L.orc_transform_mul.argtypes = [c_dp, c_dp, c_dp]
        L.orc_transform_inverse.argtypes = [c_dp, c_dp]
        L.orc_se3_exp.argtypes = [c_dp, c_dp, c_dp]
        L.orc_se3_log.argtypes = [c_dp, c_dp, c_dp]
        L.orc_spline_get_pose.argtypes = [C.c_int, C.c_double, C.c_double, c_dp, c_dp, C.c_double, c_dp, c_dp]
        L.orc_spline_transform_by_right.argtypes = [c_dp, c_dp, C.c_int, c_dp, c_dp]
        L.orc_spline_transform_to.argtypes = [C.c_int, C.c_double, C.c_double, c_dp, c_dp, C.c_int, C.c_double, c_dp, c_dp]
        L.orc_detect_semidense.argtypes = [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_float, c_fp, c_fp, C.c_int]
        L.orc_keypoint_depths.argtypes = [c_fp, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, c_dp, c_dp]
        L.orc_is_keyframe.argtypes = [c_dp, c_dp, c_dp, C.c_int, C.c_int, C.c_double, C.c_double, c_dp, c_dp,
                                      C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_dp, c_dp]
        L.orc_vo_create.restype = C.c_void_p
        L.orc_vo_create.argtypes = [C.POINTER(OrcVoOpts)]
        L.orc_vo_destroy.argtypes = [C.c_void_p]
        L.orc_vo_set_spline.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, c_dp, c_dp]
        L.orc_vo_num_keypoints.argtypes = [C.c_void_p, C.c_int]
        L.orc_vo_last_trace.argtypes = [C.c_void_p, C.POINTER(OrcTraceRec), C.c_int]
        L.orc_vo_get_state.argtypes = [C.c_void_p, C.POINTER(OrcVoState)]
        L.orc_vo_set_state.argtypes = [C.c_void_p, C.POINTER(OrcVoState)]
        L.orc_vo_set_keyframe.argtypes = [C.c_void_p, c_u8p, c_fp]
        L.orc_margins_get.argtypes = [c_dp]
        L.orc_vo_keypoints.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp]
        L.orc_vo_spline.argtypes = [C.c_void_p, c_dp, c_dp, c_ip, c_dp, c_dp]
        L.orc_vo_track_frame.argtypes = [C.c_void_p, c_u8p, c_fp, C.c_double, c_u8p, C.c_double, C.c_double, c_dp,
                                         C.POINTER(OrcVoInfo)]
    return L


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmbavo_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = _bind(path)
    return _LIB


class _Variant:
    """This module with lib() answering for another build of the same sources (everything else is shared)."""

    def __init__(self, path):
        self._path, self._lib = path, None

    def lib(self):
        if self._lib is None:
            self._lib = _bind(self._path)
        return self._lib

    def __getattr__(self, name):
        return globals()[name]


def fma_variant():
    """The restatement compiled WITH floating-point contraction (-ffp-contract=fast -mfma: what nvcc does to the reference's
    .cu files by default, -fmad=true): NOT the pinned oracle -- the long-horizon runs use it to show how far two roundings of the
    same algorithm drift apart (tools/long_horizon.py).  None where the host compiler / CPU has no FMA."""
    path = os.path.join(_HERE, "libmbavo_oracle_fma.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", _HERE, "fma"], check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return _Variant(path) if os.path.exists(path) else None


os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # (see bench.py: metered sandboxes; read by libgomp at its first load)

def ref():
    """The reference's own compilable sources (oracle/_ref), or None."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libmbavo_ref.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_bilinear.restype = C.c_int
        R.ref_pixel_intensity.restype = C.c_int
        R.ref_bilinear.argtypes = [c_u8p, c_fp, C.c_int, C.c_int, C.c_double, C.c_double, c_dp]
        R.ref_pixel_intensity.argtypes = [c_u8p, c_fp, C.c_int, C.c_int, c_dp, c_dp, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_double, c_dp, c_dp]
        R.ref_spline_segment.argtypes = [C.c_double, C.c_double, C.c_double, c_ip, c_dp]
        for n in ("ref_c2_vec3", "ref_c4_vec3", "ref_c2_rot3", "ref_c4_rot3"):
            getattr(R, n).argtypes = [c_dp, C.c_double, c_dp, c_dp]
        R.ref_pyramid_u8.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, C.POINTER(c_u8p)]
        R.ref_image_gradients_u8.argtypes = [c_u8p, C.c_int, C.c_int, c_fp, c_fp]
        if hasattr(R, "ref_compute_pixel_jacobian_residual"):
            R.ref_compute_virtual_camera_poses.argtypes = [C.c_int, C.c_int, c_dp, c_dp, C.c_int, C.c_double, C.c_double,
                                                           c_dp, c_dp, c_dp, c_dp, c_dp]
            if hasattr(R, "ref_compute_local_patches_xy"):
                R.ref_compute_local_patches_xy.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_int, c_dp, c_dp]
            R.ref_compute_pixel_jacobian_residual.argtypes = [
                c_u8p, c_fp, C.POINTER(c_u8p), C.c_int, C.c_int, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp, C.c_int,
                c_ip, C.c_int, c_dp, C.c_int, C.c_int, c_dp, c_dp]
        if hasattr(R, "ref_evaluate_omp"):
            R.ref_evaluate_omp.restype = C.c_int
            R.ref_evaluate_omp.argtypes = [c_u8p, c_fp, C.POINTER(c_u8p), C.c_int, C.c_int, c_dp, c_dp, C.c_int, C.c_double, C.c_double,
                                           c_dp, c_dp, c_dp, c_dp, C.c_int, c_ip, C.c_int, c_dp, C.c_int, C.c_int, C.c_double,
                                           C.c_void_p, C.c_int, C.c_int, c_dp]
        R.ref_lm_new.restype = C.c_void_p
        R.ref_tr_new.restype = C.c_void_p
        R.ref_tr_new.argtypes = [C.c_int]
        for n in ("ref_lm_delete", "ref_lm_reset", "ref_lm_rejected", "ref_tr_delete"):
            getattr(R, n).argtypes = [C.c_void_p]
        R.ref_lm_accepted.argtypes = [C.c_void_p, C.c_double]
        R.ref_lm_radius.argtypes = [C.c_void_p]
        R.ref_lm_radius.restype = C.c_double
        R.ref_tr_reset.argtypes = [C.c_void_p, C.c_double]
        R.ref_tr_quality.argtypes = [C.c_void_p, C.c_double, C.c_double]
        R.ref_tr_quality.restype = C.c_double
        R.ref_tr_accepted.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _REF = R
    return _REF


# --------------------------------------------------------------------------
# numpy-level helpers shared by tests / smoke / bench
# --------------------------------------------------------------------------
def packed_len(k):
    nd = 6 * k + 1
    return nd * (nd + 1) // 2


def make_problem(S, F, K, P, k, N, H, W, ref_img, ref_dIxy, cur_imgs, kp_xy, kp_z, pattern,
                 intr, cap, exp_t, t0, dt, knots_t, knots_R, start_idx, huber_a,
                 outlier=None, num_bad=0):
    """Returns (OrcProblem, keepalive list)."""
    keep = [ref_img, ref_dIxy, cur_imgs, kp_xy, kp_z, pattern, cap, exp_t, knots_t, knots_R, start_idx, outlier]
    cur_arr = (c_u8p * F)(*[u8p(c) for c in cur_imgs])
    keep.append(cur_arr)
    p = OrcProblem()
    p.S, p.F, p.K, p.P, p.k, p.N, p.H, p.W = S, F, K, P, k, N, H, W
    p.ref_img = u8p(ref_img)
    p.ref_dIxy = fp(ref_dIxy)
    p.cur_imgs = cur_arr
    p.kp_xy = dp(kp_xy)
    p.kp_z = dp(kp_z)
    p.pattern = ip(pattern)
    p.outlier = u8p(outlier) if outlier is not None else None
    p.num_bad = num_bad
    for i in range(4):
        p.intr[i] = float(intr[i])
    p.cap = dp(cap)
    p.exp_t = dp(exp_t)
    p.t0, p.dt = float(t0), float(dt)
    p.knots_t = dp(knots_t)
    p.knots_R = dp(knots_R)
    p.start_idx = ip(start_idx)
    p.huber_a = float(huber_a)
    return p, keep


def evaluate(prob, with_hessian=True, patch_blocks=None):
    """orc_evaluate -> dict(cost, H, g, frame_blocks, patch_blocks)."""
    L = lib()
    E = packed_len(prob.k)
    n = 6 * prob.N
    if patch_blocks is None:
        patch_blocks = np.zeros(prob.F * prob.K * E)
    frame_blocks = np.zeros(prob.F * E)
    cost = np.zeros(1)
    H = np.zeros(n * n) if with_hessian else None
    g = np.zeros(n) if with_hessian else None
    L.orc_evaluate(C.byref(prob), dp(patch_blocks), dp(frame_blocks), dp(cost), dp(H), dp(g))
    return dict(cost=float(cost[0]), H=None if H is None else H.reshape(n, n).T.copy(), g=g,
                frame_blocks=frame_blocks.reshape(prob.F, E),
                patch_blocks=patch_blocks.reshape(prob.F, prob.K, E))


def evaluate_fast(prob, num_threads=1, with_hessian=True):
    L = lib()
    E = packed_len(prob.k)
    n = 6 * prob.N
    frame_blocks = np.zeros(prob.F * E)
    cost = np.zeros(1)
    H = np.zeros(n * n) if with_hessian else None
    g = np.zeros(n) if with_hessian else None
    L.orc_evaluate_fast(C.byref(prob), int(num_threads), dp(frame_blocks), dp(cost), dp(H), dp(g))
    return dict(cost=float(cost[0]), H=None if H is None else H.reshape(n, n).T.copy(), g=g,
                frame_blocks=frame_blocks.reshape(prob.F, E))


def count_valid(prob, num_threads=1):
    """valid pixels per frame (all S warps in bounds), as mbavo_eval_batch's d_valid counts them"""
    v = np.zeros(prob.F)
    lib().orc_count_valid(C.byref(prob), int(num_threads), dp(v))
    return v


def stages_with_reference(prob_args, with_jacobians=True):
    """One evaluation with the REFERENCE's own per-sample code for the two dominant stages (oracle/_ref:
    ref_compute_virtual_camera_poses, ref_compute_pixel_jacobian_residual) and the oracle's restated reductions for the
    rest (patch / frame blocks, merge).  prob_args: dict with the make_problem() arguments.  Returns dict with
    poses, J_t, J_R, centres, residuals, jacobians, frame_blocks, cost."""
    R, L = ref(), lib()
    assert R is not None and hasattr(R, "ref_compute_pixel_jacobian_residual"), "oracle/_ref with the stage drivers is not built"
    a = prob_args
    S, F, K, P, k, N, H, W = a["S"], a["F"], a["K"], a["P"], a["k"], a["N"], a["H"], a["W"]
    E = packed_len(k)
    poses, Jt, JR = np.zeros(F * S * 7), np.zeros(F * S * 9 * k), np.zeros(F * S * 12 * k)
    R.ref_compute_virtual_camera_poses(S, F, dp(a["cap"]), dp(a["exp_t"]), k, a["t0"], a["dt"], dp(a["knots_t"]),
                                       dp(a["knots_R"]), dp(poses), dp(Jt), dp(JR))
    centres = np.zeros(F * K * 2)
    (R.ref_compute_local_patches_xy if hasattr(R, "ref_compute_local_patches_xy") else L.orc_compute_local_patches_xy)(
        S, F, dp(poses), dp(a["kp_xy"]), dp(a["kp_z"]), K, dp(a["intr"]), dp(centres))
    res = np.zeros(F * K * P)
    jac = np.zeros(F * K * P * 6 * k) if with_jacobians else None
    cur_arr = (c_u8p * F)(*[u8p(c) for c in a["cur_imgs"]])
    R.ref_compute_pixel_jacobian_residual(u8p(a["ref_img"]), fp(a["ref_dIxy"]), cur_arr, S, F, dp(poses), k, dp(Jt), dp(JR),
                                          dp(centres), dp(a["kp_z"]), K, ip(a["pattern"]), P, dp(a["intr"]), H, W, dp(res),
                                          dp(jac))
    inv = 1.0 / (K * F * P) if K * F * P > 0 else 0.0
    pb = np.zeros(F * K * E)
    L.orc_compute_patch_cost_gradient_hessian(F, K, P, k, dp(res), dp(jac), a["huber_a"], inv, dp(pb))
    fb = np.zeros(F * E)
    L.orc_compute_frame_cost_gradient_hessian(F, K, k, dp(pb), 1 if with_jacobians else 0, None, dp(fb))
    return dict(poses=poses, J_t=Jt, J_R=JR, centres=centres, residuals=res, jacobians=jac,
                frame_blocks=fb.reshape(F, E), cost=float(fb.reshape(F, E)[:, 0].sum()))


def evaluate_with_reference(prob_args, chunk=4096, threads=1):
    """Whole evaluation (frame blocks [F, E]) with the reference's per-sample code, keypoints in chunks so that the
    per-pixel Jacobians and per-patch blocks the reference pipeline materialises stay small (dense problems: 800 MB of
    patch blocks otherwise).  The chunks' frame sums are added in keypoint order.  threads > 1: the chunks are
    independent and run on a thread pool (the C calls release the GIL); the sum order stays the keypoint order."""
    a = dict(prob_args)
    K, F, P, k = a["K"], a["F"], a["P"], a["k"]
    E = packed_len(k)
    R = ref()
    if R is not None and hasattr(R, "ref_evaluate_omp") and os.environ.get("MBAVO_REF_OMP", "1") != "0":
        # the loop over keypoint chunks INSIDE the compiled code (OpenMP; round 4): a baseline of the code, not of a Python
        # thread pool around it.  Chunks of 64 keypoints; a thread adds its chunks in ascending order, the threads' frame
        # blocks are added in thread order.
        fb = np.zeros((F, E))
        cur_arr = (c_u8p * F)(*[u8p(c) for c in a["cur_imgs"]])
        patch_fn = C.cast(lib().orc_compute_patch_cost_gradient_hessian, C.c_void_p)
        used = R.ref_evaluate_omp(u8p(a["ref_img"]), fp(a["ref_dIxy"]), cur_arr, a["S"], F, dp(a["cap"]), dp(a["exp_t"]), k, a["t0"], a["dt"],
                                  dp(a["knots_t"]), dp(a["knots_R"]), dp(a["kp_xy"]), dp(a["kp_z"]), K, ip(a["pattern"]), P, dp(a["intr"]),
                                  a["H"], a["W"], a["huber_a"], patch_fn, int(threads), int(chunk) if chunk <= 256 else 64, dp(fb))
        assert used >= 1
        return fb
    total = np.zeros((F, E))
    spans = [(k0, min(K, k0 + chunk)) for k0 in range(0, max(K, 1), chunk) if min(K, k0 + chunk) > k0]

    def one(span):
        k0, k1 = span
        sub = dict(a, K=k1 - k0, kp_xy=np.ascontiguousarray(a["kp_xy"][k0:k1]), kp_z=np.ascontiguousarray(a["kp_z"][k0:k1]))
        return stages_with_reference(sub)["frame_blocks"] * ((k1 - k0) * F * P)  # un-normalise (inv = 1 / (K_chunk F P))

    if threads > 1 and len(spans) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as ex:
            parts = list(ex.map(one, spans))
    else:
        parts = [one(sp) for sp in spans]
    for part in parts:
        total += part
    return total / (K * F * P) if K * F * P > 0 else total
