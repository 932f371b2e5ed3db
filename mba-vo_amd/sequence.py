"""Synthetic blurred sequence + trackFrame driver, product code only (no oracle): bench.py's `trackframe` config and
tools use it; the parity tests have their own oracle-rendered twin in tests/frontend.py.

A textured fronto-parallel plane seen by a camera moving along a ground-truth spline
(ba_tracker/generate_synthetic_data.cpp:127-214): per time step a sharp image (keyframe candidate), its z-depth map
and a motion-blurred image (the tracked frame), rendered on the GPU by mbavo_synthesize_blur.
"""
import ctypes as C
import time

import numpy as np

from . import capi, synth

PATTERN_SMALL = np.array([0, 0, 1, 0, 0, 1, -1, 0, 0, -1, 1, 1, -1, -1, 1, -1], np.int32)
# the reference-shaped configuration: 640 x 480, 4 levels, 30-px grid keypoints x 8-pixel pattern, k = 2, S = 8
REFERENCE_CFG = dict(levels=4, S=(8, 8, 8, 8), k=2, huber_k=10.0, max_nonmono=5, max_iter=30, solver=0, min_quality=0.5,
                     min_dec=1e-3, chi=3.0, flow0=10.0, flow1=24.0, flow2=0.5, kernel=3.0, thr=3.0, cell=30)


def _quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def plane_depth_map(H, W, intr, q, t, D):
    """z-depth, in the camera at pose (q, t) [camera -> plane frame], of the plane z = D of the plane frame."""
    fx, fy, cx, cy = intr
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    ray = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], -1)
    rz = ray @ _quat_R(q)[2]
    return np.ascontiguousarray(((D - t[2]) / rz).astype(np.float32))


def make_sequence(ctx, H=480, W=640, M=8, k_gt=4, trans_scale=0.15, rot_scale=0.02, D=7.5, exp=0.04, frame_dt=0.1,
                  t_first=0.1, blur_samples=8, seed=3, device="cuda:0", trajectory="harness"):
    """trajectory: "harness" = the 7-knot spline of the reference's module harness at the given scales (leaves the texture after
    ~15 frames at 640 x 480); "loop" = synth.loop_spline, bounded, any M (the long-horizon parity runs)."""
    import torch
    L = ctx.lib
    I0 = synth.texture_image(H, W, seed=seed, octaves=(32, 16, 8, 4))
    intr = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0])
    N = 7 if trajectory == "harness" else int((t_first + frame_dt * M) / 0.5) + 6  # (the bounded families: any M, synth.trajectory)
    kt, kR = synth.trajectory(trajectory, N, trans_scale, rot_scale)
    kt, kR = np.ascontiguousarray(kt.ravel()), np.ascontiguousarray(kR.ravel())
    t0, dtk = 0.0, 0.5
    times = t_first + frame_dt * np.arange(M + 1)
    d_ref = torch.from_numpy(I0).to(device)
    d_out = torch.zeros(H * W, dtype=torch.uint8, device=device)
    sharp, depth, blur, gt = [], [], [], []
    for t in times:
        p, q = np.zeros(3), np.zeros(4)
        capi.check(L.mbavo_spline_get_pose(k_gt, t0, dtk, capi.dp(kt), capi.dp(kR), N, float(t), capi.dp(p), capi.dp(q), None, None),
                   "mbavo_spline_get_pose")
        for n_s, e, dst in ((2, 0.0, sharp), (blur_samples, exp, blur)):  # exposure 0: the sharp warp
            capi.check(L.mbavo_synthesize_blur(d_ref.data_ptr(), H, W, float(D), capi.dp(intr), k_gt, t0, dtk, capi.dp(kt), capi.dp(kR),
                                               N, float(t), float(e), n_s, d_out.data_ptr(), None), "mbavo_synthesize_blur")
            dst.append(np.ascontiguousarray(d_out.cpu().numpy().reshape(H, W)))
        depth.append(plane_depth_map(H, W, intr, q, p, D))
        gt.append(np.r_[p, q])
    return dict(H=H, W=W, intr=intr, times=times, exp=exp, frame_dt=frame_dt, sharp=sharp, depth=depth, blur=blur,
                gt=np.array(gt), D=D)


def vo_options(seq, cfg=REFERENCE_CFG):
    o = capi.VoOptions()
    pats = [synth.PATTERN8 if l == 0 else PATTERN_SMALL for l in range(cfg["levels"])]
    o.H, o.W, o.num_pyramid_levels = seq["H"], seq["W"], cfg["levels"]
    for i in range(4):
        o.intrinsics[i] = float(seq["intr"][i])
    for l in range(cfg["levels"]):
        o.num_virtual_poses_per_frame[l], o.patch_size[l] = cfg["S"][l], pats[l].size // 2
        o.local_patch_pattern_xy[l] = capi.ip(pats[l])
    o.huber_k, o.max_consecutive_nonmonotonic_steps = cfg["huber_k"], cfg["max_nonmono"]
    o.max_num_iterations, o.solver_type = cfg["max_iter"], cfg["solver"]
    o.spline_deg_k, o.min_step_quality, o.min_abs_cost_decrease = cfg["k"], cfg["min_quality"], cfg["min_dec"]
    o.dt_frame, o.dt_ctrl_knot, o.max_chi_square_error = seq["frame_dt"], seq["frame_dt"], cfg["chi"]
    o.keyframe_max_flow_mag0, o.keyframe_max_flow_mag1 = cfg["flow0"], cfg["flow1"]
    o.keyframe_max_flow_mag2, o.keyframe_max_blur_kernel_mag = cfg["flow2"], cfg["kernel"]
    o.score_threshold, o.grid_selection_cell_H, o.grid_selection_cell_W = cfg["thr"], cfg["cell"], cfg["cell"]
    return o, pats


def track_sequence(ctx, seq, cfg=REFERENCE_CFG):
    """One pass of BlurAwareDirectTracker::trackFrame over the sequence: list of per-frame dicts (pose, keyframe decision,
    LM records, wall seconds of the mbavo_vo_track_frame call)."""
    o, keep = vo_options(seq, cfg)
    vo = capi.vp()
    capi.check(ctx.lib.mbavo_vo_create(ctx.handle, C.byref(o), C.byref(vo)), "mbavo_vo_create")
    out = []
    try:
        for i, t in enumerate(seq["times"]):
            T, info = np.zeros(7), capi.VoInfo()
            sharp, depth, blur = seq["sharp"][i], seq["depth"][i], seq["blur"][i]
            t_call = time.perf_counter()
            rc = ctx.lib.mbavo_vo_track_frame(vo, sharp.ctypes.data, depth.ctypes.data, float(t), blur.ctypes.data, float(t),
                                              float(seq["exp"]), capi.dp(T), C.byref(info))
            dt = time.perf_counter() - t_call
            capi.check(rc, "mbavo_vo_track_frame")
            out.append(dict(T=T, is_keyframe=info.is_keyframe, K0=info.num_keypoints0, num_trace=info.num_trace, cost=info.final_cost,
                            start_idx=info.start_idx, seconds=dt))
    finally:
        ctx.lib.mbavo_vo_destroy(vo)
    return out


def gt_relative(ctx, seq):
    """Ground-truth pose of every frame relative to the first sharp frame (the tracker's world), through the product's
    own Transformation helpers (core/states/Transformation.cpp)."""
    L = ctx.lib
    T0i = np.zeros(7)
    capi.check(L.mbavo_transform_inverse(capi.dp(np.ascontiguousarray(seq["gt"][0])), capi.dp(T0i)), "mbavo_transform_inverse")
    out = []
    for g in seq["gt"]:
        T = np.zeros(7)
        capi.check(L.mbavo_transform_mul(capi.dp(T0i), capi.dp(np.ascontiguousarray(g)), capi.dp(T)), "mbavo_transform_mul")
        out.append(T)
    return np.array(out)


def ate(run, gt_rel):
    """Absolute trajectory error: RMSE over the frames of |t_est - t_gt| (same world frame, no alignment)."""
    return float(np.sqrt(np.mean([np.sum((f["T"][:3] - g[:3]) ** 2) for f, g in zip(run, gt_rel)])))
