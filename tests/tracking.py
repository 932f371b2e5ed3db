"""Synthetic tracking scenes shared by the CPU (oracle) and GPU tracker tests: a keyframe, blurred
current frames synthesised by warping the keyframe along a ground-truth spline
(generate_synthetic_data.cpp:127-214 restated in the oracle), a perturbed initial spline."""
import ctypes as C

import numpy as np

from mba_vo_amd import synth


def make_tracking_scene(orc, H=120, W=160, levels=3, S=8, k=4, F=1, seed=0, mode="semidense", z=7.5,
                        blur_samples=16, perturb=4e-3, exp=0.1, trans_scale=0.02, rot_scale=0.3, image="noise"):
    """Returns dict with per-level numpy data, GT knots, initial knots, times."""
    rng = np.random.default_rng(seed)
    L = orc.lib()
    if image == "noise":
        ref0 = synth.texture_image(H, W, seed=seed + 1, octaves=(32, 16, 8, 4))
    else:
        ref0 = synth.shapes_image(H, W)
    t0, dt = 0.0, 0.5
    cap = np.ascontiguousarray(0.25 + dt * np.arange(F))
    expv = np.full(F, exp)
    N = int((cap[-1] + exp - t0) / dt) + k
    kt_gt, kR_gt = synth.harness_spline(trans_scale, rot_scale, N)
    intr0 = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0])
    curs0 = []
    for f in range(F):
        out = np.zeros((H, W), np.uint8)
        L.orc_synthesize_blur(orc.u8p(ref0), H, W, float(z), orc.dp(intr0), k, t0, dt, orc.dp(kt_gt.ravel().copy()),
                              orc.dp(kR_gt.ravel().copy()), float(cap[f]), float(expv[f]), blur_samples, orc.u8p(out))
        curs0.append(out)
    refs = synth.pyramid(ref0, levels)
    curs = [synth.pyramid(c, levels) for c in curs0]
    lv = []
    for l in range(levels):
        Hl, Wl = refs[l].shape
        if mode == "dense":
            xy, zz = synth.dense_keypoints(Hl, Wl, margin=max(2, 12 // 2 ** l), const_z=z)
            pat = np.zeros(2, np.int32)
        else:
            xy, zz = synth.semi_dense_keypoints(refs[l], cell=max(6, 16 // 2 ** l), thresh=1.0, margin=max(4, 14 // 2 ** l), const_z=z)
            pat = synth.PATTERN8 if l == 0 else np.array([0, 0, 1, 0, 0, 1, -1, 0, 0, -1, 1, 1, -1, -1, 1, -1], np.int32)
        lv.append(dict(H=Hl, W=Wl, ref=refs[l], grad=synth.image_gradients(refs[l]), cur=[c[l] for c in curs],
                       kp_xy=xy, kp_z=zz, pattern=np.ascontiguousarray(pat, np.int32), S=S))
    kt0 = kt_gt + rng.normal(0, perturb, kt_gt.shape)
    dR = rng.normal(0, perturb, (N, 3))
    kR0 = np.zeros((N, 4))
    L.orc_plus_R(orc.dp(kR_gt.ravel().copy()), orc.dp(dR.ravel().copy()), N, orc.dp(kR0.ravel()))
    tmp = np.zeros(4 * N)
    L.orc_plus_R(orc.dp(np.ascontiguousarray(kR_gt.ravel())), orc.dp(np.ascontiguousarray(dR.ravel())), N, orc.dp(tmp))
    kR0 = tmp.reshape(N, 4)
    return dict(levels=lv, k=k, N=N, F=F, cap=cap, exp=expv, t0=t0, dt=dt, intr=intr0, kt_gt=kt_gt, kR_gt=kR_gt,
                kt0=np.ascontiguousarray(kt0), kR0=np.ascontiguousarray(kR0), z=z)


OPTS = dict(max_num_iterations=50, max_nonmono=5, solver_type=0, huber_k=10.0, min_step_quality=0.5,
            min_abs_cost_decrease=1e-3, max_chi_square_error=3.0)


def run_oracle_tracker(orc, sc, opts=OPTS, trace_cap=1024):
    L = orc.lib()
    nl, F = len(sc["levels"]), sc["F"]
    levels = (orc.OrcLevel * nl)()
    keep = []
    for i, lv in enumerate(sc["levels"]):
        cur_arr = (orc.c_u8p * F)(*[orc.u8p(c) for c in lv["cur"]])
        keep.append(cur_arr)
        q = levels[i]
        q.H, q.W, q.K, q.P, q.S = lv["H"], lv["W"], lv["kp_xy"].shape[0], lv["pattern"].size // 2, lv["S"]
        q.ref_img, q.ref_dIxy, q.cur_imgs = orc.u8p(lv["ref"]), orc.fp(lv["grad"]), cur_arr
        q.kp_xy, q.kp_z, q.pattern = orc.dp(lv["kp_xy"]), orc.dp(lv["kp_z"]), orc.ip(lv["pattern"])
    o = orc.OrcTrackOpts()
    o.num_levels, o.k = nl, sc["k"]
    o.max_num_iterations, o.max_nonmono, o.solver_type = opts["max_num_iterations"], opts["max_nonmono"], opts["solver_type"]
    for i in range(4):
        o.intr[i] = float(sc["intr"][i])
    o.huber_k, o.min_step_quality = opts["huber_k"], opts["min_step_quality"]
    o.min_abs_cost_decrease, o.max_chi_square_error = opts["min_abs_cost_decrease"], opts["max_chi_square_error"]
    kt, kR = sc["kt0"].ravel().copy(), sc["kR0"].ravel().copy()
    start = np.zeros(F, np.int32)
    cost = np.zeros(1)
    trace = (orc.OrcTraceRec * trace_cap)()
    n = L.orc_optimize_trajectory(C.byref(o), levels, F, orc.dp(sc["cap"]), orc.dp(sc["exp"]), sc["t0"], sc["dt"],
                                  orc.dp(kt), orc.dp(kR), sc["N"], orc.ip(start), orc.dp(cost), trace, trace_cap)
    assert 0 <= n <= trace_cap
    return dict(kt=kt.reshape(-1, 3), kR=kR.reshape(-1, 4), start=start, cost=float(cost[0]),
                trace=[(r.level, r.iter, r.kind, r.num_outliers, r.radius, r.eval_cost, r.candidate_cost, r.model_change, r.quality)
                       for r in trace[:n]])


def run_gpu_tracker(mbavo, ctx, sc, opts=OPTS, trace_cap=1024):
    import torch
    cap_mod = mbavo.capi
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    nl, F = len(sc["levels"]), sc["F"]
    levels = (cap_mod.Level * nl)()
    keep = []
    for i, lv in enumerate(sc["levels"]):
        ref, grad = t(lv["ref"]), t(lv["grad"])
        curs = [t(c) for c in lv["cur"]]
        ptrs = torch.tensor([c.data_ptr() for c in curs], dtype=torch.int64, device=dev)
        xy, z, pat = t(lv["kp_xy"]), t(lv["kp_z"]), t(lv["pattern"])
        keep += [ref, grad, curs, ptrs, xy, z, pat]
        q = levels[i]
        q.H, q.W, q.K, q.P, q.S = lv["H"], lv["W"], lv["kp_xy"].shape[0], lv["pattern"].size // 2, lv["S"]
        q.d_ref_img, q.d_ref_dIxy, q.d_cur_imgs = ref.data_ptr(), grad.data_ptr(), ptrs.data_ptr()
        q.d_kp_xy, q.d_kp_z, q.d_pattern = xy.data_ptr(), z.data_ptr(), pat.data_ptr()
    torch.cuda.synchronize()
    o = cap_mod.TrackOpts()
    o.num_levels, o.spline_deg_k = nl, sc["k"]
    o.max_num_iterations, o.max_consecutive_nonmonotonic_steps, o.solver_type = opts["max_num_iterations"], opts["max_nonmono"], opts["solver_type"]
    for i in range(4):
        o.intrinsics[i] = float(sc["intr"][i])
    o.huber_k, o.min_step_quality = opts["huber_k"], opts["min_step_quality"]
    o.min_abs_cost_decrease, o.max_chi_square_error = opts["min_abs_cost_decrease"], opts["max_chi_square_error"]
    for name in ("fast_solve_ratio", "speculate", "persist_levels", "ride_along", "resum"):  # ABI 3 tail of mbavo_track_opts (zero = default)
        if name in opts:
            setattr(o, name, opts[name])
    kt, kR = sc["kt0"].ravel().copy(), sc["kR0"].ravel().copy()
    start = np.zeros(F, np.int32)
    cost = np.zeros(1)
    trace = (cap_mod.TraceRec * trace_cap)()
    n = ctx.lib.mbavo_optimize_trajectory(ctx.handle, C.byref(o), levels, F, cap_mod.dp(sc["cap"]), cap_mod.dp(sc["exp"]),
                                          sc["t0"], sc["dt"], cap_mod.dp(kt), cap_mod.dp(kR), sc["N"], cap_mod.ip(start),
                                          cap_mod.dp(cost), trace, trace_cap)
    assert 0 <= n <= trace_cap, n
    return dict(kt=kt.reshape(-1, 3), kR=kR.reshape(-1, 4), start=start, cost=float(cost[0]),
                trace=[(r.level, r.iter, r.kind, r.num_outliers, r.radius, r.eval_cost, r.candidate_cost, r.model_change, r.quality)
                       for r in trace[:n]])


def pose_at(orc, k, t0, dt, kt, kR, t):
    L = orc.lib()
    idx, u = C.c_int(), C.c_double()
    L.orc_spline_segment(float(t), t0, dt, C.byref(idx), C.cast(C.byref(u), orc.c_dp))
    p, q = np.zeros(3), np.zeros(4)
    nm = "c2" if k == 2 else "c4"
    getattr(L, "orc_%s_vec3" % nm)(orc.dp(np.ascontiguousarray(kt[idx.value:].ravel())), u.value, orc.dp(p), None)
    getattr(L, "orc_%s_rot3" % nm)(orc.dp(np.ascontiguousarray(kR[idx.value:].ravel())), u.value, orc.dp(q), None)
    return p, q


def ate(orc, sc, kt, kR):
    """RMSE over frames of || t_est(t_cap) - t_gt(t_cap) || (same world frame, no alignment)."""
    e = []
    for c in sc["cap"]:
        p, _ = pose_at(orc, sc["k"], sc["t0"], sc["dt"], kt, kR, c)
        pg, _ = pose_at(orc, sc["k"], sc["t0"], sc["dt"], sc["kt_gt"], sc["kR_gt"], c)
        e.append(np.sum((p - pg) ** 2))
    return float(np.sqrt(np.mean(e)))


def flow_error(orc, sc, kt, kR):
    """Mean distance (level-0 pixels) between where the keypoints land in the current frames under the
    estimated and the ground-truth pose at capture time: gauge-free accuracy measure (on a fronto-parallel
    plane small rotations and translations trade off, so translation-only ATE is a weak one)."""
    lv = sc["levels"][0]
    K = lv["kp_xy"].shape[0]
    errs = []
    for c in sc["cap"]:
        out = []
        for a, b in ((kt, kR), (sc["kt_gt"], sc["kR_gt"])):
            p, q = pose_at(orc, sc["k"], sc["t0"], sc["dt"], a, b, c)
            pose = np.ascontiguousarray(np.r_[p, q])
            cen = np.zeros(K * 2)
            orc.lib().orc_compute_local_patches_xy(1, 1, orc.dp(pose), orc.dp(lv["kp_xy"]), orc.dp(lv["kp_z"]), K,
                                                   orc.dp(sc["intr"]), orc.dp(cen))
            out.append(cen.reshape(K, 2))
        errs.append(np.linalg.norm(out[0] - out[1], axis=1).mean())
    return float(np.mean(errs))
