"""Long-horizon parity statistics for trackFrame runs: two runs of tests/frontend.py's drivers over
the same frames -- the HIP tracker's and the oracle's -- compared frame by frame.  numpy only; used by
tests/test_gpu_horizon.py, tools/long_horizon.py and bench.py's checker leg.

Discrete results (must be identical): knot start index, keyframe decision, keypoint counts of every level, and the DISCRETE
part of every LM record (level, iteration, kind = initial / accepted / rejected / invalid, outlier count).  Continuous results
(stated tolerances): poses, costs along the trace, ATE against the ground truth."""
import numpy as np

POSE_THRESHOLDS = (1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4)


def _discrete(rows):
    return [r[:4] for r in rows]


def ate(run, gt_rel, lo=0, hi=None):
    hi = len(run) if hi is None else hi
    return float(np.sqrt(np.mean([np.sum((run[i]["T"][:3] - gt_rel[i][:3]) ** 2) for i in range(lo, hi)])))


def compare(got, want, gt_rel, min_step_quality=0.5, flow_thresholds=(), window=50):
    """got / want: lists of per-frame dicts (frontend.run_gpu_vo / run_oracle_vo).  Returns a dict of divergence statistics;
    frame indices are positions in the sequence (frame 0 is the first keyframe), None = never."""
    n = len(want)
    assert len(got) == n
    first = dict(start_idx=None, keyframe=None, keypoints=None, trace_length=None, trace_discrete=None)
    for i, (a, b) in enumerate(zip(got, want)):
        if first["start_idx"] is None and a["start_idx"] != b["start_idx"]: first["start_idx"] = i
        if first["keyframe"] is None and a["is_keyframe"] != b["is_keyframe"]: first["keyframe"] = i
        if first["keypoints"] is None and list(a["K"]) != list(b["K"]): first["keypoints"] = i
        if first["trace_length"] is None and a["num_trace"] != b["num_trace"]: first["trace_length"] = i
        if first["trace_discrete"] is None and _discrete(a["trace"]) != _discrete(b["trace"]): first["trace_discrete"] = i
    hits = [v for v in first.values() if v is not None]
    first_discrete = min(hits) if hits else None
    dpose = np.array([np.abs(a["T"] - b["T"]).max() for a, b in zip(got, want)])
    first_pose = {}
    for thr in POSE_THRESHOLDS:
        w = np.nonzero(dpose > thr)[0]
        first_pose["%g" % thr] = int(w[0]) if w.size else None
    # costs along the trace, over the frames whose discrete records agree
    cost_rel = 0.0
    for a, b in zip(got, want):
        if _discrete(a["trace"]) != _discrete(b["trace"]): continue
        for ra, rb in zip(a["trace"], b["trace"]):
            for j in (5, 6):  # eval_cost, candidate_cost
                if rb[j] != 0 and np.isfinite(rb[j]): cost_rel = max(cost_rel, abs(ra[j] - rb[j]) / abs(rb[j]))
    ate_g, ate_o = ate(got, gt_rel), ate(want, gt_rel)
    win = [abs(ate(got, gt_rel, s, s + window) - ate(want, gt_rel, s, s + window)) for s in range(0, max(1, n - window + 1))] if n >= window else []
    rmse = float(np.sqrt(np.mean([np.sum((a["T"][:3] - b["T"][:3]) ** 2) for a, b in zip(got, want)])))
    # margins on the ORACLE's run: how close any discrete decision came to its threshold (what a flip would need)
    q_margin, q_frame = np.inf, None
    for i, b in enumerate(want):
        for r in b["trace"]:
            if r[2] in (1, 2) and np.isfinite(r[8]):
                m = abs(r[8] - min_step_quality)
                if m < q_margin: q_margin, q_frame = m, i
    f_margin, f_frame = np.inf, None
    for i, b in enumerate(want[1:], 1):
        for thr in flow_thresholds:
            m = abs(b["avg_flow"] - thr)
            if m < f_margin: f_margin, f_frame = m, i
    return dict(frames=n, keyframes_oracle=int(sum(b["is_keyframe"] for b in want)), keyframes_gpu=int(sum(a["is_keyframe"] for a in got)),
                lm_records_oracle=int(sum(len(b["trace"]) for b in want)),
                first_discrete_divergence=first_discrete, first_divergence=first,
                first_pose_divergence=first_pose, max_abs_pose_diff=float(dpose.max()), max_abs_pose_diff_frame=int(dpose.argmax()),
                pose_diff_quantiles={"50%": float(np.quantile(dpose, 0.5)), "90%": float(np.quantile(dpose, 0.9)), "99%": float(np.quantile(dpose, 0.99))},
                trace_cost_max_rel_diff=cost_rel, ate_gt_gpu=ate_g, ate_gt_oracle=ate_o, abs_delta_ate=abs(ate_g - ate_o),
                abs_delta_ate_windows_max=float(max(win)) if win else None, window=window, trajectory_rmse_gpu_vs_oracle=rmse,
                min_quality_margin=None if q_frame is None else float(q_margin), min_quality_margin_frame=q_frame,
                min_flow_margin=None if f_frame is None else float(f_margin), min_flow_margin_frame=f_frame)


def report(title, st):
    fp, fd = st["first_pose_divergence"], st["first_divergence"]
    L = ["== %s" % title,
         "frames %d, keyframes oracle / gpu %d / %d, LM records (oracle) %d" % (st["frames"], st["keyframes_oracle"], st["keyframes_gpu"], st["lm_records_oracle"]),
         "(a) first frame with a discrete difference: %s   [start_idx %s | keyframe decision %s | keypoint counts %s | trace length %s | LM record (level, iter, kind, outliers) %s]"
         % (st["first_discrete_divergence"], fd["start_idx"], fd["keyframe"], fd["keypoints"], fd["trace_length"], fd["trace_discrete"]),
         "(b) first frame with max|pose_gpu - pose_oracle| above: " + ", ".join("%s: %s" % (k, fp[k]) for k in fp),
         "    max |pose diff| %.3e at frame %d; quantiles 50%% %.2e 90%% %.2e 99%% %.2e"
         % (st["max_abs_pose_diff"], st["max_abs_pose_diff_frame"], st["pose_diff_quantiles"]["50%"], st["pose_diff_quantiles"]["90%"], st["pose_diff_quantiles"]["99%"]),
         "    costs along the trace (frames with identical records): max relative difference %.3e" % st["trace_cost_max_rel_diff"],
         "ATE vs ground truth: gpu %.9e, oracle %.9e, |dATE| %.3e over the run; max over sliding %d-frame windows %s; trajectory RMSE gpu vs oracle %.3e"
         % (st["ate_gt_gpu"], st["ate_gt_oracle"], st["abs_delta_ate"], st["window"],
            "n/a" if st["abs_delta_ate_windows_max"] is None else "%.3e" % st["abs_delta_ate_windows_max"], st["trajectory_rmse_gpu_vs_oracle"]),
         "margins on the oracle's run: min |step quality - min_step_quality| %s (frame %s); min |avg_flow - keyframe threshold| %s (frame %s)"
         % ("n/a" if st["min_quality_margin"] is None else "%.3e" % st["min_quality_margin"], st["min_quality_margin_frame"],
            "n/a" if st["min_flow_margin"] is None else "%.3e" % st["min_flow_margin"], st["min_flow_margin_frame"])]
    return "\n".join(L)


# ---------------------------------------------------------------------------------------------------------------------------
# mbavo_lm_batch against the ORACLE's optimizePyramidLevel (blur_aware_direct_tracker.cpp:590-924) on pairs of a rendered batch
LM_OPTS = dict(max_it=25, max_nonmono=5, min_q=0.5, min_dec=1e-3, chi=3.0)


def lm_batch_vs_oracle(orc, mbavo, ctx, batch, pairs, k, N, solver, opts=LM_OPTS, trace_cap=64):
    """Runs mbavo_lm_batch on the WHOLE batch (workloads.RenderedPairBatch) and the oracle's loop on `pairs` of it.  Returns
    statistics: pairs whose discrete records (iter, kind, outlier count) differ, max final-cost / pose differences, |dATE|,
    the smallest |step quality - threshold| over the oracle's records (what a flipped decision would need)."""
    import ctypes as C
    import time
    import torch
    import tracking
    capi = mbavo.capi
    B = batch.B
    batch.reset_knots()
    arr = (capi.Problem * B)()
    for b in range(B):
        C.memmove(C.byref(arr[b]), C.byref(batch.array[b]), C.sizeof(capi.Problem))
        arr[b].N = N
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, opts["max_it"], opts["max_nonmono"]
    o.solver_type, o.sync_every = solver, 0
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = opts["min_q"], opts["min_dec"], opts["chi"]
    res = (capi.LmBatchResult * B)()
    trace = (capi.TraceRec * (B * trace_cap))()
    rc = ctx.lib.mbavo_lm_batch(ctx.handle, B, arr, C.byref(o), res, trace, trace_cap)
    assert rc == 0, rc
    torch.cuda.synchronize()
    differ, cost_rel, dpose, e_gpu, e_orc, q_margin, t_orc, records, accepted = [], 0.0, 0.0, [], [], np.inf, 0.0, 0, 0
    for b in pairs:
        p = batch.host_problem(b)
        sc = dict(levels=[dict(H=p.H, W=p.W, ref=p.ref, grad=p.grad, cur=p.cur, kp_xy=p.kp_xy, kp_z=p.kp_z, pattern=p.pattern, S=p.S)],
                  k=k, N=N, F=1, cap=p.cap, exp=p.exp, t0=p.t0, dt=p.dt, intr=p.intr,
                  kt0=np.ascontiguousarray(p.knots_t.reshape(-1, 3)[:N]), kR0=np.ascontiguousarray(p.knots_R.reshape(-1, 4)[:N]))
        t = time.perf_counter()
        want = tracking.run_oracle_tracker(orc, sc, dict(max_num_iterations=opts["max_it"], max_nonmono=opts["max_nonmono"], solver_type=solver,
                                                         huber_k=p.huber, min_step_quality=opts["min_q"],
                                                         min_abs_cost_decrease=opts["min_dec"], max_chi_square_error=opts["chi"]))
        t_orc += time.perf_counter() - t
        got = [(r.iter, r.kind, r.num_outliers) for r in trace[b * trace_cap:b * trace_cap + res[b].num_trace]]
        records += len(want["trace"])
        accepted += res[b].accepted
        for r in want["trace"]:
            if r[2] in (1, 2) and np.isfinite(r[8]): q_margin = min(q_margin, abs(r[8] - opts["min_q"]))
        h = batch._host[b]
        kt = h["dkt"].cpu().numpy().reshape(-1, 3)[:N]
        kR = h["dkR"].cpu().numpy().reshape(-1, 4)[:N]
        tc = float(sc["cap"][0])
        pg, qg = tracking.pose_at(orc, k, sc["t0"], sc["dt"], kt, kR, tc)
        po, qo = tracking.pose_at(orc, k, sc["t0"], sc["dt"], want["kt"], want["kR"], tc)
        p_gt, _ = tracking.pose_at(orc, 4, sc["t0"], sc["dt"], h["kt_gt"], h["kR"], tc)
        e_gpu.append(np.sum((pg - p_gt) ** 2))
        e_orc.append(np.sum((po - p_gt) ** 2))
        if got != [(r[1], r[2], r[3]) for r in want["trace"]]:
            differ.append(int(b))
            continue
        cost_rel = max(cost_rel, abs(res[b].final_cost - want["cost"]) / max(1.0, want["cost"]))
        dpose = max(dpose, float(np.abs(pg - po).max()), float(np.abs(qg - qo).max()))
    ate_g, ate_o = float(np.sqrt(np.mean(e_gpu))), float(np.sqrt(np.mean(e_orc)))
    return dict(batch=B, pairs_compared=len(list(pairs)), k=k, N=N, solver=solver, lm_records_oracle=records, accepted_steps_gpu=int(accepted),
                pairs_with_different_records=differ, final_cost_max_rel_diff=cost_rel, pose_max_abs_diff=dpose,
                ate_gt_gpu=ate_g, ate_gt_oracle=ate_o, abs_delta_ate=abs(ate_g - ate_o),
                min_quality_margin=None if not np.isfinite(q_margin) else float(q_margin), oracle_seconds=round(t_orc, 2))
