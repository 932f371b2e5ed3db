"""bench_checks.py -- the CHECKER legs of bench.py, kept in a file of their own so that the timed product path (bench.py,
bench_core.py, bench_side.py) and everything that touches the CPU oracle are told apart by the file name:

  * cpu_baseline: the reference's per-sample code (oracle/_ref, kind "reference") or the oracle's fused port (kind "port") timed on
    the host cores on the same workload -- a reported baseline, run AFTER the timed region, rank 0 at N = 1 only;
  * trackframe_checker / trackframe_long_horizon: the oracle's trackFrame on the sequences the product tracked (parity figures of the
    line: identical discrete decisions, |dATE|, the horizon over which the free-running runs agree to 1e-5).

bench.py imports this module behind its timing, in one place."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

def cpu_baseline(probs, budget_s):
    """The reference's per-sample code (oracle/_ref; kind "reference") or the oracle's fused port (kind "port") timed on
    the host cores on the SAME workload: one sample on 1 thread and one on all host threads, each bounded by `budget_s`
    (a whole number of full evaluations; at least one).  Returns (dict, frame_blocks of the last evaluation)."""
    from oracle import binding as B
    B.build()
    T = max(1, int(os.environ.get("MBAVO_CPU_THREADS", str(os.cpu_count() or 1))))
    # thread counts of the all-threads sample: every logical CPU and, because the sandboxed hosts hand a process a CPU-time
    # quota far below their logical CPU count (256 OpenMP threads ran SLOWER than one there), the cgroup's quota and 16
    cands = {T}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cands.add(max(1, min(T, -(-int(q) // int(per)))))
    except Exception:
        pass
    if T > 16:
        cands.add(16)
    cands = sorted(c for c in cands if c > 1)
    ps = sum(p.pixel_samples for p in probs)
    R = B.ref()
    use_ref = R is not None and hasattr(R, "ref_compute_pixel_jacobian_residual") and \
        os.environ.get("MBAVO_CPU_BASELINE", "reference") == "reference"
    if use_ref:
        args = [dict(S=p.S, F=p.F, K=p.K, P=p.P, k=p.k, N=p.N, H=p.H, W=p.W, ref_img=p.ref, ref_dIxy=p.grad, cur_imgs=p.cur,
                     kp_xy=p.kp_xy, kp_z=p.kp_z, pattern=p.pattern, intr=p.intr, cap=p.cap, exp_t=p.exp, t0=p.t0, dt=p.dt,
                     knots_t=p.knots_t, knots_R=p.knots_R, huber_a=p.huber) for p in probs]
        run = lambda threads: [B.evaluate_with_reference(a, threads=threads) for a in args]
        kind = "reference"
        how = ("per-sample code = the reference's compute_pixel_intensity<double>, C2/C4 spline functors and "
               "Core::MatrixMatrixMultiply compiled from its sources (oracle/_ref, g++ -O2 -ffp-contract=off); kernel launch "
               "geometry, Huber and block reductions = oracle restatement; chunks of 64 keypoints spread over OpenMP threads INSIDE the "
               "compiled code (oracle/ref_shim.cpp: ref_evaluate_omp), per-thread frame blocks added in thread order")
    else:
        plist, keeps = [], []
        for p in probs:
            op, keep = B.make_problem(p.S, p.F, p.K, p.P, p.k, p.N, p.H, p.W, p.ref, p.grad, p.cur, p.kp_xy, p.kp_z,
                                      p.pattern, p.intr, p.cap, p.exp, p.t0, p.dt, p.knots_t, p.knots_R, p.start_idx,
                                      p.huber)
            plist.append(op)
            keeps.append(keep)
        run = lambda threads: [B.evaluate_fast(op, num_threads=threads)["frame_blocks"] for op in plist]
        kind = "port"
        how = "oracle/mbavo_oracle.c orc_evaluate_fast (fused OpenMP port), gcc -O2 -ffp-contract=off"

    def sample(threads, budget=None):
        budget = budget_s if budget is None else budget
        t_all, reps, blocks = 0.0, 0, None
        while reps < 1 or (t_all + t_all / reps < budget and reps < 20):
            t0 = time.perf_counter()
            blocks = run(threads)
            t_all += time.perf_counter() - t0
            reps += 1
        return ps * reps / t_all / 1e6, reps, t_all, blocks

    v1, r1, t1, blocks = sample(1)
    out = dict(value=round(v1, 3), unit="Mpixel-samples/s", cores=1, kind=kind,
               sample="%d full H/g evaluation(s) of the same workload (%d pixel-samples each) on 1 thread, %.1f s; %s"
                      % (r1, ps, t1, how), host_logical_cpus=os.cpu_count())
    if cands:
        best = None
        tried = {}
        for c in cands:  # the budget is shared; the best count is the quoted one
            vc, rc_, tc, blk = sample(c, budget_s / len(cands))
            tried[c] = round(vc, 3)
            if best is None or vc > best[0]:
                best = (vc, rc_, tc, blk, c)
        vT, rT, tT, blocks, T = best
        try:
            usable = len(os.sched_getaffinity(0))
        except Exception:
            usable = None
        out["all_threads"] = dict(value=round(vT, 3), unit="Mpixel-samples/s", cores=T,
                                  sample="%d evaluation(s) on %d threads, %.1f s" % (rT, T, tT),
                                  speedup_over_1_thread=round(vT / v1, 2), cpus_in_affinity_mask=usable, thread_counts_tried=tried,
                                  note="a stated baseline, not a tuned one: an OpenMP loop over keypoint chunks inside the compiled "
                                       "reference code (round 4; a Python thread pool around it before).  The sandboxed host gives "
                                       "this process a fraction of its logical CPUs' real time, so the speed-up over 1 thread is "
                                       "bounded by the sandbox's CPU quota, not by the code")
        if vT > v1:  # the better of the two is the quoted baseline, its thread count stated
            out.update(value=round(vT, 3), cores=T)
            out["single_thread"] = dict(value=round(v1, 3), cores=1)
            out["sample"] = "%d full H/g evaluation(s) of the same workload (%d pixel-samples each) on %d threads, %.1f s " \
                            "(1 thread: %.3f Mpixel-samples/s); %s" % (rT, ps, T, tT, v1, how)
    out["sample_short"] = "%d H/g evaluations of the same workload, %d thread(s), %.1f s" % ((rT, T, tT) if out["cores"] > 1 else (r1, 1, t1))
    out["single_thread_value"] = round(v1, 3)
    return out, np.concatenate(blocks, 0)


def run(ctx, host_probs, fb_gpu, track, cpu_seconds, long_frames):
    """bench.py's one call: (cpu_baseline block with the GPU's result compared against the CPU's, parity block or None)."""
    cb, fb_cpu = cpu_baseline(host_probs, cpu_seconds)
    scale = np.abs(fb_cpu).max(axis=1, keepdims=True)
    cb["gpu_vs_cpu_max_rel_diff"] = float((np.abs(fb_gpu - fb_cpu) / scale).max())
    if track:  # the caller of the path against the checker's trackFrame on the same sequence
        try:
            cb["trackframe_vs_oracle"] = trackframe_checker(ctx, *track, long_frames=long_frames)
        except Exception as e:
            cb["trackframe_vs_oracle"] = {"error": repr(e)}
    return cb, parity_summary(cb.get("trackframe_vs_oracle"))


def _g(x):
    return None if x is None else float("%.4g" % float(x))


def parity_summary(tv):
    """The line's `parity` block (numbers only) from trackframe_checker's report: the criterion of north_star -- identical discrete
    results, ATE within 1e-5 of the reference's -- with the HORIZON over which a free-running pair of runs meets it named, not a
    bare flag."""
    if not isinstance(tv, dict):
        return None
    if "error" in tv:
        return {"error": str(tv["error"])[:120]}
    par = {"trackframe_frames": tv["frames"], "trackframe_abs_delta_ate": _g(tv["abs_delta_ate_vs_oracle"]),
           "trackframe_discrete_results_equal": bool(tv["start_idx_equal"] and tv["keyframe_decisions_equal"] and tv["trace_lengths_equal"])}
    lh = tv.get("long_horizon")
    if isinstance(lh, dict):
        par.update({"long_frames": lh["frames"], "free_running_within_1e-5_frames": lh["free_running_within_1e-5_frames"],
                    "free_running_first_discrete_divergence_frame": lh["free_running"]["first_discrete_divergence_frame"],
                    "free_running_abs_delta_ate": _g(lh["free_running"]["abs_delta_ate"]),
                    "teacher_forced_within_1e-5_frames": lh["teacher_forced_within_1e-5_frames"],
                    "teacher_forced_abs_delta_ate": _g(lh["teacher_forced"]["abs_delta_ate"]),
                    "teacher_forced_discrete_results_equal": lh["teacher_forced_all_discrete_results_identical"]})
    l8 = tv.get("long_horizon_exposure_0.08")
    if isinstance(l8, dict):
        par.update({"exposure_0.08_free_running_within_1e-5_frames": l8["free_running_within_1e-5_frames"],
                    "exposure_0.08_first_discrete_divergence_frame": l8["first_discrete_divergence_frame"],
                    "exposure_0.08_abs_delta_ate": _g(l8["abs_delta_ate"])})
    return par


def trackframe_checker(ctx, seq, got, gt_rel, long_frames=120):
    """The CPU oracle's trackFrame on the SAME rendered sequence the product tracked -- knot start indices and keyframe decisions
    must be identical, |ATE_gt(gpu) - ATE_gt(oracle)| <= 1e-5 (BASELINE.json north_star) -- and the long-horizon leg."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frontend
    from oracle import binding as B
    from mba_vo_amd import sequence
    t0 = time.perf_counter()
    want = frontend.run_oracle_vo(B, seq, sequence.REFERENCE_CFG)
    dt = time.perf_counter() - t0
    ate_o = float(np.sqrt(np.mean([np.sum((w["T"][:3] - g[:3]) ** 2) for w, g in zip(want, gt_rel)])))
    ate_g = sequence.ate(got, gt_rel)
    return {"frames": len(want), "ate_gt_oracle": ate_o, "abs_delta_ate_vs_oracle": abs(ate_g - ate_o),
            "start_idx_equal": bool(all(a["start_idx"] == b["start_idx"] for a, b in zip(got, want))),
            "keyframe_decisions_equal": bool(all(a["is_keyframe"] == b["is_keyframe"] for a, b in zip(got, want))),
            "trace_lengths_equal": bool(all(a["num_trace"] == b["num_trace"] for a, b in zip(got, want))),
            "max_abs_pose_diff": float(max(np.abs(a["T"] - b["T"]).max() for a, b in zip(got, want))),
            "oracle_ms_per_frame_1_thread": round(1e3 * dt / len(want), 3), "within_1e-5": bool(abs(ate_g - ate_o) <= 1e-5),
            "long_horizon": trackframe_long_horizon(ctx, long_frames) if long_frames > 0 else None,
            # the same trajectory with the exposure doubled (0.08 of the 0.1 frame interval): the scene on which north_star's criterion
            # holds FREE-RUNNING over the whole sequence (profiles/r06_long_horizon.txt: 301 of 301 frames)
            "long_horizon_exposure_0.08": trackframe_long_horizon(ctx, long_frames, exp=0.08, teacher_forced=False) if long_frames > 0 else None}


def ate_horizon(got, want, gt, tol=1e-5):
    """Frames (counted from the first) over which BOTH hold for every prefix: |ATE_gt(got) - ATE_gt(want)| <= tol and every pose
    within tol of the oracle's -- the horizon on which north_star's criterion is met by a free-running pair of runs."""
    eg = np.array([np.sum((a["T"][:3] - g[:3]) ** 2) for a, g in zip(got, gt)])
    eo = np.array([np.sum((b["T"][:3] - g[:3]) ** 2) for b, g in zip(want, gt)])
    n = np.arange(1, len(eg) + 1)
    d_ate = np.abs(np.sqrt(np.cumsum(eg) / n) - np.sqrt(np.cumsum(eo) / n))
    d_pose = np.array([np.abs(a["T"] - b["T"]).max() for a, b in zip(got, want)])
    bad = np.nonzero((d_ate > tol) | (d_pose > tol))[0]
    return int(bad[0]) if bad.size else len(eg)


def trackframe_long_horizon(ctx, frames=120, exp=0.04, teacher_forced=True):
    """trackFrame over `frames` rendered 640x480 frames on a bounded trajectory (synth.loop_spline, ~40 % keyframes) against the
    oracle, free-running and TEACHER-FORCED (the HIP tracker put into the oracle's state before every frame); the 300-frame
    statistics over four scenes and what they mean are in profiles/r06_long_horizon.txt and tests/test_gpu_horizon.py."""
    import frontend
    import horizon
    import mba_vo_amd as M
    from oracle import binding as B
    from mba_vo_amd import sequence
    seq = sequence.make_sequence(ctx, H=480, W=640, M=frames, trajectory="loop", exp=exp)
    cfg = dict(sequence.REFERENCE_CFG)
    t0 = time.perf_counter()
    want = frontend.run_oracle_vo(B, seq, cfg)
    dt = time.perf_counter() - t0
    gt = frontend.gt_relative(B, seq)
    got_free = frontend.run_gpu_vo(M, ctx, seq, cfg)
    free = horizon.compare(got_free, want, gt, min_step_quality=cfg["min_quality"])
    if not teacher_forced:  # (the second scene of the line: the free-running criterion only)
        return {"frames": frames + 1, "exposure": exp, "keyframes_oracle": free["keyframes_oracle"], "lm_records_oracle": free["lm_records_oracle"],
                "free_running_within_1e-5_frames": ate_horizon(got_free, want, gt),
                "first_discrete_divergence_frame": free["first_discrete_divergence"], "abs_delta_ate": free["abs_delta_ate"],
                "max_abs_pose_diff": free["max_abs_pose_diff"]}
    got_tf = frontend.run_gpu_vo(M, ctx, seq, cfg, teacher=want)
    tf = horizon.compare(got_tf, want, gt, min_step_quality=cfg["min_quality"])
    pick = lambda st: {"first_discrete_divergence_frame": st["first_discrete_divergence"], "first_pose_divergence_frame": st["first_pose_divergence"],
                       "max_abs_pose_diff": st["max_abs_pose_diff"], "abs_delta_ate": st["abs_delta_ate"],
                       "abs_delta_ate_50_frame_windows_max": st["abs_delta_ate_windows_max"], "ate_gt_gpu": st["ate_gt_gpu"], "ate_gt_oracle": st["ate_gt_oracle"]}
    return {"frames": frames + 1, "exposure": exp, "keyframes_oracle": free["keyframes_oracle"], "lm_records_oracle": free["lm_records_oracle"],
            "oracle_seconds_1_thread": round(dt, 2), "free_running": pick(free), "teacher_forced": pick(tf),
            "free_running_within_1e-5_frames": ate_horizon(got_free, want, gt),
            "teacher_forced_within_1e-5_frames": ate_horizon(got_tf, want, gt),
            "teacher_forced_all_discrete_results_identical": tf["first_discrete_divergence"] is None,
            "teacher_forced_within_1e-5": bool(tf["abs_delta_ate"] <= 1e-5 and (tf["abs_delta_ate_windows_max"] or 0) <= 1e-5)}
