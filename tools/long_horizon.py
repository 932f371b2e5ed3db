"""Long-horizon parity of the HIP tracker against the CPU oracle: 300-frame rendered 640x480
sequences through BlurAwareDirectTracker::trackFrame (blur_aware_direct_tracker.cpp:88-203, 590-699) for k = 2 and k = 4, and
mbavo_lm_batch on ALL 64 pairs of configs[2] and a 64-pair sample of configs[3]'s 512 against the oracle's loop.
Round 6 adds WHOSE AMPLIFICATION IT IS: the same free-running comparison on four scenes that differ in what
conditions the problem -- exposure, trajectory family, plane tilt -- with the per-frame growth factor of |pose_gpu - pose_oracle|, the
first discrete divergence, |dATE| at 25 / 100 / 300 frames and the horizon over which north_star's criterion holds, next to the same
figures for the oracle's own FMA-contracted build against the pinned oracle.
Usage (GPU box): python tools/long_horizon.py [frames] > gpurun_out/long_horizon.txt     (copy to profiles/r06_long_horizon.txt)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


SCENES = (("loop, exposure 0.04 (round 5's scene)", dict(trajectory="loop", exp=0.04)),
          ("loop, exposure 0.08 of the 0.1 frame interval", dict(trajectory="loop", exp=0.08)),
          ("zigzag: the harness family (diagonal legs + its rpy table), bounded", dict(trajectory="zigzag", exp=0.04)),
          ("loop seen under a 12 deg pitch / 5 deg roll tilt (depth 5.9 .. 10.8 across the image)", dict(trajectory="loop_tilted", exp=0.04)),
          ("zigzag, exposure 0.08", dict(trajectory="zigzag", exp=0.08)),
          ("tilted loop, exposure 0.08", dict(trajectory="loop_tilted", exp=0.08)))


def growth_factor(d, lo=2, hi=20):
    """per-frame growth of the pose difference: exp(slope) of a least-squares line through log d over frames lo..hi"""
    idx = np.array([i for i in range(lo, min(hi, len(d) - 1) + 1) if d[i] > 0])
    if idx.size < 3:
        return float("nan")
    return float(np.exp(np.polyfit(idx, np.log(d[idx]), 1)[0]))


def pair_summary(horizon, got, want, gt, cfg):
    """free-running statistics of one pair of runs"""
    st = horizon.compare(got, want, gt, min_step_quality=cfg["min_quality"], flow_thresholds=(cfg["flow0"], cfg["flow1"]))
    d = np.array([np.abs(a["T"] - b["T"]).max() for a, b in zip(got, want)])
    eg = np.array([np.sum((a["T"][:3] - g[:3]) ** 2) for a, g in zip(got, gt)])
    eo = np.array([np.sum((b["T"][:3] - g[:3]) ** 2) for b, g in zip(want, gt)])
    n = np.arange(1, len(eg) + 1)
    d_ate = np.abs(np.sqrt(np.cumsum(eg) / n) - np.sqrt(np.cumsum(eo) / n))
    bad = np.nonzero((d_ate > 1e-5) | (d > 1e-5))[0]
    return dict(growth_per_frame_2_20=growth_factor(d), pose_diff_frame_1=float(d[1]), pose_diff_frame_10=float(d[min(10, len(d) - 1)]),
                pose_diff_frame_20=float(d[min(20, len(d) - 1)]), first_discrete_divergence=st["first_discrete_divergence"],
                first_pose_divergence_1e5=st["first_pose_divergence"]["1e-05"], within_1e5_frames=int(bad[0]) if bad.size else len(d),
                abs_delta_ate={str(m): float(d_ate[min(m, len(d_ate)) - 1]) for m in (25, 100, 300)},
                ate_gt=float(np.sqrt(np.mean(eg))), ate_gt_other=float(np.sqrt(np.mean(eo))),
                keyframes=st["keyframes_oracle"], lm_records=st["lm_records_oracle"])


def scenes_section(M, ctx, orc, frontend, horizon, sequence, frames):
    """The free-running divergence on four scenes: is the ~1.4x per frame the tracker's or the one scene's?"""
    fma = orc.fma_variant()
    cfg = dict(sequence.REFERENCE_CFG)
    rows = {}
    print("== WHOSE AMPLIFICATION: free-running trackFrame (k = 2, reference configuration), %d frames, six scenes" % (frames + 1))
    for name, kw in SCENES:
        seq = sequence.make_sequence(ctx, H=480, W=640, M=frames, **kw)
        gt = frontend.gt_relative(orc, seq)
        want = frontend.run_oracle_vo(orc, seq, cfg)
        got = frontend.run_gpu_vo(M, ctx, seq, cfg)
        tf = frontend.run_gpu_vo(M, ctx, seq, cfg, teacher=want)
        r = dict(gpu=pair_summary(horizon, got, want, gt, cfg))
        if fma is not None:
            r["oracle_fma"] = pair_summary(horizon, frontend.run_oracle_vo(fma, seq, cfg), want, gt, cfg)
        stf = horizon.compare(tf, want, gt, min_step_quality=cfg["min_quality"])
        dtf = np.array([np.abs(a["T"] - b["T"]).max() for a, b in zip(tf, want)])
        r["teacher_forced"] = dict(first_discrete_divergence=stf["first_discrete_divergence"], abs_delta_ate=stf["abs_delta_ate"],
                                   pose_median=float(np.median(dtf)), pose_max=float(dtf.max()))
        rows[name] = r
        print("-- scene: %s   [oracle: %d keyframes, %d LM records, ATE vs ground truth %.4e]" % (name, r["gpu"]["keyframes"], r["gpu"]["lm_records"], r["gpu"]["ate_gt_other"]))
        for who, key in (("HIP tracker vs oracle", "gpu"), ("oracle built with -ffp-contract=fast -mfma vs oracle", "oracle_fma")):
            if key not in r:
                continue
            q = r[key]
            print("   %-52s growth x%.2f per frame (frames 2-20: |pose diff| %.1e -> %.1e -> %.1e); first discrete divergence at frame %s; "
                  "criterion (poses and ATE within 1e-5) holds for %d frames; |dATE| at 25 / 100 / 300 frames: %.1e / %.1e / %.1e"
                  % (who + ":", q["growth_per_frame_2_20"], q["pose_diff_frame_1"], q["pose_diff_frame_10"], q["pose_diff_frame_20"], q["first_discrete_divergence"],
                     q["within_1e5_frames"], q["abs_delta_ate"]["25"], q["abs_delta_ate"]["100"], q["abs_delta_ate"]["300"]))
        t = r["teacher_forced"]
        print("   teacher-forced (every frame from the oracle's state):  first discrete divergence %s; one-step |pose diff| median %.1e max %.1e; |dATE| %.1e"
              % (t["first_discrete_divergence"], t["pose_median"], t["pose_max"], t["abs_delta_ate"]))
        del seq
    g = [r["gpu"]["growth_per_frame_2_20"] for r in rows.values()]
    f = [r["oracle_fma"]["growth_per_frame_2_20"] for r in rows.values() if "oracle_fma" in r]
    h = [r["gpu"]["within_1e5_frames"] for r in rows.values()]
    print("SUMMARY: growth per frame of the HIP-vs-oracle pose difference over the scenes: %s; of the oracle's FMA build vs the oracle: %s; "
          "frames for which north_star's criterion holds free-running: %s (min %d)"
          % (", ".join("x%.2f" % v for v in g), ", ".join("x%.2f" % v for v in f) or "n/a", ", ".join(str(v) for v in h), min(h)))
    return rows


def main():
    import torch
    import mba_vo_amd as M
    import frontend
    import horizon
    from mba_vo_amd import sequence, workloads
    from oracle import binding as orc
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    scenes_only = len(sys.argv) > 2 and sys.argv[2] == "scenes"
    ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    print("# long-horizon parity, library %s" % ctx.lib.mbavo_version().decode())
    t = time.perf_counter()
    seq = sequence.make_sequence(ctx, H=480, W=640, M=frames, trajectory="loop")
    print("# %d frames of 640 x 480 rendered on the GPU in %.2f s (synth.loop_spline; sharp + depth + 8-sample blurred per frame)"
          % (frames + 1, time.perf_counter() - t))
    gt = frontend.gt_relative(orc, seq)
    out = {}
    if scenes_only:
        del seq
        out["scenes"] = scenes_section(M, ctx, orc, frontend, horizon, sequence, frames)
        print("JSON " + json.dumps(out))
        ctx.close()
        return
    for name, cfg, knots in (("trackFrame k = 2 (the reference's default: the tracker's own two knots), Jacobi-SVD solver type", dict(sequence.REFERENCE_CFG), 0),
                             ("trackFrame k = 4, four identity knots through getSplineTrajectory(), solver type 0 (minimum-norm step)", dict(sequence.REFERENCE_CFG, k=4), 4),
                             ("trackFrame k = 2, LDLT solver type", dict(sequence.REFERENCE_CFG, solver=1), 0)):
        t = time.perf_counter()
        want = frontend.run_oracle_vo(orc, seq, cfg, init_knots=knots)
        t_o = time.perf_counter() - t
        sec = []
        got = frontend.run_gpu_vo(M, ctx, seq, cfg, frame_seconds=sec, init_knots=knots)
        got2 = frontend.run_gpu_vo(M, ctx, seq, cfg, init_knots=knots)
        st = horizon.compare(got, want, gt, min_step_quality=cfg["min_quality"], flow_thresholds=(cfg["flow0"], cfg["flow1"]))
        st["gpu_run_reproducible_bit_for_bit"] = bool(all(np.array_equal(a["T"], b["T"]) for a, b in zip(got, got2)))
        st["oracle_ms_per_frame"] = round(1e3 * t_o / frames, 2)
        st["gpu_ms_per_frame"] = round(1e3 * sum(sec[1:]) / frames, 4)
        print(horizon.report(name, st))
        print("    oracle %.2f ms / frame on one thread, GPU %.3f ms / frame; GPU run bit-reproducible: %s"
              % (st["oracle_ms_per_frame"], st["gpu_ms_per_frame"], st["gpu_run_reproducible_bit_for_bit"]))
        # how the pose difference grows: max |pose diff| per block of 25 frames
        d = np.array([np.abs(a["T"] - b["T"]).max() for a, b in zip(got, want)])
        print("    max |pose diff| per 25-frame block: " + " ".join("%.1e" % d[i:i + 25].max() for i in range(0, len(d), 25)))
        # what predicts the first jumps: the oracle's distance from a discontinuity in that frame against the difference it inherits
        print("    frames where the pose difference first passes a threshold -- inherited |pose diff| (frame before), the frame's own, and the ORACLE's margins in "
              "that frame: min |pixel coordinate - integer| over every truncation (A3) [px], min relative distance of a patch cost from the outlier threshold")
        for thr, fr in st["first_pose_divergence"].items():
            if fr is not None and fr > 0:
                print("      > %s at frame %d: inherited %.2e -> %.2e; truncation margin %.2e px (a pose difference of 1e-9 moves a level-0 coordinate by ~3e-7 px), outlier margin %.2e"
                      % (thr, fr, d[fr - 1], d[fr], want[fr]["margins"][0], want[fr]["margins"][1]))
        mt = np.array([w["margins"][0] for w in want[1:]])
        print("    truncation margin of the oracle's run per frame [px]: median %.2e, 10%% quantile %.2e, min %.2e (frame %d); frames with an EXACT integer coordinate (margin 0): %d"
              % (np.median(mt), np.quantile(mt, 0.1), mt[mt > 0].min(), 1 + int(np.argmin(np.where(mt > 0, mt, 1e9))), int((mt == 0).sum())))
        # TEACHER FORCING: every frame from the oracle's state -- 300 one-step comparisons, nothing accumulates
        tf = frontend.run_gpu_vo(M, ctx, seq, cfg, init_knots=knots, teacher=want)
        stf = horizon.compare(tf, want, gt, min_step_quality=cfg["min_quality"], flow_thresholds=(cfg["flow0"], cfg["flow1"]))
        dt_ = np.array([np.abs(a["T"] - b["T"]).max() for a, b in zip(tf, want)])
        same = [i for i in range(len(want)) if horizon._discrete(tf[i]["trace"]) == horizon._discrete(want[i]["trace"]) and tf[i]["is_keyframe"] == want[i]["is_keyframe"]]
        worst = np.argsort(dt_)[::-1][:5]
        print("  TEACHER-FORCED (the HIP tracker put into the oracle's state before every frame: mbavo_vo_set_state / _set_keyframe; %d keyframe re-makes):" % tf[-1]["keyframe_resyncs"])
        print("    frames with identical discrete results (keyframe decision + every LM record): %d of %d; first frame that differs: %s"
              % (len(same), len(want), stf["first_discrete_divergence"]))
        print("    one-step |pose diff|: median %.2e, 90%% %.2e, 99%% %.2e, max %.2e; frames above 1e-6: %d, above 1e-5: %d, above 1e-4: %d; worst frames %s"
              % (np.median(dt_), np.quantile(dt_, 0.9), np.quantile(dt_, 0.99), dt_.max(), int((dt_ > 1e-6).sum()), int((dt_ > 1e-5).sum()), int((dt_ > 1e-4).sum()),
                 ", ".join("%d (%.1e, truncation margin %.1e)" % (i, dt_[i], want[i]["margins"][0]) for i in worst)))
        print("    |dATE| over the run %.3e, max over sliding 50-frame windows %.3e, trajectory RMSE %.3e; costs along identical traces: max rel diff %.2e"
              % (stf["abs_delta_ate"], stf["abs_delta_ate_windows_max"], stf["trajectory_rmse_gpu_vs_oracle"], stf["trace_cost_max_rel_diff"]))
        st["teacher_forced"] = dict(identical_frames=len(same), first_discrete_divergence=stf["first_discrete_divergence"], pose_median=float(np.median(dt_)),
                                    pose_q99=float(np.quantile(dt_, 0.99)), pose_max=float(dt_.max()), above_1e6=int((dt_ > 1e-6).sum()),
                                    above_1e5=int((dt_ > 1e-5).sum()), abs_delta_ate=stf["abs_delta_ate"], rmse=stf["trajectory_rmse_gpu_vs_oracle"])
        if knots == 0 and cfg["solver"] == 0:
            # two roundings of the SAME algorithm: the restatement compiled with floating-point contraction (what nvcc does to the
            # reference's .cu files by default) against the pinned, contraction-free oracle
            fma = orc.fma_variant()
            if fma is not None:
                alt = frontend.run_oracle_vo(fma, seq, cfg, init_knots=knots)
                sf = horizon.compare(alt, want, gt, min_step_quality=cfg["min_quality"], flow_thresholds=(cfg["flow0"], cfg["flow1"]))
                print(horizon.report("for scale -- the ORACLE'S OWN C code compiled with -ffp-contract=fast -mfma against the pinned (contraction-free) oracle, same run", sf))
                st["oracle_fma_build_vs_oracle"] = dict(first_discrete_divergence=sf["first_discrete_divergence"], first_pose_divergence=sf["first_pose_divergence"],
                                                        abs_delta_ate=sf["abs_delta_ate"], rmse=sf["trajectory_rmse_gpu_vs_oracle"])
        out[name] = st
    del seq
    out["scenes"] = scenes_section(M, ctx, orc, frontend, horizon, sequence, frames)
    for title, B, pairs, k, N in (("mbavo_lm_batch, ALL 64 pairs of configs[2], k = 4 N = 4", 64, range(64), 4, 4),
                                  ("mbavo_lm_batch, ALL 64 pairs of configs[2], k = 2 N = 2", 64, range(64), 2, 2),
                                  ("mbavo_lm_batch, configs[3]'s 512 pairs (two groups, late slots re-tiled), every 8th pair compared, k = 4 N = 4", 512, range(0, 512, 8), 4, 4)):
        batch = workloads.RenderedPairBatch(ctx, B, H=480, W=640, S=8, k=k, seed=1)
        st = horizon.lm_batch_vs_oracle(orc, M, ctx, batch, pairs, k, N, 0)
        print("== %s, solver type 0, against the ORACLE's loop" % title)
        print("pairs compared %d, LM records (oracle) %d, accepted steps (gpu) %d; pairs with different (iter, kind, outliers) records: %s"
              % (st["pairs_compared"], st["lm_records_oracle"], st["accepted_steps_gpu"], st["pairs_with_different_records"] or "none"))
        print("final cost max rel diff %.3e, pose at capture time max abs diff %.3e, ATE gpu %.6e oracle %.6e |dATE| %.3e; min |quality - threshold| %s; oracle %.1f s"
              % (st["final_cost_max_rel_diff"], st["pose_max_abs_diff"], st["ate_gt_gpu"], st["ate_gt_oracle"], st["abs_delta_ate"],
                 st["min_quality_margin"], st["oracle_seconds"]))
        out[title] = st
        del batch
    print("JSON " + json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
