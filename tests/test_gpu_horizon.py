"""Long-horizon parity of BlurAwareDirectTracker::trackFrame and of the batched LM against the ORACLE (VERDICT r04 next-round 1;
blur_aware_direct_tracker.cpp:88-203, 590-699): 300 rendered 640x480 frames with ~125 keyframe changes, k = 2 and k = 4, and
mbavo_lm_batch on ALL 64 pairs of configs[2] plus a 64-pair sample of configs[3]'s 512.

What holds and what does not (profiles/r05_long_horizon.txt has the full statistics):
  * FREE-RUNNING, the two trackers agree to 1e-8 for ~18 frames, to 1e-6 for ~23, to 1e-5 for ~29, then decorrelate to the tracker's
    own drift (ATE 3.6e-2 against 3.2e-2): the tracker is a feedback loop (constant-velocity prediction, LM stopped at a finite
    tolerance along a weakly constrained direction) that multiplies a rounding-level difference by ~1.4 per frame until a discrete
    decision flips.  Nothing specific to this implementation: the oracle's OWN C code compiled with FMA contraction leaves the
    pinned oracle at frame 1.  So "ATE within 1e-5 on the same sequence" (north_star) is asserted on the horizon where it is a
    property of the implementation and not of the rounding (the first 25 frames), and statistically over the whole run.
  * TEACHER-FORCED -- the HIP tracker put into the oracle's state before every frame (mbavo_vo_set_state) -- every one of the 300
    frames is a one-step comparison from identical inputs: keyframe decisions, start indices, keypoint counts and EVERY LM record
    (level, iteration, accepted / rejected / invalid, outlier count) identical, poses within 1e-8, |dATE| 1e-11.  This is the
    long-horizon parity statement that an implementation can be held to."""
import numpy as np
import pytest

import frontend
import horizon

pytestmark = pytest.mark.gpu

_CACHE = {}


def _long_run(orc, mbavo, gpu_ctx, frames=300):
    if "seq" not in _CACHE:
        from mba_vo_amd import sequence
        seq = sequence.make_sequence(gpu_ctx, H=480, W=640, M=frames, trajectory="loop")  # rendered on the GPU (bit-exact vs the oracle's renderer)
        cfg = dict(sequence.REFERENCE_CFG)
        _CACHE["seq"], _CACHE["cfg"] = seq, cfg
        _CACHE["want"] = frontend.run_oracle_vo(orc, seq, cfg)
        _CACHE["gt"] = frontend.gt_relative(orc, seq)
    return _CACHE["seq"], _CACHE["cfg"], _CACHE["want"], _CACHE["gt"]


def test_track_frame_300_frames_teacher_forced_k2(orc, mbavo, gpu_ctx):
    seq, cfg, want, gt = _long_run(orc, mbavo, gpu_ctx)
    assert sum(w["is_keyframe"] for w in want) >= 100  # >= 30 keyframe changes asked for; this trajectory has ~125
    got = frontend.run_gpu_vo(mbavo, gpu_ctx, seq, cfg, teacher=want)
    st = horizon.compare(got, want, gt, min_step_quality=cfg["min_quality"])
    assert st["first_discrete_divergence"] is None, st["first_divergence"]
    assert st["max_abs_pose_diff"] <= 1e-6, (st["max_abs_pose_diff"], st["max_abs_pose_diff_frame"])
    assert st["pose_diff_quantiles"]["99%"] <= 1e-7
    assert st["abs_delta_ate"] <= 1e-5 and st["abs_delta_ate_windows_max"] <= 1e-5  # north_star's bound, whole run and every 50-frame window
    assert st["trace_cost_max_rel_diff"] <= 1e-4  # costs along 4 000 LM records
    assert got[-1]["keyframe_resyncs"] == 0       # the keyframe decisions never had to be corrected


def test_track_frame_300_frames_free_running_k2(orc, mbavo, gpu_ctx):
    seq, cfg, want, gt = _long_run(orc, mbavo, gpu_ctx)
    got = frontend.run_gpu_vo(mbavo, gpu_ctx, seq, cfg)
    again = frontend.run_gpu_vo(mbavo, gpu_ctx, seq, cfg)
    assert all(np.array_equal(a["T"], b["T"]) for a, b in zip(got, again))  # 300 frames, bit-reproducible
    st = horizon.compare(got, want, gt, min_step_quality=cfg["min_quality"])
    d = np.array([np.abs(a["T"] - b["T"]).max() for a, b in zip(got, want)])
    # the horizon on which parity is the implementation's property: identical discrete results and 1e-6 for the first 15 frames
    # (observed: 24 and 23), |dATE| <= 1e-5 over the first 25 (observed 3e-7)
    assert st["first_discrete_divergence"] is None or st["first_discrete_divergence"] >= 15, st["first_divergence"]
    assert d[:15].max() <= 1e-6, d[:15].max()
    assert abs(horizon.ate(got, gt, 0, 25) - horizon.ate(want, gt, 0, 25)) <= 1e-5
    # beyond it the runs decorrelate (see the module docstring): both stay at the tracker's drift level
    assert st["ate_gt_gpu"] < 0.08 and st["ate_gt_oracle"] < 0.08
    assert 0.5 < st["ate_gt_gpu"] / st["ate_gt_oracle"] < 2.0
    assert abs(st["keyframes_gpu"] - st["keyframes_oracle"]) <= 10


def test_track_frame_300_frames_free_running_exposure_008(orc, mbavo, gpu_ctx):
    """north_star's sentence as worded -- "bit-identical pose indices and ATE within 1e-5 of the reference on the same synthetic blurred
    sequence" -- FREE-RUNNING over a whole 301-frame sequence: the same trajectory with the exposure doubled (0.08 of the 0.1 frame
    interval), where the two knots are constrained along the blur direction and a rounding-level difference does not grow to a
    flipped decision (profiles/r06_long_horizon.txt: growth x1.06 per frame against x1.16-1.41 at exposure 0.04).  Every discrete
    result of every frame identical (keyframe decisions, start indices, keypoint counts, every LM record), every pose within 1e-5
    (observed 3e-9 at frame 20), |dATE| <= 1e-5 over the run and over every 50-frame window (observed 1.5e-8)."""
    from mba_vo_amd import sequence
    seq = sequence.make_sequence(gpu_ctx, H=480, W=640, M=300, trajectory="loop", exp=0.08)
    cfg = dict(sequence.REFERENCE_CFG)
    want = frontend.run_oracle_vo(orc, seq, cfg)
    gt = frontend.gt_relative(orc, seq)
    got = frontend.run_gpu_vo(mbavo, gpu_ctx, seq, cfg)
    st = horizon.compare(got, want, gt, min_step_quality=cfg["min_quality"])
    assert st["keyframes_oracle"] >= 30 and st["lm_records_oracle"] >= 3000
    assert st["first_discrete_divergence"] is None, st["first_divergence"]
    assert st["max_abs_pose_diff"] <= 1e-5, (st["max_abs_pose_diff"], st["max_abs_pose_diff_frame"])
    assert st["abs_delta_ate"] <= 1e-5 and st["abs_delta_ate_windows_max"] <= 1e-5
    print("exposure 0.08, free-running, 301 frames: max |pose diff| %.2e at frame %d, |dATE| %.2e, %d keyframes, %d LM records"
          % (st["max_abs_pose_diff"], st["max_abs_pose_diff_frame"], st["abs_delta_ate"], st["keyframes_oracle"], st["lm_records_oracle"]))


def test_track_frame_teacher_forced_k4(orc, mbavo, gpu_ctx):
    """k = 4 through trackFrame (four identity knots through getSplineTrajectory(), minimum-norm steps): the ORACLE itself loses the
    scene on this sequence (four knots constrained by one short exposure: at frame 90 a minimum-norm step lands where no pixel is
    valid), so only the one-step form is a parity statement: identical discrete results on every frame, poses to 1e-6 while the
    oracle's tracker is alive, the losing step to 5e-3 of its length, nothing moving afterwards."""
    seq, cfg, _, gt = _long_run(orc, mbavo, gpu_ctx)
    short = dict(seq, times=seq["times"][:151])
    cfg4 = dict(cfg, k=4)
    want = frontend.run_oracle_vo(orc, short, cfg4, init_knots=4)
    got = frontend.run_gpu_vo(mbavo, gpu_ctx, short, cfg4, init_knots=4, teacher=want)
    st = horizon.compare(got, want, gt[:151], min_step_quality=cfg["min_quality"])
    assert st["first_discrete_divergence"] is None, st["first_divergence"]
    d = np.array([np.abs(a["T"] - b["T"]).max() for a, b in zip(got, want)])
    size = np.array([max(1.0, np.abs(b["T"]).max()) for b in want])
    rel = d / size
    assert np.median(rel) <= 1e-9 and np.quantile(rel, 0.9) <= 1e-6, (np.median(rel), np.quantile(rel, 0.9))
    # THE OUTLIER, not hidden under a quantile (gpurun_out / profiles: tools/k4_outlier.py).  One frame (90)
    # has a one-step pose difference of 3.8e-2 on IDENTICAL discrete records.  It is the frame in which the ORACLE's own k = 4 run
    # loses the scene: on the coarsest level an accepted step flags 46 outliers, the next accepted step is ~40 units long (the
    # minimum-norm solution of a rank-deficient 24 x 24 system: four knots, one exposure) and lands where NO pixel is valid -- final
    # cost exactly 0 on both sides, the state jumps from 1.3 to 38 and every later frame ends at cost 0.  The two ~40-unit steps agree
    # to 1e-3 of their length; before that frame the runs agree to 1e-6 absolutely, after it (one-step comparisons of a lost
    # tracker: nothing moves) to 1e-12 of the state.
    cost = np.array([b["cost"] for b in want])
    dead = np.nonzero((cost == 0.0) & (np.arange(len(cost)) > 0))[0]
    assert dead.size > 0, "the oracle's k = 4 run no longer loses the scene: re-derive this test's statement"
    lost = int(dead[0])
    assert np.array_equal(dead, np.arange(lost, len(cost))) and all(a["cost"] == 0.0 for a in got[lost:])  # lost for good, on both sides
    assert size[lost - 1] < 10.0 and size[lost] > 20.0                                                      # the jump
    assert d[1:lost].max() <= 2e-6, (int(np.argmax(d[1:lost])) + 1, float(d[1:lost].max()))               # alive: absolute (observed 1.0e-6 at frame 1, <= 1e-7 after)
    assert rel[lost] <= 5e-3, ("frame %d: |pose diff| %.3e on a state of size %.3e" % (lost, d[lost], size[lost]))  # observed 1.0e-3
    assert rel[lost + 1:].max() <= 1e-9, float(rel[lost + 1:].max())
    print("k = 4 teacher-forced: the oracle loses the scene at frame %d (state %.2f -> %.2f, final cost 0): |pose diff| there %.3e = %.2e of the state; "
          "before: max %.2e; after: max %.2e of the state" % (lost, size[lost - 1], size[lost], d[lost], rel[lost], d[1:lost].max(), rel[lost + 1:].max()))


@pytest.mark.parametrize("B,step,k,N", [(64, 1, 4, 4), (64, 1, 2, 2), (512, 8, 4, 4)])
def test_lm_batch_whole_batches_against_oracle(orc, mbavo, gpu_ctx, B, step, k, N):
    """mbavo_lm_batch on ALL 64 pairs of configs[2] (k = 4 and the reference's default k = 2) and on configs[3]'s 512 pairs (two
    groups, late slots re-tiled; every 8th pair compared) against the ORACLE's optimizePyramidLevel, not the host loop: identical
    (iteration, kind, outlier count) records for every pair, final cost 1e-5, pose at capture time 1e-5, |dATE| 1e-5."""
    from mba_vo_amd import workloads
    batch = workloads.RenderedPairBatch(gpu_ctx, B, H=480, W=640, S=8, k=k, seed=1)
    st = horizon.lm_batch_vs_oracle(orc, mbavo, gpu_ctx, batch, range(0, B, step), k, N, 0)
    assert st["pairs_with_different_records"] == [], st
    assert st["accepted_steps_gpu"] >= st["pairs_compared"]
    assert st["final_cost_max_rel_diff"] <= 1e-5 and st["pose_max_abs_diff"] <= 1e-5 and st["abs_delta_ate"] <= 1e-5, st
