"""Seeded random sweep of the fused evaluation against the oracle: sizes, blur samples, patch sizes, frames, spline
degree, keypoint layouts and outlier flags drawn at random, so that every kernel variant (lane-per-pixel with and
without a sample-parallel remainder round, the sample-parallel kernel, fp32 / cost-only) and the tile boundaries are hit
in combinations the hand-written cases do not list.  Tolerance as everywhere: 1e-9 relative on the packed blocks; per-patch
costs exact up to rare one-ulp fp32 weight flips (see below)."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _tol(sc):
    """1e-9, or what ONE flipped fp32 bilinear weight may do to a SMALL problem: the kernel contracts the warp to FMAs,
    the oracle does not, and once in ~1e5 taps the last fp64 bit moves a fractional coordinate across an fp32 rounding
    boundary -- that pixel's intensity then changes by ~1e-7 of its value, i.e. the normalised sums by up to
    ~2e-6 / (number of residuals)."""
    return max(1e-9, 2e-6 / max(sc.K * sc.F * sc.P, 1))


@pytest.mark.parametrize("seed", range(24))
def test_random_problem_matches_oracle(orc, mbavo, gpu_ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([2, 4]))
    S = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 32, 64]))
    P = int(rng.choice([1, 1, 3, 8, 8, 13]))  # seeds fixed: do not reorder (see test_patch_sizes for 2 .. 128)
    F = int(rng.choice([1, 1, 2, 3]))
    dense = P == 1 and rng.random() < 0.6
    kw = dict(S=S, F=F, k=k, P=P, seed=seed + 50)
    if dense:
        H, W = int(rng.integers(12, 60)), int(rng.integers(16, 90))
        kw.update(H=H, W=W, kp="dense", margin=int(rng.integers(0, 3)))
    else:
        kw.update(K=int(rng.integers(1, 700)), kp=str(rng.choice(["random", "border"])))
    if S >= 32:
        kw.update(trans_scale=0.002, rot_scale=0.02)
    if rng.random() < 0.3:
        kw.update(outlier_frac=0.15)
    if rng.random() < 0.3:
        kw.update(huber=float(rng.choice([0.1, 1.0, 30.0])))
    sc = scenes.Scene(**kw)
    p, keep = sc.oracle_problem(orc)
    ro = orc.evaluate(p)
    d = scenes.DeviceScene(sc, vec2d=bool(seed % 2) and not dense)
    fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, [d], k)
    assert _rel(fb, ro["frame_blocks"]) < _tol(sc), kw
    E = sc.E
    want_pc = ro["patch_blocks"].reshape(-1, E)[:, 0]
    # per-patch costs: exact, except where an fp64 rounding difference of the warp (the kernel contracts to FMAs, the
    # oracle does not) moves a tap coordinate across an fp32 rounding boundary -- one bilinear weight then changes by
    # one ulp (6e-8 of that pixel's intensity); at most a few patches per thousand
    got_pc = pc.ravel()[:want_pc.size]
    mism = got_pc != want_pc
    assert mism.sum() <= max(2, int(0.01 * want_pc.size)), (int(mism.sum()), want_pc.size, kw)
    assert np.abs(got_pc - want_pc).max() <= 1e-5 * max(np.abs(want_pc).max(), 1e-300), kw
    fc, _, _ = scenes.gpu_eval_batch(gpu_ctx, [d], k, with_hessian=False)
    assert np.abs(fc[:, 0] - fb[:, 0]).max() <= 1e-11 * max(np.abs(fb[:, 0]).max(), 1e-300), kw


@pytest.mark.parametrize("seed", range(16))
def test_random_problem_packed_keyframe_matches_float(mbavo, gpu_ctx, seed):
    """The same random problems with the keyframe handed over packed (one word per pixel, mbavo_problem.grad_fp16 = 2) and as the
    u8 image + float gradient image: every tap value is the same, so the two agree to 1e-12 relative on the packed blocks (small
    lists take the sample-parallel kernel with the float image and the lane-per-pixel kernel with the packed one: another order
    of sums), with identical valid-pixel counts; border keypoints (taps next to the zero-gradient border) included."""
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.choice([2, 4]))
    S = int(rng.choice([1, 2, 3, 4, 5, 8, 16]))
    P = int(rng.choice([1, 1, 3, 8, 8, 13]))
    F = int(rng.choice([1, 1, 2, 3]))
    dense = P == 1 and rng.random() < 0.6
    kw = dict(S=S, F=F, k=k, P=P, seed=seed + 90)
    if dense:
        kw.update(H=int(rng.integers(12, 60)), W=int(rng.integers(16, 90)), kp="dense", margin=int(rng.integers(0, 3)))
    else:
        kw.update(K=int(rng.integers(1, 700)), kp=str(rng.choice(["random", "border"])))
    if rng.random() < 0.3:
        kw.update(outlier_frac=0.15)
    sc = scenes.Scene(**kw)
    fb0, pc0, v0 = scenes.gpu_eval_batch(gpu_ctx, [scenes.DeviceScene(sc)], k)
    fb2, pc2, v2 = scenes.gpu_eval_batch(gpu_ctx, [scenes.DeviceScene(sc, packed=True)], k)
    assert np.array_equal(v0, v2), kw
    assert _rel(fb2, fb0) < 1e-12, kw
    assert np.abs(pc2 - pc0).max() <= 1e-12 * max(np.abs(pc0).max(), 1e-300), kw


def _random_scene_kwargs(rng, k, seed, S=None):
    S = int(rng.choice([1, 2, 3, 4, 8, 16])) if S is None else S
    P = int(rng.choice([1, 2, 3, 4, 8, 16]))
    kw = dict(S=S, F=int(rng.choice([1, 1, 2])), k=k, P=P, seed=seed)
    if P == 1 and rng.random() < 0.5:
        kw.update(H=int(rng.integers(12, 50)), W=int(rng.integers(16, 70)), kp="dense", margin=int(rng.integers(0, 3)))
    else:
        kw.update(K=int(rng.integers(1, 400)), kp=str(rng.choice(["random", "border"])))
    if rng.random() < 0.3:
        kw.update(outlier_frac=0.15)
    return kw


@pytest.mark.parametrize("seed", range(10))
def test_random_batch_matches_oracle(orc, mbavo, gpu_ctx, seed):
    """Several unrelated problems in ONE mbavo_eval_batch call (sizes, S, P and frame counts mixed, or all with the same S
    so that the sample-parallel kernel takes the list): every problem's frame blocks and patch costs must be what the
    oracle gives for that problem alone -- the offsets into the shared pose table, partials, patch costs and tiles are
    what is being exercised."""
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.choice([2, 4]))
    B = int(rng.integers(2, 7))
    same_S = int(rng.choice([4, 8, 16])) if seed % 2 else None
    kws = [_random_scene_kwargs(rng, k, 300 + 10 * seed + b, S=same_S) for b in range(B)]
    scs = [scenes.Scene(**kw) for kw in kws]
    dscs = [scenes.DeviceScene(sc) for sc in scs]
    fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, dscs, k)
    fc, _, _ = scenes.gpu_eval_batch(gpu_ctx, dscs, k, with_hessian=False)
    f0 = p0 = 0
    for sc, kw in zip(scs, kws):
        p, keep = sc.oracle_problem(orc)
        ro = orc.evaluate(p)
        want = ro["frame_blocks"].reshape(sc.F, sc.E)
        assert _rel(fb[f0:f0 + sc.F], want) < _tol(sc), kw
        assert np.abs(fc[f0:f0 + sc.F, 0] - want[:, 0]).max() <= _tol(sc) * max(np.abs(want[:, 0]).max(), 1e-300), kw
        want_pc = ro["patch_blocks"].reshape(-1, sc.E)[:, 0]
        got_pc = pc[p0:p0 + want_pc.size]
        assert (got_pc != want_pc).sum() <= max(2, int(0.01 * want_pc.size)), kw
        assert np.abs(got_pc - want_pc).max() <= 1e-5 * max(np.abs(want_pc).max(), 1e-300), kw
        f0 += sc.F
        p0 += want_pc.size


@pytest.mark.parametrize("P,S", [(2, 3), (4, 5), (8, 3), (16, 3), (32, 5), (64, 3), (128, 3), (12, 3)])
def test_patch_sizes_sum_in_reference_order(orc, mbavo, gpu_ctx, P, S):
    """Per-patch costs for patch sizes 2 .. 128 with an odd number of blur samples (residuals with full 53-bit
    mantissas, so the ORDER of the sum over a patch's pixels shows): power-of-two patches follow the stride-halving tree of
    the reference's shared-memory reduce() (reduction.h:13-55), the others the ascending sum (A10) -- equal to the
    oracle's bit for bit up to the rare fp32 weight flips."""
    sc = scenes.Scene(S=S, F=2, k=4, P=P, K=150, kp="random", seed=900 + P)
    p, keep = sc.oracle_problem(orc)
    ro = orc.evaluate(p)
    d = scenes.DeviceScene(sc)
    fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, [d], 4)
    assert _rel(fb, ro["frame_blocks"]) < 1e-9
    want_pc = ro["patch_blocks"].reshape(-1, sc.E)[:, 0]
    got_pc = pc.ravel()[:want_pc.size]
    assert (got_pc != want_pc).sum() <= max(1, int(0.01 * want_pc.size)), int((got_pc != want_pc).sum())
    assert np.abs(got_pc - want_pc).max() <= 1e-5 * np.abs(want_pc).max()


def test_flat_tolerance_failure_rate_is_bounded(orc, mbavo, gpu_ctx, monkeypatch):
    """The widened small-problem tolerance of _tol() must not hide a NEW class of difference: with a FLAT 1e-9 on the
    packed blocks the known one-ulp fp32-weight flips fail about once in 2 000 random problems (profiles/r02_fuzz.txt:
    6 of 12 000).  200 further seeds: at most ONE may exceed the flat bound, and a failure must stay inside _tol()."""
    import sys
    mod = sys.modules[__name__]
    widened = _tol
    worst = []
    monkeypatch.setattr(mod, "_tol", lambda sc: 1e-9)
    for seed in range(5000, 5200):
        try:
            test_random_problem_matches_oracle(orc, mbavo, gpu_ctx, seed)
        except AssertionError:
            worst.append(seed)
    monkeypatch.setattr(mod, "_tol", widened)
    for seed in worst:  # whatever exceeded the flat bound is one of the small-problem flips: inside the stated tolerance
        test_random_problem_matches_oracle(orc, mbavo, gpu_ctx, seed)
    assert len(worst) <= 1, worst
