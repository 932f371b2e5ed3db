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


@pytest.mark.parametrize("seed", range(24))
def test_random_problem_matches_oracle(orc, mbavo, gpu_ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([2, 4]))
    S = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 32, 64]))
    P = int(rng.choice([1, 1, 3, 8, 8, 13]))
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
    assert _rel(fb, ro["frame_blocks"]) < 1e-9, kw
    E = sc.E
    want_pc = ro["patch_blocks"].reshape(-1, E)[:, 0]
    # per-patch costs: exact, except where an fp64 rounding difference of the warp (the kernel contracts to FMAs, the
    # oracle does not) moves a tap coordinate across an fp32 rounding boundary -- one bilinear weight then changes by
    # one ulp (6e-8 of that pixel's intensity); at most a few patches per thousand
    got_pc = pc.ravel()[:want_pc.size]
    mism = got_pc != want_pc
    assert mism.sum() <= max(1, int(0.01 * want_pc.size)), (int(mism.sum()), want_pc.size, kw)
    assert np.abs(got_pc - want_pc).max() <= 1e-5 * max(np.abs(want_pc).max(), 1e-300), kw
    fc, _, _ = scenes.gpu_eval_batch(gpu_ctx, [d], k, with_hessian=False)
    assert np.abs(fc[:, 0] - fb[:, 0]).max() <= 1e-11 * max(np.abs(fb[:, 0]).max(), 1e-300), kw
