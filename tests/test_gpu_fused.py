"""-m gpu: the fused evaluation (mbavo_eval / mbavo_eval_batch) against the oracle's
evaluate_cost_hessian_gradient restatement.  Tolerances: 1e-9 relative on the fp64 blocks
(observed ~1e-15), exact validity counts; stated per assertion."""
import ctypes as C

import numpy as np
import pytest

import scenes
from mba_vo_amd import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


CASES = {
    "k4_S8_P8_F2": dict(S=8, F=2, k=4, P=8, K=145),
    "k2_S8_P8": dict(S=8, F=1, k=2, P=8, K=145),
    "k4_S1_dense_sharp": dict(H=96, W=128, S=1, F=1, k=4, P=1, kp="dense", margin=0),   # config 0 shape, small
    "k2_S1_dense_sharp": dict(H=60, W=80, S=1, F=1, k=2, P=1, kp="dense", margin=0),
    "k4_S8_P5_border_outliers": dict(S=8, F=2, k=4, P=5, K=300, kp="border", outlier_frac=0.1),
    "k4_S16_P8_F3": dict(S=16, F=3, k=4, P=8, K=500),
    "k2_S4_huber_small": dict(S=4, F=1, k=2, P=8, K=145, huber=0.1),
    "k4_S3_nonpow2": dict(S=3, F=1, k=4, P=7, K=100),
    "k4_ramp_same": dict(S=8, F=2, k=4, P=8, K=145, image="ramp", cur="same"),
    "k4_K1": dict(S=8, F=1, k=4, P=8, K=1),
    "k4_P128": dict(S=4, F=1, k=4, P=128, K=9),
    "k4_6knots_C5shape": dict(H=135, W=240, S=16, F=2, k=4, P=1, kp="dense", margin=2, N=6),
    "k4_S64_F16_max_sizes": dict(S=64, F=16, k=4, P=8, K=25, trans_scale=0.002, rot_scale=0.02, exp=0.3),   # reference maxima: 64 samples, 16 frames
    "k2_S64_F16_max_sizes": dict(S=64, F=16, k=2, P=8, K=25, trans_scale=0.002, rot_scale=0.02, exp=0.3),
    # small problems with S = 4 .. 32 take the sample-parallel kernel (one lane per blur sample)
    "k4_S32_P8_sample_parallel": dict(S=32, F=2, k=4, P=8, K=60, trans_scale=0.002, rot_scale=0.02),
    "k2_S32_P8_sample_parallel": dict(S=32, F=1, k=2, P=8, K=60, trans_scale=0.002, rot_scale=0.02),
    "k4_S4_P3_sample_parallel": dict(S=4, F=2, k=4, P=3, K=77),
    # S = 64: lane-per-pixel kernel whose whole (tiny) tile is the sample-parallel remainder round, one pixel per wave
    "k4_S64_K1_remainder_round": dict(S=64, F=1, k=4, P=8, K=1, trans_scale=0.002, rot_scale=0.02, exp=0.3),
    "k2_S64_K1_remainder_round": dict(S=64, F=2, k=2, P=5, K=2, trans_scale=0.002, rot_scale=0.02, exp=0.3),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_fused_eval_matches_oracle(orc, mbavo, gpu_ctx, name):
    import torch
    sc = scenes.Scene(**CASES[name])
    p, keep = sc.oracle_problem(orc)
    ro = orc.evaluate(p)
    d = scenes.DeviceScene(sc, vec2d=("P8" in name))
    E = sc.E
    pb = torch.full((sc.F * sc.K * E,), -1.0, dtype=torch.float64, device="cuda:0")
    rg = scenes.gpu_eval(gpu_ctx, d, patch_blocks=pb)
    assert abs(rg["cost"] - ro["cost"]) <= RTOL * abs(ro["cost"])
    assert _rel(rg["H"], ro["H"]) < RTOL and _rel(rg["g"], ro["g"]) < RTOL
    assert np.array_equal(rg["H"], rg["H"].T)
    # slot 0 of every patch block, stride E, like cuda_patch_cost_gradient_hessian_tR; other slots untouched
    pbh = pb.cpu().numpy().reshape(sc.F, sc.K, E)
    assert _rel(pbh[:, :, 0], ro["patch_blocks"][:, :, 0]) < RTOL
    assert (pbh[:, :, 1:] == -1.0).all()
    # cost-only mode (nullptr, nullptr): same cost, computed without the gradient taps
    rc = scenes.gpu_eval(gpu_ctx, d, with_hessian=False)
    roc = orc.evaluate(p, with_hessian=False)
    assert abs(rc["cost"] - roc["cost"]) <= RTOL * abs(roc["cost"])
    assert abs(rc["cost"] - rg["cost"]) <= 1e-12 * abs(rg["cost"])
    # batch entry point on the same problem: per-frame packed blocks, patch costs, valid-pixel counts
    fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, [d], sc.k)
    assert _rel(fb, ro["frame_blocks"]) < RTOL
    assert _rel(pc, ro["patch_blocks"][:, :, 0].ravel()) < RTOL
    res = np.zeros(sc.F * sc.K * sc.P)
    # validity from the oracle's stage 3 (a pixel is valid iff current pixel and all S warps are in bounds)
    o_valid = _oracle_valid_counts(orc, sc)
    assert np.array_equal(valid, o_valid)


def _oracle_valid_counts(orc, sc):
    O = orc.lib()
    k, S, F, K, P = sc.k, sc.S, sc.F, sc.K, sc.P
    poses, Jt, JR = np.zeros(F * S * 7), np.zeros(F * S * 9 * k), np.zeros(F * S * 12 * k)
    O.orc_compute_virtual_camera_poses(S, F, orc.dp(sc.cap), orc.dp(sc.exp), k, sc.t0, sc.dt, orc.dp(sc.knots_t),
                                       orc.dp(sc.knots_R), orc.dp(poses), orc.dp(Jt), orc.dp(JR), None)
    c = np.zeros(F * K * 2)
    O.orc_compute_local_patches_xy(S, F, orc.dp(poses), orc.dp(sc.kp_xy), orc.dp(sc.kp_z), K, orc.dp(sc.intr), orc.dp(c))
    # use an all-255 current image so that a valid pixel can never have residual exactly 0
    cur = np.full((sc.H, sc.W), 255, np.uint8)
    ref = np.minimum(sc.ref, 254)
    curs = (orc.c_u8p * F)(*[orc.u8p(cur)] * F)
    res = np.zeros(F * K * P)
    O.orc_compute_pixel_jacobian_residual(orc.u8p(ref), None, curs, S, F, orc.dp(poses), k, None, None, orc.dp(c),
                                          orc.dp(sc.kp_z), K, orc.ip(sc.pattern), P, orc.dp(sc.intr), sc.H, sc.W,
                                          orc.dp(res), None)
    return (res.reshape(F, K * P) != 0).sum(1).astype(np.float64)


def test_batch_of_mixed_problems(orc, mbavo, gpu_ctx):
    """One launch over problems of different image size, S, P, F (pyramid levels / independent pairs)."""
    kws = [dict(S=8, F=1, k=4, P=8, K=145, seed=1), dict(H=240, W=320, S=4, F=2, k=4, P=1, kp="dense", margin=60, seed=2),
           dict(H=120, W=160, S=8, F=1, k=4, P=5, K=77, seed=3, margin=10), dict(S=1, F=3, k=4, P=8, K=33, seed=4)]
    scs = [scenes.Scene(**kw) for kw in kws]
    ds = [scenes.DeviceScene(s) for s in scs]
    fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, ds, 4)
    row, prow = 0, 0
    for sc in scs:
        p, keep = sc.oracle_problem(orc)
        ro = orc.evaluate(p)
        assert _rel(fb[row:row + sc.F], ro["frame_blocks"]) < RTOL
        n = sc.F * sc.K
        assert _rel(pc[prow:prow + n], ro["patch_blocks"][:, :, 0].ravel()) < RTOL
        row += sc.F
        prow += n
    # a second call with the same list must reuse the cached layout and give identical bits
    fb2, _, _ = scenes.gpu_eval_batch(gpu_ctx, ds, 4)
    assert np.array_equal(fb, fb2)


def test_out_of_range_spline_is_reported(orc, mbavo, gpu_ctx):
    """The reference reads past the knot array when a blur sample leaves the spline (SplineFunctor.h:13-19
    has no check); here the evaluation returns MBAVO_E_RANGE."""
    sc = scenes.Scene(S=8, F=1, k=4, P=8, K=50)
    sc.cap = sc.cap + 10.0
    d = scenes.DeviceScene(sc)
    p = d.problem()
    cost = np.zeros(1)
    rc = gpu_ctx.lib.mbavo_eval(gpu_ctx.handle, C.byref(p), 4, mbavo.capi.dp(cost), None, None, None)
    assert rc == -2


def test_bad_arguments_rejected(mbavo, gpu_ctx):
    sc = scenes.Scene(S=8, F=1, k=4, P=8, K=50)
    d = scenes.DeviceScene(sc)
    p = d.problem()
    cost = np.zeros(1)
    assert gpu_ctx.lib.mbavo_eval(gpu_ctx.handle, C.byref(p), 3, mbavo.capi.dp(cost), None, None, None) == -1
    p.d_ref_img = None
    assert gpu_ctx.lib.mbavo_eval(gpu_ctx.handle, C.byref(p), 4, mbavo.capi.dp(cost), None, None, None) == -1
    p = d.problem()
    p.H, p.W = 1 << 15, 1 << 15  # 2^30 pixels: beyond the 32-bit tap offsets (rejected before anything is launched)
    assert gpu_ctx.lib.mbavo_eval(gpu_ctx.handle, C.byref(p), 4, mbavo.capi.dp(cost), None, None, None) == -1


def test_zero_motion_dense_integer_centres(orc, mbavo, gpu_ctx):
    """Identity spline + dense integer keypoints: every patch centre lands (up to rounding) ON an integer and is
    then truncated (compute_hessian_gradients_cost.cu:69-70), so one differing last bit would read a different
    pixel.  The centre computation is kept free of FMA contraction; residual-derived outputs must match exactly."""
    sc = scenes.Scene(H=120, W=160, S=4, F=1, k=4, P=1, kp="dense", margin=0, z_range=(3.0, 30.0))
    sc.knots_t[:] = 0.0
    sc.knots_R[:] = np.tile([0.0, 0.0, 0.0, 1.0], sc.N)
    sc.intr = np.array([83.3, 79.9, 80.2, 59.7])   # non-dyadic intrinsics: (z*(x-cx)/fx)/z*fx+cx is not exact
    p, keep = sc.oracle_problem(orc)
    ro = orc.evaluate(p)
    d = scenes.DeviceScene(sc)
    fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, [d], sc.k)
    assert np.array_equal(valid, _oracle_valid_counts(orc, sc))
    assert np.array_equal(pc, ro["patch_blocks"][:, :, 0].ravel())      # per-pixel Huber cost: bit-exact
    assert _rel(fb, ro["frame_blocks"]) < RTOL


def test_rccl_allreduce_entry_point(mbavo, gpu_ctx):
    """mbavo_allreduce_blocks on a 1-rank RCCL communicator (the box has one GPU): in-place sum == identity;
    exercises the run-time binding to librccl and the stream ordering with the evaluation."""
    import ctypes as C
    import torch
    rccl = C.CDLL("librccl.so.1")
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    sc = scenes.Scene(S=4, F=2, k=4, P=8, K=60)
    d = scenes.DeviceScene(sc)
    arr = (mbavo.capi.Problem * 1)(d.problem())
    fb = torch.zeros(sc.F * sc.E, dtype=torch.float64, device="cuda:0")
    assert gpu_ctx.lib.mbavo_eval_batch(gpu_ctx.handle, 1, arr, 4, 1, fb.data_ptr(), None, None) == 0
    ref = fb.clone()
    assert gpu_ctx.lib.mbavo_allreduce_blocks(gpu_ctx.handle, comm, fb.data_ptr(), fb.numel()) == 0
    torch.cuda.synchronize()
    assert torch.equal(fb, ref) and float(ref.abs().max()) > 0
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


def test_empty_and_ragged_batches(orc, mbavo, gpu_ctx):
    """K = 0 (no keypoints survived detection) alone and inside a batch: all-zero blocks instead of the reference's
    division by zero; the neighbours in the batch are unaffected."""
    full = scenes.Scene(S=4, F=2, k=4, P=8, K=60, seed=3)
    empty = scenes.Scene(S=4, F=2, k=4, P=8, K=60, seed=4)
    empty.kp_xy, empty.kp_z, empty.K = np.zeros((0, 2)), np.zeros(0), 0
    tiny = scenes.Scene(S=8, F=1, k=4, P=3, K=1, seed=5)
    ds = [scenes.DeviceScene(full), scenes.DeviceScene(empty), scenes.DeviceScene(tiny)]
    # zero-length device tensors have a null data_ptr: give the empty problem valid (unused) pointers
    ds[1].kp_xy_ptr, ds[1].kp_z = ds[0].kp_xy_ptr, ds[0].kp_z
    fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, ds, 4)
    assert np.all(fb[2:4] == 0.0) and np.all(valid[2:4] == 0.0)
    for sc, rows in ((full, slice(0, 2)), (tiny, slice(4, 5))):
        p, keep = sc.oracle_problem(orc)
        ro = orc.evaluate(p)
        assert _rel(fb[rows], ro["frame_blocks"]) < RTOL
    alone, _, _ = scenes.gpu_eval_batch(gpu_ctx, [ds[1]], 4)
    assert np.all(alone == 0.0)


def test_non_finite_and_wild_knots_do_not_fault(mbavo, gpu_ctx):
    """A diverging LM step can hand the path NaN / huge control knots: every tap coordinate is then out of bounds or
    NaN, the pixels are invalid (zero residual, zero row, A9) and nothing is read outside the images."""
    import torch
    for mode in ("nan", "huge", "inf"):
        sc = scenes.Scene(S=8, F=1, k=4, P=8, K=200, seed=9)
        if mode == "nan":
            sc.knots_t[4] = np.nan
            sc.knots_R[5] = np.nan
        elif mode == "huge":
            sc.knots_t[:] = 1e300
        else:
            sc.knots_t[2] = np.inf
        d = scenes.DeviceScene(sc)
        fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, [d], 4)
        torch.cuda.synchronize()
        assert valid.sum() == 0 or mode == "nan"  # NaN in one knot may leave samples of other segments valid
        finite = np.isfinite(fb)
        assert finite.all() or mode != "huge"
    # the context is still usable afterwards
    ok = scenes.Scene(S=4, F=1, k=4, P=8, K=50, seed=2)
    fb, _, valid = scenes.gpu_eval_batch(gpu_ctx, [scenes.DeviceScene(ok)], 4)
    assert np.isfinite(fb).all() and valid.sum() > 0


def test_many_small_problems_flat_finalize(orc, mbavo, gpu_ctx):
    """70 small problems in one call (>= 64 slots of <= 4 tiles each: the one-block-per-slot finalize kernel) against the
    same problems evaluated alone (the tree finalize kernel): the tiles are added in the same order, so the blocks of a
    problem whose tiling is the same in both calls are identical bit for bit (a problem of at most one sample-parallel
    round, 64 pixels since round 3, is one tile either way); otherwise only the grouping of the sum differs (1e-13); all
    of them match the oracle."""
    rng = np.random.default_rng(77)
    scs = [scenes.Scene(S=8, F=1 + (i % 2), k=4, P=8, K=int(rng.integers(5, 40)), seed=500 + i) for i in range(70)]
    ds = [scenes.DeviceScene(s) for s in scs]
    fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, ds, 4)
    row = 0
    for i, sc in enumerate(scs):
        if i % 7 == 0:
            p, keep = sc.oracle_problem(orc)
            ro = orc.evaluate(p)
            assert _rel(fb[row:row + sc.F], ro["frame_blocks"]) < RTOL
        row += sc.F
    row = 0
    exact = 0
    for i, sc in enumerate(scs):
        if i % 9 == 3 or sc.K * sc.P <= 64:
            one, _, _ = scenes.gpu_eval_batch(gpu_ctx, [ds[i]], 4)
            got = fb[row:row + sc.F]
            assert np.abs(one - got).max() <= 1e-13 * np.abs(got).max()
            if sc.K * sc.P <= 64:  # one tile per slot in both calls -> identical sums
                assert np.array_equal(one, got)
                exact += 1
        row += sc.F
    assert exact >= 3


@pytest.mark.parametrize("sp", ["0", "1"])
def test_non_finite_depth_is_dropped_in_both_kernels(orc, mbavo, sp):
    """A keypoint with inf / NaN depth passes the detector's `!(z < 1e-2)` test (blur_aware_direct_tracker.cpp:403-405)
    and reaches the path: its pixels must be dropped (invalid: zero residual, zero row), not weighted by zero --
    0 * NaN would turn the frame's H and g into NaN.  Both the lane-per-pixel kernel (mbavo_engine_opts.sample_parallel = -1) and
    the sample-parallel kernel (= 1) against the oracle."""
    import torch
    ctx = mbavo.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    ctx.engine_opts(sample_parallel=1 if sp == "1" else -1)
    try:
        for k in (4, 2):
            sc = scenes.Scene(S=8, F=2, k=k, P=8, K=90, seed=21)
            sc.kp_z[3], sc.kp_z[40], sc.kp_z[77] = np.inf, np.nan, -np.inf
            p, keep = sc.oracle_problem(orc)
            ro = orc.evaluate(p)
            assert np.isfinite(ro["frame_blocks"]).all()
            fb, pc, valid = scenes.gpu_eval_batch(ctx, [scenes.DeviceScene(sc)], k)
            assert np.isfinite(fb).all() and np.isfinite(pc).all()
            assert _rel(fb, ro["frame_blocks"]) < RTOL
            assert np.array_equal(valid, _oracle_valid_counts(orc, sc)) and valid.max() <= (sc.K - 3) * sc.P
            assert (pc.reshape(sc.F, sc.K)[:, [3, 40, 77]] == 0.0).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["k4_dense_S8", "k4_dense_S1", "k4_P8_lane_per_pixel", "k4_batch_mixed_S", "k2_dense_S8", "k2_P8_three_frames",
                                  "k2_batch_mixed_S"])
def test_pose_prologue_equals_pose_kernel(orc, mbavo, name):
    """Tiles <= CUs, S <= 8: the pose entries are the fused kernel's prologue (k_fused<.., POSE>, two launches);
    mbavo_engine_opts.fused_pose = -1 keeps k_pose_table (three launches).  Same arithmetic per entry -> the frame blocks, per-patch
    costs and valid counts must be IDENTICAL, H/g and cost-only, and both must match the oracle (1e-9).  k = 2 -- the
    reference's default degree, blur_aware_direct_tracker.h:50 -- takes the prologue since round 4 (through the two stages:
    the one-lane chain spilled at the k = 2 kernels' 128-register budget)."""
    import torch
    kws = {"k4_dense_S8": [dict(H=96, W=128, S=8, F=2, k=4, P=1, kp="dense", margin=0)],
           "k4_dense_S1": [dict(H=60, W=80, S=1, F=1, k=4, P=1, kp="dense", margin=0)],
           "k4_P8_lane_per_pixel": [dict(S=8, F=3, k=4, P=8, K=211, N=6)],
           "k4_batch_mixed_S": [dict(S=8, F=1, k=4, P=8, K=97, seed=3), dict(S=4, F=2, k=4, P=5, K=60, seed=4),
                                dict(S=2, F=1, k=4, P=8, K=31, seed=5)],
           "k2_dense_S8": [dict(H=96, W=128, S=8, F=2, k=2, P=1, kp="dense", margin=0)],
           "k2_P8_three_frames": [dict(S=8, F=3, k=2, P=8, K=211, N=4)],
           "k2_batch_mixed_S": [dict(S=8, F=1, k=2, P=8, K=97, seed=3), dict(S=4, F=2, k=2, P=5, K=60, seed=4),
                                dict(S=1, F=1, k=2, P=8, K=31, seed=5)]}[name]
    kdeg = 2 if name.startswith("k2") else 4
    scs = [scenes.Scene(**kw) for kw in kws]
    got = {}
    for mode in ("0", "1"):
        ctx = mbavo.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        ctx.engine_opts(sample_parallel=-1, fused_pose=1 if mode == "1" else -1)  # the lane-per-pixel kernel whatever the size
        try:
            ds = [scenes.DeviceScene(sc) for sc in scs]
            fb, pc, valid = scenes.gpu_eval_batch(ctx, ds, kdeg)
            kern = ctx.lib.mbavo_last_kernel(ctx.handle).decode()
            fbc, pcc, _ = scenes.gpu_eval_batch(ctx, ds, kdeg, with_hessian=False)
            got[mode] = (fb.copy(), pc.copy(), valid.copy(), fbc.copy(), pcc.copy(), kern)
        finally:
            ctx.close()
    assert got["0"][5].endswith(",false>") and got["1"][5].endswith(",true>"), (got["0"][5], got["1"][5])
    for a, b in zip(got["0"][:5], got["1"][:5]):
        assert np.array_equal(a, b)
    row = 0
    for sc in scs:
        p, keep = sc.oracle_problem(orc)
        ro = orc.evaluate(p)
        assert _rel(got["1"][0][row:row + sc.F], ro["frame_blocks"]) < RTOL
        row += sc.F


def test_patch_centre_near_integer_takes_reference_order(orc, mbavo, gpu_ctx):
    """The round loop keeps the cheap patch centre (rotation matrix, fused arithmetic) only where centre + offset is
    farther than 1e-5 from an integer; otherwise the wave evaluates the reference's operation order.  Zero motion puts
    EVERY pixel of a dense grid exactly on an integer (all waves take the reference order); a translation that moves the
    projections by ~1e-7 px puts them inside the guard band with fractions the two orders could truncate differently.
    Valid counts exact, blocks 1e-9 against the oracle in both."""
    for scale in (0.0, 1e-10):
        sc = scenes.Scene(H=60, W=80, S=4, F=1, k=4, P=1, kp="dense", margin=0, trans_scale=scale, rot_scale=scale, seed=9)
        p, keep = sc.oracle_problem(orc)
        ro = orc.evaluate(p)
        fb, pc, valid = scenes.gpu_eval_batch(gpu_ctx, [scenes.DeviceScene(sc)], 4)
        assert np.array_equal(valid, _oracle_valid_counts(orc, sc))
        assert _rel(fb, ro["frame_blocks"]) < RTOL
        assert _rel(pc, ro["patch_blocks"][:, :, 0].ravel()) < RTOL


# ---------------------------------------------------------------------------------------------------------------------------
# a8 inside the evaluation: mbavo_eval_batch_merged ends in the reference's unit
# [cost | g | H] per problem (spline_update_step.cpp:232-239 -> merge_hessian_gradient_cost.cpp:39-86) -- stored by the finalize
# step itself for lists of one-frame problems on N == k knots, by the merge kernel behind it otherwise.
MERGED_CASES = {
    # finalize KERNEL (one dense problem, many tiles), F = 1, N == k: fused
    "dense_k4": ([dict(H=120, W=160, S=4, F=1, k=4, P=1, kp="dense", margin=2)], 4),
    "dense_k2": ([dict(H=120, W=160, S=4, F=1, k=2, P=1, kp="dense", margin=2)], 2),
    # sample-parallel single launch (ticket epilogue), F = 1, N == k: fused
    "semidense_one_launch_k4": ([dict(S=8, F=1, k=4, P=8, K=145)], 4),
    "semidense_one_launch_k2": ([dict(S=8, F=1, k=2, P=8, K=145)], 2),
    # many slots of few tiles: the flat finalize kernel, fused
    "batch_flat_k4": ([dict(H=96, W=128, S=8, F=1, k=4, P=8, K=90 + 3 * i, seed=i) for i in range(70)], 4),
    # not a plain unpack: two frames, or more knots than the degree -> the gather kernel behind the finalize
    "two_frames_k4": ([dict(S=8, F=2, k=4, P=8, K=145)], 4),
    "six_knots_k4": ([dict(H=135, W=240, S=16, F=2, k=4, P=1, kp="dense", margin=2, N=6)], 4),
    "mixed_batch_k2": ([dict(H=96, W=128, S=4, F=1 + (i % 2), k=2, P=8, K=60, seed=i) for i in range(5)], 2),
}


@pytest.mark.parametrize("name", sorted(MERGED_CASES))
def test_eval_batch_merged_equals_eval_then_merge(mbavo, gpu_ctx, name):
    """Bit-identical to mbavo_eval_batch followed by mbavo_merge_device (itself bit-exact against the reference-executed merge
    fixture, tests/test_golden_blocks.py), packed frame blocks unchanged, every system symmetric; the single-problem host path
    (mbavo_eval: host merge) gives the same H, g, cost."""
    import torch
    specs, k = MERGED_CASES[name]
    scs = [scenes.Scene(**s) for s in specs]
    ds = [scenes.DeviceScene(sc) for sc in scs]
    B, E = len(ds), synth.packed_len(k)
    arr = (mbavo.capi.Problem * B)(*[d.problem() for d in ds])
    nbf = sum(sc.F for sc in scs)
    lens = [1 + 6 * sc.N + 36 * sc.N * sc.N for sc in scs]
    fb_a = torch.zeros(nbf * E, dtype=torch.float64, device="cuda:0")
    fb_b = torch.full((nbf * E,), -3.0, dtype=torch.float64, device="cuda:0")
    sys_a = torch.full((sum(lens),), -1.0, dtype=torch.float64, device="cuda:0")
    sys_b = torch.full((sum(lens),), -2.0, dtype=torch.float64, device="cuda:0")
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    assert lib.mbavo_eval_batch(h, B, arr, k, 1, fb_a.data_ptr(), None, None) == 0
    assert lib.mbavo_merge_device(h, B, arr, k, fb_a.data_ptr(), sys_a.data_ptr()) == 0
    assert lib.mbavo_eval_batch_merged(h, B, arr, k, fb_b.data_ptr(), sys_b.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(fb_a, fb_b)
    assert torch.equal(sys_a, sys_b)
    # a plain evaluation afterwards does not write systems (the target is one-shot)
    sys_b.fill_(-7.0)
    assert lib.mbavo_eval_batch(h, B, arr, k, 1, fb_b.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    assert bool((sys_b == -7.0).all())
    off = 0
    sa = sys_a.cpu().numpy()
    for sc, d, n_sys in zip(scs, ds, lens):
        n = 6 * sc.N
        Hm = sa[off + 1 + n:off + n_sys].reshape(n, n)
        assert np.array_equal(Hm, Hm.T)
        r = scenes.gpu_eval(gpu_ctx, d)
        assert abs(r["cost"] - sa[off]) <= 1e-12 * abs(r["cost"]) and _rel(sa[off + 1:off + 1 + n], r["g"]) < 1e-12 and _rel(Hm, r["H"]) < 1e-12
        off += n_sys
    assert lib.mbavo_eval_batch_merged(h, B, arr, k, fb_b.data_ptr(), None, None, None) == -1
