"""-m gpu: batches of independent keyframe pairs (BASELINE configs[2]/[3]) built the way the configs describe them --
consecutive frames of ONE GPU-rendered blurred sequence, every pair with its own keyframe, gradient image, keypoints,
depths and knots (mba_vo_amd.workloads.RenderedPairBatch) -- against the oracle, and the pair -> rank sharding of
SURVEY.md 8e(1) (shard.ShardedEvaluation mode 'pairs': pair b on rank b % world into a contiguous slice of one zero
send buffer, ONE out-of-place all-reduce) with the ranks of a 2 / 4 / 8-rank run evaluated one after the other on this GPU.
Also the K == 0 regressions (a slot without tiles must still produce its all-zero block)."""
import ctypes as C

import numpy as np
import pytest

import scenes
from mba_vo_amd import shard, synth, workloads as wl

pytestmark = pytest.mark.gpu


def _oracle_blocks(orc, p):
    op, keep = orc.make_problem(p.S, p.F, p.K, p.P, p.k, p.N, p.H, p.W, p.ref, p.grad, p.cur, p.kp_xy, p.kp_z, p.pattern,
                                p.intr, p.cap, p.exp, p.t0, p.dt, p.knots_t, p.knots_R, p.start_idx, p.huber)
    return orc.evaluate(op)["frame_blocks"]


def test_rendered_pairs_match_oracle_and_are_consistent(orc, mbavo, gpu_ctx):
    """Six pairs at 240 x 320: every pair's packed block against the oracle on the downloaded inputs (1e-9); the pairs
    really have their own keyframes and keypoints; and the knots derived by left-multiplying the ground-truth world
    spline with the keyframe's inverse pose ARE the pose the images were rendered with: the photometric cost at the
    unperturbed relative knots is far below the cost at knots pushed 0.05 units away."""
    import torch
    batch = wl.RenderedPairBatch(gpu_ctx, 6, H=240, W=320, S=8, k=4, seed=4, perturb=2e-3)
    batch.step(gpu_ctx, True)
    torch.cuda.synchronize()
    fb = batch.frame_blocks.cpu().numpy().reshape(6, batch.E)
    valid = batch.valid.cpu().numpy()
    hosts = [batch.host_problem(b) for b in range(6)]
    for b, p in enumerate(hosts):
        assert p.K == batch.probs[b].K and p.K > 20
        ref = _oracle_blocks(orc, p)[0]
        assert np.abs(fb[b] - ref).max() <= 1e-9 * np.abs(ref).max()
        assert valid[b] > 0.9 * p.K * p.P
    assert not np.array_equal(hosts[0].ref, hosts[1].ref) and not np.array_equal(hosts[0].kp_xy[:10], hosts[3].kp_xy[:10])
    assert len({p.K for p in hosts}) >= 1 and all(np.abs(p.kp_z - 7.5).max() < 1.0 for p in hosts)
    # consistency of the relative spline: cost at the ground truth vs 0.05 units off (oracle, pair 2)
    p = hosts[2]
    h = batch._host[2]
    at_gt = wl.Prob(p.ref, p.cur, p.kp_xy, p.kp_z, p.pattern, p.intr, p.S, p.k, 4, p.cap, p.exp, p.t0, p.dt, h["kt_gt"], h["kR"],
                    p.huber, grad=p.grad)
    off = wl.Prob(p.ref, p.cur, p.kp_xy, p.kp_z, p.pattern, p.intr, p.S, p.k, 4, p.cap, p.exp, p.t0, p.dt,
                  h["kt_gt"] + np.array([0.05, -0.05, 0.0]), h["kR"], p.huber, grad=p.grad)
    c_gt, c_off = _oracle_blocks(orc, at_gt)[0, 0], _oracle_blocks(orc, off)[0, 0]
    assert c_gt < 0.5 * c_off, (c_gt, c_off)


@pytest.mark.parametrize("collective", ["allgather", "allreduce"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_pair_shards_of_every_rank_on_one_gpu(orc, mbavo, gpu_ctx, world, collective):
    """bench.py --workload c4_batch512 --gpus N (pair sharding): rank r evaluates pairs b % N == r into ITS slice of the
    result buffer (all-gather, the default: equal slices, B = 21 does not divide by N, so the fuller ranks' slices set the
    width) or of the zero send buffer (all-reduce).  The N ranks run one after the other on this GPU; the sum of their
    buffers (what either collective leaves on every rank: the slices are disjoint, x + 0 + ... + 0 is exact) must hold
    every pair's block -- bit-identical to the rank's own evaluation, within 1e-12 of the whole batch evaluated at once
    (another tile partition, so another summation grouping), and 1e-9 from the oracle for sampled pairs."""
    import torch
    B = 21
    batch = wl.RenderedPairBatch(gpu_ctx, B, H=240, W=320, S=8, k=4, seed=6)
    total, ref, mine = None, None, {}
    for r in range(world):
        se = shard.ShardedEvaluation(gpu_ctx, batch.array, 4, r, world, "pairs", "cuda:0", pair_collective=collective)
        R = se.rows
        assert se.n_live == len(shard.pairs_of_rank(B, r, world)) and se.count == R * batch.E
        assert R == (B if collective == "allreduce" else world * -(-B // world))
        se.step(True, reduce=False)
        torch.cuda.synchronize()
        part = se.send.clone()
        rows = part.view(R, batch.E)
        lo, hi = se.row_base[r], se.row_base[r + 1]
        assert float(rows[:lo].abs().max() if lo else 0.0) == 0.0 and float(rows[hi:].abs().max() if hi < R else 0.0) == 0.0
        for b in shard.pairs_of_rank(B, r, world):
            mine[b] = rows[se.row_of_pair[b]].clone()
        total = part if total is None else total + part
        if r == world - 1:
            ref = se.reference()
            layout = se
    got = total.view(layout.rows, batch.E)
    for b in range(B):
        assert torch.equal(got[layout.row_of_pair[b]], mine[b])                      # the sum is exact
    assert float((total - ref).abs().max() / ref.abs().max()) <= 1e-12                # vs the whole batch at once
    for b in (0, 7, 20):
        o = _oracle_blocks(orc, batch.host_problem(b))[0]
        g = got[layout.row_of_pair[b]].cpu().numpy()
        assert np.abs(g - o).max() <= 1e-9 * np.abs(o).max()


def test_empty_slots_are_finalized(orc, mbavo, gpu_ctx):
    """a (problem, frame) slot without tiles (K == 0: a keypoint shard of K < world, a pyramid level without
    a surviving keypoint) has no workgroup to finalize it in the single-launch form.  mbavo_eval on K == 0 must return an
    all-zero system (not the previous evaluation's pinned blocks), a mixed batch must write the empty problem's zero
    block and valid count, and the evaluation must complete without the completion-word timeout."""
    import time
    import torch
    sc = scenes.Scene(S=8, F=2, k=4, P=8, K=40, seed=3)
    d = scenes.DeviceScene(sc)
    full = scenes.gpu_eval(gpu_ctx, d)               # leaves non-zero blocks in the context's pinned staging
    assert full["cost"] > 0
    n = 6 * sc.N
    for with_h in (True, False):
        p = d.problem()
        p.K = 0
        cost, H, g = np.full(1, 7.0), np.full(n * n, 7.0), np.full(n, 7.0)
        t0 = time.perf_counter()
        rc = gpu_ctx.lib.mbavo_eval(gpu_ctx.handle, C.byref(p), sc.k, mbavo.capi.dp(cost), mbavo.capi.dp(H) if with_h else None,
                                    mbavo.capi.dp(g) if with_h else None, None)
        assert rc == 0 and time.perf_counter() - t0 < 1.0
        assert cost[0] == 0.0
        if with_h:
            assert not H.any() and not g.any()
    # mixed batch: [empty shard, the full problem, empty shard]
    E = sc.E
    arr = (mbavo.capi.Problem * 3)(d.problem(), d.problem(), d.problem())
    arr[0].K = 0
    arr[2].K = 0
    fb = torch.full((6 * E,), 5.0, dtype=torch.float64, device="cuda:0")
    valid = torch.full((6,), 5.0, dtype=torch.float64, device="cuda:0")
    assert gpu_ctx.lib.mbavo_eval_batch(gpu_ctx.handle, 3, arr, sc.k, 1, fb.data_ptr(), None, valid.data_ptr()) == 0
    torch.cuda.synchronize()
    fbh, vh = fb.cpu().numpy().reshape(6, E), valid.cpu().numpy()
    assert not fbh[[0, 1, 4, 5]].any() and not vh[[0, 1, 4, 5]].any()
    one, _, v1 = scenes.gpu_eval_batch(gpu_ctx, [d], sc.k)
    assert np.abs(fbh[2:4] - one).max() <= 1e-12 * np.abs(one).max() and np.array_equal(vh[2:4], v1)
    # keypoint shards of a problem with fewer keypoints than ranks (K = 3, world = 8): five shards are empty
    sc3 = scenes.Scene(S=8, F=1, k=4, P=8, K=3, seed=9)
    d3 = scenes.DeviceScene(sc3)
    whole = (mbavo.capi.Problem * 1)(d3.problem())
    ref, _, _ = scenes.gpu_eval_batch(gpu_ctx, [d3], 4)
    acc = np.zeros_like(ref)
    for r in range(8):
        sh, first = shard.shard_array(gpu_ctx.lib, whole, r, 8, "keypoints")
        out = torch.full((E,), 3.0, dtype=torch.float64, device="cuda:0")
        assert gpu_ctx.lib.mbavo_eval_batch(gpu_ctx.handle, 1, sh, 4, 1, out.data_ptr(), None, None) == 0
        torch.cuda.synchronize()
        acc += out.cpu().numpy()
    assert np.abs(acc - ref).max() <= 1e-12 * np.abs(ref).max()


def test_tracker_level_without_keypoints(orc, mbavo, gpu_ctx):
    """The LM loop with a pyramid level that has no keypoints (K == 0 at the coarsest level): the level evaluates to zero
    cost, takes the reference's course through the loop (a zero step, no decrease, level ends) and the finer levels run
    as usual -- same trace kinds and final knots as the oracle."""
    import tracking
    sc = tracking.make_tracking_scene(orc, H=120, W=160, levels=3, S=8, k=2, F=1, seed=2)
    lv = sc["levels"][2]
    lv["kp_xy"], lv["kp_z"] = np.zeros((0, 2)), np.zeros(0)
    want = tracking.run_oracle_tracker(orc, sc)
    # (device tensors of zero elements have null data pointers: give the empty level one unused keypoint's storage)
    lv["kp_xy"], lv["kp_z"] = np.zeros((1, 2)), np.ones(1)
    got = _run_gpu_tracker_with_empty_level(mbavo, gpu_ctx, sc, empty_level=2)
    assert [t[:3] for t in got["trace"]] == [t[:3] for t in want["trace"]]
    assert np.abs(got["kt"] - want["kt"]).max() < 1e-6 and np.abs(got["kR"] - want["kR"]).max() < 1e-6


def _run_gpu_tracker_with_empty_level(mbavo, ctx, sc, empty_level):
    import torch
    import tracking
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
        q.H, q.W, q.K, q.P, q.S = lv["H"], lv["W"], (0 if i == empty_level else lv["kp_xy"].shape[0]), lv["pattern"].size // 2, lv["S"]
        q.d_ref_img, q.d_ref_dIxy, q.d_cur_imgs = ref.data_ptr(), grad.data_ptr(), ptrs.data_ptr()
        q.d_kp_xy, q.d_kp_z, q.d_pattern = xy.data_ptr(), z.data_ptr(), pat.data_ptr()
    torch.cuda.synchronize()
    opts = tracking.OPTS
    o = cap_mod.TrackOpts()
    o.num_levels, o.spline_deg_k = nl, sc["k"]
    o.max_num_iterations, o.max_consecutive_nonmonotonic_steps, o.solver_type = opts["max_num_iterations"], opts["max_nonmono"], opts["solver_type"]
    for i in range(4):
        o.intrinsics[i] = float(sc["intr"][i])
    o.huber_k, o.min_step_quality = opts["huber_k"], opts["min_step_quality"]
    o.min_abs_cost_decrease, o.max_chi_square_error = opts["min_abs_cost_decrease"], opts["max_chi_square_error"]
    kt, kR = sc["kt0"].ravel().copy(), sc["kR0"].ravel().copy()
    start, cost = np.zeros(F, np.int32), np.zeros(1)
    trace = (cap_mod.TraceRec * 256)()
    n = ctx.lib.mbavo_optimize_trajectory(ctx.handle, C.byref(o), levels, F, cap_mod.dp(sc["cap"]), cap_mod.dp(sc["exp"]),
                                          sc["t0"], sc["dt"], cap_mod.dp(kt), cap_mod.dp(kR), sc["N"], cap_mod.ip(start),
                                          cap_mod.dp(cost), trace, 256)
    assert 0 <= n <= 256, n
    return dict(kt=kt.reshape(-1, 3), kR=kR.reshape(-1, 4),
                trace=[(r.level, r.iter, r.kind, r.num_outliers, r.radius, r.eval_cost, r.candidate_cost, r.model_change, r.quality)
                       for r in trace[:n]])
