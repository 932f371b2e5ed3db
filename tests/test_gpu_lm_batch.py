"""Device-side batched LM against the host-driven loop (mbavo_optimize_trajectory, one level,
one problem at a time) and the oracle: the accept / reject / invalid sequence, iteration and outlier counts are
exact; costs, radii and knots agree to rounding amplified by the conditioning of the normal equations (the wave
reduces dot products as a butterfly, the host sums sequentially; on these plane scenes cond(H) ~ 1e8-1e10).
Stated tolerances: costs 1e-5 relative, radius 1e-4 relative, knots 1e-4 absolute (the same as the host-loop
tracker against the oracle, tests/test_gpu_tracker.py)."""
import ctypes as C

import numpy as np
import pytest

from mba_vo_amd import synth, workloads

pytestmark = pytest.mark.gpu

OPTS = dict(max_it=25, max_nonmono=5, min_q=0.5, min_dec=1e-3, chi=3.0)


def _scene(B, k, N, F, seed, H=96, W=128, S=4):
    """B independent pairs with their own splines; F frames per pair at different knot segments."""
    rng = np.random.default_rng(seed)
    probs = []
    ref0 = synth.texture_image(H, W, seed=seed, octaves=(32, 16, 8, 4))
    grad0 = synth.image_gradients(ref0)
    xy, z = synth.semi_dense_keypoints(ref0, cell=10, thresh=2.0, margin=8, seed=seed + 1)
    intr = np.array([W / 2.0, W / 2.0, W / 2.0, H / 2.0])
    t0, dt = 0.0, 0.5
    cap = [0.25 + 0.5 * f for f in range(F)]
    exp = [0.1] * F
    assert int((cap[-1] + exp[-1]) / dt) + k <= N
    for b in range(B):
        kt, kR = synth.harness_spline(0.004, 0.05, N)
        kt = kt + rng.normal(0, 2e-3, kt.shape)
        curs = [workloads._current_image(ref0, rng, shift=(1 + (b + f) % 2, -(1 + b % 2)), noise=3) for f in range(F)]
        probs.append(workloads.Prob(ref0, curs, xy, z, synth.PATTERN8, intr, S, k, N, cap, exp, t0, dt, kt, kR, 10.0, grad=grad0))
    return probs


def _host_lm(mbavo, ctx, dw, b, p, solver, trace_cap=64, fast_solve_ratio=0.0):
    capi = mbavo.capi
    lv = (capi.Level * 1)()
    q, a = lv[0], dw.array[b]
    q.H, q.W, q.K, q.P, q.S = p.H, p.W, p.K, p.P, p.S
    q.d_ref_img, q.d_ref_dIxy, q.d_cur_imgs = a.d_ref_img, a.d_ref_dIxy, a.d_cur_imgs
    q.d_kp_xy, q.d_kp_z, q.d_pattern = a.d_kp_xy, a.d_kp_z, a.d_pattern
    o = capi.TrackOpts()
    o.num_levels, o.spline_deg_k, o.max_num_iterations = 1, p.k, OPTS["max_it"]
    o.max_consecutive_nonmonotonic_steps, o.solver_type = OPTS["max_nonmono"], solver
    for i in range(4):
        o.intrinsics[i] = float(p.intr[i])
    o.huber_k, o.min_step_quality = p.huber, OPTS["min_q"]
    o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_dec"], OPTS["chi"]
    o.fast_solve_ratio = fast_solve_ratio  # (0: the default stand-in up to a pivot ratio of 1e8; -1: the reference's solvers for every system)
    kt, kR = p.knots_t.copy(), p.knots_R.copy()
    start, cost = np.zeros(p.F, np.int32), np.zeros(1)
    trace = (capi.TraceRec * trace_cap)()
    n = ctx.lib.mbavo_optimize_trajectory(ctx.handle, C.byref(o), lv, p.F, capi.dp(p.cap), capi.dp(p.exp), p.t0, p.dt,
                                          capi.dp(kt), capi.dp(kR), p.N, capi.ip(start), capi.dp(cost), trace, trace_cap)
    assert 0 < n <= trace_cap, n
    return kt, kR, float(cost[0]), [(r.iter, r.kind, r.num_outliers, r.radius, r.eval_cost, r.candidate_cost, r.model_change, r.quality)
                                     for r in trace[:n]]


@pytest.mark.parametrize("k,N,F,solver,fast", [(4, 4, 1, 0, "0"), (4, 6, 2, 0, "0"), (2, 3, 2, 1, None), (4, 4, 1, 1, None), (2, 2, 1, 0, "0"),
                                                   (4, 16, 2, 0, None), (4, 8, 2, 0, None),
                                                   (4, 4, 1, 0, None), (2, 2, 1, 0, None), (2, 3, 2, 0, None), (2, 3, 2, 1, "0"), (4, 4, 1, 1, "0"),
                                                   (4, 6, 3, 0, None), (2, 5, 4, 0, None), (4, 6, 3, 1, None)])  # n = 36, 30 with data on every knot: the workgroup LDL^T
def test_lm_batch_matches_host_loop(mbavo, gpu_ctx, k, N, F, solver, fast):
    """fast = "0" (fast_solve_ratio = -1 in both option structs): solver type 0 through the Jacobi solvers only -- n = 24, 36 (three blocks per thread), 12:
    eigenvalue Jacobi; solver type 1 through the pivoted LDL^T only.  Default: LDL^T in registers, refined in double-double above a
    pivot ratio of 1e8, stands in for either (n = 12, 18, 24); n = 96: one-wave sweeps, 48: the eigenvalue Jacobi's largest."""
    import torch
    ratio = -1.0 if fast == "0" else 0.0
    capi = mbavo.capi
    B = 6
    probs = _scene(B, k, N, F, seed=11 + k + N)
    dw = workloads.DeviceWorkload(probs)
    host = [_host_lm(mbavo, gpu_ctx, dw, b, p, solver, fast_solve_ratio=ratio) for b, p in enumerate(probs)]  # does not touch the device knots
    o = capi.LmBatchOpts()
    o.fast_solve_ratio = ratio
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, OPTS["max_it"], OPTS["max_nonmono"]
    o.solver_type, o.sync_every = solver, 3
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
    cap = 64
    res = (capi.LmBatchResult * B)()
    trace = (capi.TraceRec * (B * cap))()
    rc = gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, dw.array, C.byref(o), res, trace, cap)
    assert rc == 0, rc
    torch.cuda.synchronize()
    kinds_seen = set()
    for b, p in enumerate(probs):
        kt_h, kR_h, cost_h, tr_h = host[b]
        r = res[b]
        tr_d = [(t.iter, t.kind, t.num_outliers, t.radius, t.eval_cost, t.candidate_cost, t.model_change, t.quality)
                for t in trace[b * cap:b * cap + r.num_trace]]
        assert [(t[0], t[1], t[2]) for t in tr_d] == [(t[0], t[1], t[2]) for t in tr_h], (b, tr_d, tr_h)
        kinds_seen |= {t[1] for t in tr_d}
        for td, th in zip(tr_d, tr_h):
            assert abs(td[3] - th[3]) <= 1e-4 * th[3]                      # radius
            for i in (4, 5, 6):                                            # costs, model change
                assert abs(td[i] - th[i]) <= 1e-5 * max(1.0, abs(th[i])), (b, i, td, th)
            assert abs(td[7] - th[7]) <= 1e-3 * max(1.0, abs(th[7]))       # quality (ratio of small differences)
        assert r.accepted == sum(1 for t in tr_h if t[1] == 1) and r.rejected == sum(1 for t in tr_h if t[1] == 2)
        assert r.invalid == sum(1 for t in tr_h if t[1] == 3)
        assert abs(r.final_cost - cost_h) <= 1e-5 * max(1.0, cost_h) and abs(r.initial_cost - tr_h[0][4]) <= 1e-12 * tr_h[0][4]
        kt_d = dw.keep_knots(b)[0].cpu().numpy()
        kR_d = dw.keep_knots(b)[1].cpu().numpy()
        assert np.abs(kt_d - kt_h).max() < 1e-4 and np.abs(kR_d - kR_h).max() < 1e-4
        if r.accepted > 0:
            assert np.abs(kt_d - p.knots_t).max() > 1e-9  # the knots really moved
    assert 1 in kinds_seen and (2 in kinds_seen or 3 in kinds_seen)


def test_lm_batch_eigenvalue_jacobi_matches_one_wave_sweeps(mbavo, gpu_ctx):
    """Solver type 0 through the workgroup-parallel eigenvalue Jacobi (default for n <= 48) and through the one-wave one-sided
    sweeps (mbavo_lm_batch_opts.eig = -1): the same records, costs to 1e-6 relative (both solve to rounding x cond(H), cond ~1e9)."""
    import torch
    capi = mbavo.capi
    B, k, N, F = 8, 4, 4, 1
    out = {}
    for eig in ("1", "0", "fast"):
        # "fast": the default -- LDL^T in registers where the pivot ratio allows; the other two: Jacobi solvers only
        probs = _scene(B, k, N, F, seed=23)
        dw = workloads.DeviceWorkload(probs)
        o = capi.LmBatchOpts()
        o.eig = -1 if eig == "0" else 1
        o.fast_solve_ratio = 0.0 if eig == "fast" else -1.0
        o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, OPTS["max_it"], OPTS["max_nonmono"]
        o.solver_type, o.sync_every = 0, 0  # (0: the event-lagged done check)
        o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
        cap = 64
        res = (capi.LmBatchResult * B)()
        trace = (capi.TraceRec * (B * cap))()
        assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, dw.array, C.byref(o), res, trace, cap) == 0
        torch.cuda.synchronize()
        out[eig] = [[(t.iter, t.kind, t.num_outliers, t.eval_cost, t.candidate_cost) for t in trace[b * cap:b * cap + res[b].num_trace]]
                    for b in range(B)]
    for other in ("0", "fast"):
        for a, b in zip(out["1"], out[other]):
            assert [t[:3] for t in a] == [t[:3] for t in b]
            for ta, tb in zip(a, b):
                assert abs(ta[3] - tb[3]) <= 1e-6 * max(1.0, abs(tb[3])) and abs(ta[4] - tb[4]) <= 1e-6 * max(1.0, abs(tb[4]))


@pytest.mark.parametrize("k,N,F,sync_every", [(4, 6, 2, 0), (2, 3, 1, 0), (4, 4, 1, 3)])
def test_lm_batch_deferred_finalize_same_bits(mbavo, gpu_ctx, k, N, F, sync_every):
    """Lists of >= 64 (problem, frame) slots: the LM kernels sum the tile partials themselves (no finalize launch; engine.h:
    set_defer_finalize) in the finalize kernel's order -- every trace record and every final knot is bit-identical to the run with
    the finalize kernels (mbavo_lm_batch_opts.defer_finalize = -1); the done word in pinned host memory (sync_every 0) against the stream-drain scheme.
    (retile = -1: the second, finer tiling of the late slots -- which only exists with deferred sums -- would change the grouping
    of the sums; test_lm_batch_retiled_late_slots covers it.)"""
    import torch
    capi = mbavo.capi
    B = 70
    out = {}
    for defer in ("1", "0"):
        probs = _scene(B, k, N, F, seed=31)
        dw = workloads.DeviceWorkload(probs)
        dw.array[3].K = 0  # a pair without keypoints: its slots have no tile (all-zero sums either way)
        o = capi.LmBatchOpts()
        o.retile, o.defer_finalize = -1, (1 if defer == "1" else -1)
        o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, 12, OPTS["max_nonmono"]
        o.solver_type, o.sync_every = 0, sync_every
        o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
        cap = 32
        res = (capi.LmBatchResult * B)()
        trace = (capi.TraceRec * (B * cap))()
        assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, dw.array, C.byref(o), res, trace, cap) == 0
        torch.cuda.synchronize()
        recs = [[(t.iter, t.kind, t.num_outliers, t.radius, t.eval_cost, t.candidate_cost, t.model_change, t.quality)
                 for t in trace[b * cap:b * cap + res[b].num_trace]] for b in range(B)]
        knots = [tuple(x.cpu().numpy().tobytes() for x in dw.keep_knots(b)) for b in range(B)]
        out[defer] = (recs, knots, [(r.iterations, r.accepted, r.rejected, r.invalid, r.final_cost) for r in res])
    assert repr(out["1"]) == repr(out["0"])  # (repr: the pair without keypoints carries NaN step qualities in both runs; 17 digits = the bits)
    assert sum(r[1] for r in out["1"][2]) > B // 2 and sum(r[2] + r[3] for r in out["1"][2]) > 0  # steps were taken and refused


def test_lm_batch_packed_keyframes_same_records(mbavo, gpu_ctx):
    """The batched LM on packed keyframes (mbavo_problem.grad_fp16 = 2) against the float-gradient form of the same pairs: the
    same records, costs to 1e-5 relative (the stated tolerance of the batched LM against the host loop: rounding x cond(H))."""
    import torch
    capi = mbavo.capi
    B, k, N, F = 6, 4, 4, 1
    out = {}
    for fmt in (0, 2):
        probs = _scene(B, k, N, F, seed=31)
        for p in probs:
            p.grad_fp16 = fmt
        dw = workloads.DeviceWorkload(probs)
        o = capi.LmBatchOpts()
        o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, OPTS["max_it"], OPTS["max_nonmono"]
        o.solver_type, o.sync_every = 0, 4
        o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
        cap = 64
        res = (capi.LmBatchResult * B)()
        trace = (capi.TraceRec * (B * cap))()
        assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, dw.array, C.byref(o), res, trace, cap) == 0
        torch.cuda.synchronize()
        out[fmt] = [[(t.iter, t.kind, t.num_outliers, t.eval_cost, t.candidate_cost) for t in trace[b * cap:b * cap + res[b].num_trace]]
                    for b in range(B)]
    for a, b in zip(out[2], out[0]):
        assert [t[:3] for t in a] == [t[:3] for t in b]
        for ta, tb in zip(a, b):
            assert abs(ta[3] - tb[3]) <= 1e-5 * max(1.0, abs(tb[3])) and abs(ta[4] - tb[4]) <= 1e-5 * max(1.0, abs(tb[4]))


def test_lm_batch_against_oracle(orc, mbavo, gpu_ctx):
    """The same loop on the CPU oracle (orc_optimize_trajectory, one level) for a few problems."""
    import torch
    import tracking
    capi = mbavo.capi
    k, N, F, B = 4, 4, 1, 3
    probs = _scene(B, k, N, F, seed=5)
    dw = workloads.DeviceWorkload(probs)
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, OPTS["max_it"], OPTS["max_nonmono"]
    o.solver_type, o.sync_every = 0, 4
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
    res = (capi.LmBatchResult * B)()
    cap = 64
    trace = (capi.TraceRec * (B * cap))()
    assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, dw.array, C.byref(o), res, trace, cap) == 0
    torch.cuda.synchronize()
    for b, p in enumerate(probs):
        sc = dict(levels=[dict(H=p.H, W=p.W, ref=p.ref, grad=p.grad, cur=p.cur, kp_xy=p.kp_xy, kp_z=p.kp_z, pattern=p.pattern, S=p.S)],
                  k=k, N=N, F=F, cap=p.cap, exp=p.exp, t0=p.t0, dt=p.dt, intr=p.intr, kt0=p.knots_t.reshape(-1, 3), kR0=p.knots_R.reshape(-1, 4))
        want = tracking.run_oracle_tracker(orc, sc, dict(max_num_iterations=OPTS["max_it"], max_nonmono=OPTS["max_nonmono"], solver_type=0,
                                                         huber_k=p.huber, min_step_quality=OPTS["min_q"],
                                                         min_abs_cost_decrease=OPTS["min_dec"], max_chi_square_error=OPTS["chi"]))
        got = [(t.iter, t.kind, t.num_outliers) for t in trace[b * cap:b * cap + res[b].num_trace]]
        assert got == [(t[1], t[2], t[3]) for t in want["trace"]]
        assert abs(res[b].final_cost - want["cost"]) <= 1e-5 * max(1.0, want["cost"])
        assert np.abs(dw.keep_knots(b)[0].cpu().numpy() - want["kt"].ravel()).max() < 1e-4


def test_lm_batch_argument_errors(mbavo, gpu_ctx):
    capi = mbavo.capi
    probs = _scene(1, 4, 4, 1, seed=2)
    dw = workloads.DeviceWorkload(probs)
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.solver_type = 3, 5, 0
    assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, 1, dw.array, C.byref(o), None, None, 0) == -1
    o.spline_deg_k, o.solver_type = 4, 7
    assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, 1, dw.array, C.byref(o), None, None, 0) == -1


def test_sharded_lm_batch_ranks_add_up_on_one_gpu(mbavo, gpu_ctx):
    """shard.ShardedLmBatch (pair b on rank b % world, the LM loop per rank, ONE all-gather of the records): the ranks of
    world = 1, 2, 4 run one after the other on this GPU (no collective: every rank's slice is checked where it lands), every
    pair's record against the whole batch's -- same accept / reject counts, knots to 1e-9 (a pair's result does not depend on
    what else is in the batch; the tile layout of a smaller list may choose another kernel form: rounding, not bits)."""
    import torch
    from mba_vo_amd import shard
    capi = mbavo.capi
    B, k, N, F = 10, 4, 4, 1
    probs = _scene(B, k, N, F, seed=41)
    dw = workloads.DeviceWorkload(probs)
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, 12, OPTS["max_nonmono"]
    o.solver_type, o.sync_every = 0, 0
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
    init = [(p.knots_t.reshape(N, 3), p.knots_R.reshape(N, 4)) for p in probs]
    ref = None
    for world in (1, 2, 4):
        recs = {}
        for rank in range(world):
            sl = shard.ShardedLmBatch(gpu_ctx, dw.array, k, rank, world, "cuda:0", o, init)
            assert sl.run(gather=False) == 0
            assert sl.run(gather=False) == 0  # a second run starts from the initial knots again
            for b in shard.pairs_of_rank(B, rank, world):
                recs[b] = sl.record(b)
            others = [b for b in range(B) if b % world != rank]
            assert all(sl.record(b, N)["iterations"] == 0 and not sl.record(b, N)["knots_t"].any() for b in others[:2])
        assert sorted(recs) == list(range(B))
        if ref is None:
            ref = recs
            assert sum(r["accepted"] for r in ref.values()) >= B // 2 and all(r["final_cost"] <= r["initial_cost"] for r in ref.values())
            assert any(r["final_cost"] < 0.9 * r["initial_cost"] for r in ref.values())
            continue
        for b in range(B):
            a, r = recs[b], ref[b]
            assert (a["iterations"], a["accepted"], a["rejected"], a["invalid"], a["num_outliers"]) == \
                   (r["iterations"], r["accepted"], r["rejected"], r["invalid"], r["num_outliers"]), (world, b, a, r)
            assert np.abs(a["knots_t"] - r["knots_t"]).max() < 1e-9 and np.abs(a["knots_R"] - r["knots_R"]).max() < 1e-9
            assert abs(a["final_cost"] - r["final_cost"]) <= 1e-9 * r["final_cost"]
    # the caller's own knot buffers were never touched: the records hold the aligned splines
    assert np.array_equal(dw.keep_knots(0)[0].cpu().numpy(), probs[0].knots_t)


# ---------------------------------------------------------------------------------------------------------------------------
# f4 against the ORACLE on configs[2]'s own data: pairs of the rendered 640x480 sequence, every
# solver form of the batched LM meeting the oracle's loop directly, at the bar the host-driven loop is held to
# (tests/test_gpu_tracker.py): identical accept / reject / invalid sequence and outlier counts, pose at capture time 1e-5,
# |delta ATE| 1e-5.  Reference: blur_aware_direct_tracker.cpp:590-924, solve_normal_equation.h:20-26.
_RENDERED = {}
_ORACLE_RUNS = {}


def _rendered(gpu_ctx, k):
    if k not in _RENDERED:
        _RENDERED[k] = workloads.RenderedPairBatch(gpu_ctx, 8, H=480, W=640, S=8, k=k, seed=1)  # the first 8 pairs of configs[2]
    return _RENDERED[k]


def _oracle_alignment(orc, batch, b, k, N, solver):
    """The oracle's optimizePyramidLevel on pair b of the rendered batch with its first N control knots."""
    import tracking
    key = (k, N, solver, b)
    if key not in _ORACLE_RUNS:
        p = batch.host_problem(b)
        sc = dict(levels=[dict(H=p.H, W=p.W, ref=p.ref, grad=p.grad, cur=p.cur, kp_xy=p.kp_xy, kp_z=p.kp_z, pattern=p.pattern, S=p.S)],
                  k=k, N=N, F=1, cap=p.cap, exp=p.exp, t0=p.t0, dt=p.dt, intr=p.intr,
                  kt0=np.ascontiguousarray(p.knots_t.reshape(-1, 3)[:N]), kR0=np.ascontiguousarray(p.knots_R.reshape(-1, 4)[:N]))
        _ORACLE_RUNS[key] = (sc, tracking.run_oracle_tracker(orc, sc, dict(max_num_iterations=OPTS["max_it"], max_nonmono=OPTS["max_nonmono"],
                                                                         solver_type=solver, huber_k=p.huber, min_step_quality=OPTS["min_q"],
                                                                         min_abs_cost_decrease=OPTS["min_dec"], max_chi_square_error=OPTS["chi"])))
    return _ORACLE_RUNS[key]


@pytest.mark.parametrize("k,N,solver,env", [
    (4, 4, 0, {}),                                                     # refined LDL^T stand-in in registers (default)
    (4, 4, 0, {"refined_ratio": -1.0}),                               # plain LDL^T up to a pivot ratio of 1e8, the Jacobi solver above
    (4, 4, 0, {"fast_solve_ratio": -1.0}),                              # workgroup-parallel eigenvalue Jacobi for every system
    (4, 4, 0, {"fast_solve_ratio": -1.0, "eig": -1}),         # one-wave one-sided Jacobi SVD: solve_normal_equation.h case 0
    (4, 4, 1, {}),                                                     # solver type 1 through the refined stand-in
    (4, 4, 1, {"fast_solve_ratio": -1.0}),                              # pivoted LDL^T: solve_normal_equation.h case 1 (wave 0 of the wide workgroup)
    (4, 4, 1, {"fast_solve_ratio": -1.0, "eig": -1}),         # the same in the one-wave workgroup
    (2, 2, 0, {}), (2, 2, 0, {"fast_solve_ratio": -1.0}), (2, 2, 1, {}),  # the reference's default degree, both solver types
    (2, 4, 0, {}),                                                     # RANK-DEFICIENT: knots 2 and 3 have no data -> pseudo-inverse
    (2, 4, 0, {"eig": -1}),
])
def test_lm_batch_rendered_pairs_against_oracle(orc, mbavo, gpu_ctx, k, N, solver, env):
    """`env`: the solver-form fields of mbavo_lm_batch_opts (they were environment variables until round 5)."""
    import torch
    import tracking
    capi = mbavo.capi
    batch = _rendered(gpu_ctx, k)
    B = batch.B
    batch.reset_knots()
    arr = (capi.Problem * B)()
    for b in range(B):
        C.memmove(C.byref(arr[b]), C.byref(batch.array[b]), C.sizeof(capi.Problem))
        arr[b].N = N  # the first N of the pair's four knots (the exposure lies in segment 0)
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, OPTS["max_it"], OPTS["max_nonmono"]
    o.solver_type, o.sync_every = solver, 0
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
    for name, v in env.items():
        setattr(o, name, v)
    cap = 64
    res = (capi.LmBatchResult * B)()
    trace = (capi.TraceRec * (B * cap))()
    assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, arr, C.byref(o), res, trace, cap) == 0
    torch.cuda.synchronize()
    e_gpu, e_orc, accepted = [], [], 0
    for b in range(B):
        sc, want = _oracle_alignment(orc, batch, b, k, N, solver)
        got = [(t.iter, t.kind, t.num_outliers) for t in trace[b * cap:b * cap + res[b].num_trace]]
        assert got == [(t[1], t[2], t[3]) for t in want["trace"]], (b, got, want["trace"])
        accepted += res[b].accepted
        assert abs(res[b].final_cost - want["cost"]) <= 1e-6 * max(1.0, want["cost"])
        h = batch._host[b]
        kt = h["dkt"].cpu().numpy().reshape(-1, 3)[:N]
        kR = h["dkR"].cpu().numpy().reshape(-1, 4)[:N]
        if k == 2 and N == 4:  # minimum-norm step: the knots without data stay where they were, on both sides, bit for bit
            assert np.array_equal(kt[2:], sc["kt0"][2:]) and np.array_equal(kR[2:], sc["kR0"][2:])
            assert np.array_equal(want["kt"][2:], sc["kt0"][2:]) and np.array_equal(want["kR"][2:], sc["kR0"][2:])
        tc = float(sc["cap"][0])
        pg, qg = tracking.pose_at(orc, k, sc["t0"], sc["dt"], kt, kR, tc)
        po, qo = tracking.pose_at(orc, k, sc["t0"], sc["dt"], want["kt"], want["kR"], tc)
        assert np.abs(pg - po).max() <= 1e-5 and np.abs(qg - qo).max() <= 1e-5, (b, pg - po, qg - qo)
        p_gt, _ = tracking.pose_at(orc, 4, sc["t0"], sc["dt"], h["kt_gt"], h["kR"], tc)  # the rendered (cubic) ground truth
        e_gpu.append(np.sum((pg - p_gt) ** 2))
        e_orc.append(np.sum((po - p_gt) ** 2))
    assert accepted >= B  # the loops really moved the knots
    ate_g, ate_o = float(np.sqrt(np.mean(e_gpu))), float(np.sqrt(np.mean(e_orc)))
    assert abs(ate_g - ate_o) <= 1e-5, (ate_g, ate_o)


def test_lm_batch_groups_same_records(mbavo, gpu_ctx):
    """mbavo_lm_batch on big batches runs as independent GROUPS (round 4: the second on its own engine, stream and host thread;
    default from 384 problems).  Forced here on 11 pairs (mbavo_lm_batch_opts.groups = 2 and 3, uneven shares) against one group: the same
    trace records per pair (kinds, iterations, outlier counts), costs 1e-9 relative (a group's list may be tiled differently),
    knots 1e-9; traces land at the pair's own rows."""
    import torch
    capi = mbavo.capi
    B, k, N, F = 11, 4, 4, 1
    out = {}
    for groups in ("1", "2", "3"):
        probs = _scene(B, k, N, F, seed=53)
        dw = workloads.DeviceWorkload(probs)
        o = capi.LmBatchOpts()
        o.groups = int(groups)
        o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, OPTS["max_it"], OPTS["max_nonmono"]
        o.solver_type, o.sync_every = 0, 0
        o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
        cap = 48
        res = (capi.LmBatchResult * B)()
        trace = (capi.TraceRec * (B * cap))()
        assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, dw.array, C.byref(o), res, trace, cap) == 0
        torch.cuda.synchronize()
        recs = [[(t.iter, t.kind, t.num_outliers, t.eval_cost, t.candidate_cost) for t in trace[b * cap:b * cap + res[b].num_trace]] for b in range(B)]
        knots = [np.concatenate([x.cpu().numpy().ravel() for x in dw.keep_knots(b)]) for b in range(B)]
        out[groups] = (recs, knots, [(r.iterations, r.accepted, r.rejected, r.invalid, r.num_outliers) for r in res], [r.final_cost for r in res])
        # without a trace buffer the results come back through pinned memory (no copy, no blocking synchronisation): the same numbers
        for kt, kR in [dw.keep_knots(b) for b in range(B)]:
            pass
    ref = out["1"]
    assert sum(r[1] for r in ref[2]) >= B // 2
    for g in ("2", "3"):
        recs, knots, counts, costs = out[g]
        assert counts == ref[2]
        for a, b in zip(recs, ref[0]):
            assert [t[:3] for t in a] == [t[:3] for t in b]
            for ta, tb in zip(a, b):
                assert abs(ta[3] - tb[3]) <= 1e-9 * max(1.0, abs(tb[3])) and abs(ta[4] - tb[4]) <= 1e-9 * max(1.0, abs(tb[4]))
        for a, b in zip(knots, ref[1]):
            assert np.abs(a - b).max() < 1e-9
        assert np.allclose(costs, ref[3], rtol=1e-9, atol=0)


def test_lm_batch_results_without_trace_come_from_pinned_memory(mbavo, gpu_ctx):
    """trace == NULL and sync_every <= 0: every problem's final state is stored to pinned host memory by the workgroup that ends it and
    the call polls the stream instead of copying back -- the results must equal those of the call WITH a trace buffer (which takes
    the copy path)."""
    import torch
    capi = mbavo.capi
    B, k, N, F = 9, 4, 4, 1
    got = []
    for with_trace in (True, False):
        probs = _scene(B, k, N, F, seed=61)
        dw = workloads.DeviceWorkload(probs)
        o = capi.LmBatchOpts()
        o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, OPTS["max_it"], OPTS["max_nonmono"]
        o.solver_type, o.sync_every = 0, 0
        o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
        res = (capi.LmBatchResult * B)()
        trace = (capi.TraceRec * (B * 32))() if with_trace else None
        for rep in range(2):  # (a second call re-arms the pinned words)
            for b, p in enumerate(probs):
                dw.keep_knots(b)[0].copy_(torch.from_numpy(p.knots_t))
                dw.keep_knots(b)[1].copy_(torch.from_numpy(p.knots_R))
            assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, dw.array, C.byref(o), res, trace, 32 if with_trace else 0) == 0
        got.append([(r.iterations, r.accepted, r.rejected, r.invalid, r.num_outliers, r.initial_cost, r.final_cost, r.radius) for r in res])
    assert got[0] == got[1] and sum(r[1] for r in got[0]) > 0


def test_lm_batch_retiled_late_slots(orc, mbavo, gpu_ctx):
    """Round 4: a batch of more pairs than its coarse tiling has tiles per CU (one or two tiles per pair) switches both passes of a slot
    to a second layout with four tiles per pair once few pairs are left (lm_batch.hip "RE-TILING"); the LM kernels are told per launch
    whose partials to sum.  160 rendered 480x640 pairs (two coarse tiles per pair), early exit so that the pairs finish at different
    slots: against mbavo_lm_batch_opts.retile = -1 the same iteration / accept / reject / invalid / outlier counts for every pair, final costs 1e-9,
    knots 1e-9 (another grouping of the sums, rounding only); three pairs against the oracle's loop (pose at capture 1e-5)."""
    import torch
    import tracking
    capi = mbavo.capi
    B = 160
    batch = workloads.RenderedPairBatch(gpu_ctx, B, H=480, W=640, S=8, k=4, seed=3)
    out = {}
    for mode in ("1", "0"):
        batch.reset_knots()
        o = capi.LmBatchOpts()
        o.retile = 1 if mode == "1" else -1
        o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = 4, OPTS["max_it"], OPTS["max_nonmono"]
        o.solver_type, o.sync_every = 0, 0
        o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
        res = (capi.LmBatchResult * B)()
        assert gpu_ctx.lib.mbavo_lm_batch(gpu_ctx.handle, B, batch.array, C.byref(o), res, None, 0) == 0
        torch.cuda.synchronize()
        knots = [np.concatenate([h["dkt"].cpu().numpy(), h["dkR"].cpu().numpy()]) for h in batch._host]
        out[mode] = ([(r.iterations, r.accepted, r.rejected, r.invalid, r.num_outliers) for r in res], [r.final_cost for r in res], knots)
    its = [c[0] for c in out["1"][0]]
    assert max(its) - min(its) >= 2 and sum(c[1] for c in out["1"][0]) >= B  # the pairs finish at different slots; steps were taken
    assert out["1"][0] == out["0"][0]
    assert np.allclose(out["1"][1], out["0"][1], rtol=1e-9, atol=0)
    for a, b in zip(out["1"][2], out["0"][2]):
        assert np.abs(a - b).max() < 1e-9
    for b in (0, 77, 159):
        p = batch.host_problem(b)
        sc = dict(levels=[dict(H=p.H, W=p.W, ref=p.ref, grad=p.grad, cur=p.cur, kp_xy=p.kp_xy, kp_z=p.kp_z, pattern=p.pattern, S=p.S)],
                  k=4, N=4, F=1, cap=p.cap, exp=p.exp, t0=p.t0, dt=p.dt, intr=p.intr,
                  kt0=np.ascontiguousarray(batch._host[b]["kt"]), kR0=np.ascontiguousarray(batch._host[b]["kR"]))
        want = tracking.run_oracle_tracker(orc, sc, dict(max_num_iterations=OPTS["max_it"], max_nonmono=OPTS["max_nonmono"], solver_type=0,
                                                         huber_k=p.huber, min_step_quality=OPTS["min_q"], min_abs_cost_decrease=OPTS["min_dec"],
                                                         max_chi_square_error=OPTS["chi"]))
        kn = out["1"][2][b]
        tc = float(sc["cap"][0])
        pg, qg = tracking.pose_at(orc, 4, sc["t0"], sc["dt"], kn[:12].reshape(4, 3), kn[12:].reshape(4, 4), tc)
        po, qo = tracking.pose_at(orc, 4, sc["t0"], sc["dt"], want["kt"], want["kR"], tc)
        assert np.abs(pg - po).max() <= 1e-5 and np.abs(qg - qo).max() <= 1e-5


# ---------------------------------------------------------------------------------------------------------------------------
# The host machinery around the groups under FAILURE: one group fails while the others run,
# destruction with parked / just-failed workers, decisions shared by the groups on mixed batches.
def _lm_opts(capi, k, groups=0, max_it=None):
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, OPTS["max_it"] if max_it is None else max_it, OPTS["max_nonmono"]
    o.solver_type, o.sync_every = 0, 0
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = OPTS["min_q"], OPTS["min_dec"], OPTS["chi"]
    o.groups = groups
    return o


def _run_lm(capi, ctx, dw, B, o, cap=48):
    import torch
    res = (capi.LmBatchResult * B)()
    trace = (capi.TraceRec * (B * cap))()
    rc = ctx.lib.mbavo_lm_batch(ctx.handle, B, dw.array, C.byref(o), res, trace, cap)
    torch.cuda.synchronize()
    recs = [[(t.iter, t.kind, t.num_outliers) for t in trace[b * cap:b * cap + res[b].num_trace]] for b in range(B)]
    return rc, recs, [r.final_cost for r in res]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("groups", [2, 4])
def test_lm_batch_one_group_fails_the_others_run(mbavo, groups):
    """A capture time outside the spline in ONE pair of the last group (the pose entries clamp the segment and count it: A2,
    MBAVO_E_RANGE) while the other groups run their loops on their own streams and host threads: the call returns the error
    with every stream drained, the next call on the context -- the same batch, repaired -- succeeds with the records of a fresh
    context, and the failed call's range count is not reported again.  Then mbavo_destroy right after a failed
    call, workers just finished."""
    import time
    import torch
    capi = mbavo.capi
    B, k, N, F = 12, 4, 4, 1
    ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    try:
        dw = workloads.DeviceWorkload(_scene(B, k, N, F, seed=71))
        bad = torch.tensor([1.0e3], dtype=torch.float64, device="cuda:0")  # far beyond the four knots
        dw.array[B - 2].d_cap_time = bad.data_ptr()
        rc, _, _ = _run_lm(capi, ctx, dw, B, _lm_opts(capi, k, groups))
        assert rc == -2, rc                                                      # MBAVO_E_RANGE, from the group that owns the pair
        assert torch.cuda.current_stream().query()                               # nothing of the call is still in flight
        dw2 = workloads.DeviceWorkload(_scene(B, k, N, F, seed=71))              # (the failed call moved some pairs' knots)
        rc2, recs2, costs2 = _run_lm(capi, ctx, dw2, B, _lm_opts(capi, k, groups))
        assert rc2 == 0, rc2
        fresh = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        try:
            dw3 = workloads.DeviceWorkload(_scene(B, k, N, F, seed=71))
            rc3, recs3, costs3 = _run_lm(capi, fresh, dw3, B, _lm_opts(capi, k, groups))
        finally:
            fresh.close()
        assert rc3 == 0 and recs2 == recs3 and costs2 == costs3
        # fail again and destroy at once
        dw2.array[B - 1].d_cap_time = bad.data_ptr()
        rc4, _, _ = _run_lm(capi, ctx, dw2, B, _lm_opts(capi, k, groups))
        assert rc4 == -2
    finally:
        t0 = time.perf_counter()
        ctx.close()
        assert time.perf_counter() - t0 < 5.0


@pytest.mark.timeout(300)
def test_destroy_with_parked_group_workers(mbavo):
    """mbavo_destroy right after a grouped call (helper threads still spinning) and after the helpers went to sleep on their
    condition variable (200 us after their last job): both return promptly; a context that never ran a grouped call too."""
    import time
    import torch
    capi = mbavo.capi
    for pause in (0.0, 0.05):
        ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        dw = workloads.DeviceWorkload(_scene(9, 2, 2, 1, seed=5))
        rc, recs, _ = _run_lm(capi, ctx, dw, 9, _lm_opts(capi, 2, 3))
        assert rc == 0 and all(len(r) > 1 for r in recs)
        time.sleep(pause)
        t0 = time.perf_counter()
        ctx.close()
        assert time.perf_counter() - t0 < 5.0
    ctx = capi.Context(0)
    ctx.close()


@pytest.mark.parametrize("case", ["mixed_N", "just_under_two_per_cu"])
def test_lm_batch_groups_on_mixed_batches(mbavo, gpu_ctx, case):
    """the groups take the batch's largest knot count (kernel form, strides) from the whole batch, so a batch that
    mixes N = 4 and N = 6 problems -- whose halves would otherwise pick different kernels -- and a batch just under two problems
    per CU (whose halves fall under the CU count and are tiled finer) give the single-group run's records: identical
    (iteration, kind, outliers), final costs to 1e-9 relative (another grouping of the tile sums)."""
    capi = mbavo.capi
    if case == "mixed_N":
        a, b = _scene(8, 4, 4, 1, seed=61), _scene(8, 4, 6, 2, seed=62)
        probs = a[:5] + b[:3] + a[5:] + b[3:]     # first half mostly N = 4, second half mostly N = 6
    else:
        probs = _scene(500, 4, 4, 1, seed=63)
    B = len(probs)
    out = {}
    for groups in (1, 2):
        dw = workloads.DeviceWorkload(probs)
        rc, recs, costs = _run_lm(capi, gpu_ctx, dw, B, _lm_opts(capi, 4, groups, max_it=8), cap=24)
        assert rc == 0
        out[groups] = (recs, costs)
    assert out[1][0] == out[2][0]
    assert np.allclose(out[1][1], out[2][1], rtol=1e-9, atol=0)
    assert sum(1 for r in out[1][0] for t in r if t[1] == 1) >= B // 2
