"""LM-iteration throughput on a batch of keyframe pairs (BASELINE configs[2]/[3]): device-side batched LM
(mbavo_lm_batch) against the host-driven loop run pair by pair (mbavo_optimize_trajectory, one level).
Usage: python tools/lm_bench.py [B] [max_iterations] [host_pairs]        (one JSON line; bench.py imports bench_line)"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def device_lm(M, ctx, batch, solver, iterations, reps=3, min_dec=0.0, N=None, **tail):
    """mbavo_lm_batch on a RenderedPairBatch: best of `reps` runs from the same initial knots (N: the first N of a pair's four knots)."""
    import torch
    capi = M.capi
    B = batch.B
    arr = batch.array
    if N is not None:
        arr = (capi.Problem * B)()
        for b in range(B):
            C.memmove(C.byref(arr[b]), C.byref(batch.array[b]), C.sizeof(capi.Problem))
            arr[b].N = N
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = batch.k, iterations, 5
    o.solver_type, o.sync_every = solver, int(os.environ.get("MBAVO_LM_SYNC_EVERY", "0"))
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = 0.5, min_dec, 3.0  # min_dec 0: run all iterations
    for name, v in tail.items():  # the ABI 3 tail of mbavo_lm_batch_opts (solver form, schedule)
        setattr(o, name, v)
    res = (capi.LmBatchResult * B)()
    times = []
    for _ in range(reps):
        batch.reset_knots()
        torch.cuda.synchronize()
        t = time.perf_counter()
        rc = ctx.lib.mbavo_lm_batch(ctx.handle, B, arr, C.byref(o), res, None, 0)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
        assert rc == 0, rc
    iters = sum(r.iterations for r in res)
    rounds = max(r.iterations for r in res)
    return {"ms_total": round(1e3 * min(times), 4), "lm_iterations": iters, "rounds": rounds,
            "us_per_round": round(1e6 * min(times) / max(rounds, 1), 2),
            "us_per_pair_iteration": round(1e6 * min(times) / max(iters, 1), 3),
            "accepted": sum(r.accepted for r in res), "rejected": sum(r.rejected for r in res),
            "final_cost_sum": float(sum(r.final_cost for r in res))}


def host_lm(M, ctx, batch, solver, iterations, pairs):
    """The host-driven LM loop (mbavo_optimize_trajectory, one level) over the first `pairs` pairs, one after the other."""
    import torch
    capi = M.capi
    lv = (capi.Level * 1)()
    t_host, it_host = 0.0, 0
    for b in range(pairs):
        a, h = batch.array[b], batch._host[b]
        q = lv[0]
        q.H, q.W, q.K, q.P, q.S = a.H, a.W, a.K, a.P, a.S
        q.d_ref_img, q.d_ref_dIxy, q.d_cur_imgs = a.d_ref_img, a.d_ref_dIxy, a.d_cur_imgs
        q.d_kp_xy, q.d_kp_z, q.d_pattern = a.d_kp_xy, a.d_kp_z, a.d_pattern
        to = capi.TrackOpts()
        to.num_levels, to.spline_deg_k, to.max_num_iterations, to.max_consecutive_nonmonotonic_steps, to.solver_type = 1, batch.k, iterations, 5, solver
        for i in range(4):
            to.intrinsics[i] = float(batch.intr[i])
        to.huber_k, to.min_step_quality, to.min_abs_cost_decrease, to.max_chi_square_error = h["huber"], 0.5, 0.0, 3.0
        kt, kR = h["kt"].ravel().copy(), h["kR"].ravel().copy()
        cap, exp = np.array([h["cap"]]), np.array([h["exp"]])
        start, cost = np.zeros(1, np.int32), np.zeros(1)
        trace = (capi.TraceRec * 64)()
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = ctx.lib.mbavo_optimize_trajectory(ctx.handle, C.byref(to), lv, 1, capi.dp(cap), capi.dp(exp), h["t0"], 0.5,
                                              capi.dp(kt), capi.dp(kR), 4, capi.ip(start), capi.dp(cost), trace, 64)
        t_host += time.perf_counter() - t
        assert n > 0, n
        it_host += max(r.iter for r in trace[:n])
    return {"pairs_timed": pairs, "ms_total": round(1e3 * t_host, 4), "lm_iterations": it_host,
            "us_per_pair_iteration": round(1e6 * t_host / max(it_host, 1), 3)}


def bench_line_k2(M, ctx, dev, B=64, iterations=10):
    """bench.py `configs.lm_batch64_k2`: the same pairs aligned with the reference's DEFAULT spline degree (k = 2 on the pair's
    first two knots, blur_aware_direct_tracker.h:50), both solver types."""
    from mba_vo_amd import workloads
    batch = workloads.RenderedPairBatch(ctx, B, S=8, k=2, device=dev, seed=1)
    out = {"workload": "%d pairs of the rendered blurred sequence (configs[2] data), k = 2 on N = 2 control knots (12 x 12 systems), device-side LM, "
                       "%d iterations per pair, no early exit" % (B, iterations), "B": B, "max_iterations": iterations}
    for solver, name in ((0, "svd"), (1, "ldlt")):
        out["device_" + name] = device_lm(M, ctx, batch, solver, iterations, N=2)
    return out


def bench_line(M, ctx, dev, B=64, iterations=10, host_pairs=8):
    """bench.py `configs.lm_batch64`: 64 pairs of the rendered sequence, 10 LM iterations each, both solvers."""
    from mba_vo_amd import workloads
    batch = workloads.RenderedPairBatch(ctx, B, S=8, k=4, device=dev, seed=1)
    out = {"workload": "%d pairs of the rendered blurred sequence (configs[2] data), device-side LM (mbavo_lm_batch), "
                       "%d iterations per pair, no early exit; us_per_round = wall time of a whole call / the largest iteration count, i.e. one "
                       "LM iteration slot of the whole batch (solve + pose entries, cost-only pass, decide, H/g pass); batches of 384+ pairs "
                       "run as two independent groups on their own streams" % (B, iterations),
           "B": B, "max_iterations": iterations, "K_mean": float(np.mean([p.K for p in batch.probs])), "P": 8, "S": 8}
    for solver, name in ((0, "svd"), (1, "ldlt")):
        out["device_" + name] = device_lm(M, ctx, batch, solver, iterations)
    if host_pairs > 0:
        out["host_svd"] = host_lm(M, ctx, batch, 0, iterations, host_pairs)
    # what the double-double refinement of the LDL^T stand-in costs: the plain stand-in admitted for every pivot ratio (NOT the default:
    # the refined solution is the centre of the cond * eps ball the reference's SVD lands in, the plain one a point of it)
    out["device_svd_plain_standin_not_default"] = device_lm(M, ctx, batch, 0, iterations, fast_solve_ratio=1e13, refined_ratio=-1.0)
    del batch
    # the same pairs with packed keyframes (mbavo_problem.grad_fp16 = 2: one word per pixel, identical tap values)
    packed = workloads.RenderedPairBatch(ctx, B, S=8, k=4, device=dev, seed=1, grad_fp16=2)
    out["device_svd_packed_keyframes"] = device_lm(M, ctx, packed, 0, iterations)
    return out


def sharded_line(M, ctx, dev, rank, world, barrier, max_over_ranks, sum_over_ranks, B=512, iterations=10, reps=3, packed=True, coll=None,
                 weak=False):
    """bench.py N > 1 `configs.lm_batch512_pairs`: B pairs of the rendered sequence aligned by the device-side LM, pair b on
    rank b % world (shard.ShardedLmBatch: no collective inside the loop, ONE all-gather of the records at the end).
    Every rank renders only its own pairs.  value = LM iterations of all pairs / max-over-ranks wall time of a call."""
    import torch
    from mba_vo_amd import shard, workloads
    capi = M.capi
    mine = shard.pairs_of_rank(B, rank, world)
    batch = workloads.RenderedPairBatch(ctx, B, S=8, k=4, device=dev, seed=1, pairs=mine, grad_fp16=2 if packed else 0)
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps, o.solver_type, o.sync_every = 4, iterations, 5, 0, 0
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = 0.5, 0.0, 3.0
    init = [None if h is None else (h["kt"], h["kR"]) for h in batch._host]
    sl = shard.ShardedLmBatch(ctx, batch.array, 4, rank, world, dev, o, init, collective=coll)
    best = None
    for _ in range(reps + 1):  # (the first call sizes the engine's arenas)
        barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        rc = sl.run()
        torch.cuda.synchronize()
        dt = max_over_ranks(time.perf_counter() - t)
        assert rc == 0, rc
        best = dt if best is None or _ == 1 else min(best, dt)
    iters = sum_over_ranks(float(sum(sl.res[j].iterations for j in range(len(mine)))))
    acc = sum_over_ranks(float(sum(sl.res[j].accepted for j in range(len(mine)))))
    # every rank holds every pair's record: the gathered iteration counts add up to the sum over the ranks
    got = sum(sl.record(b, 4)["iterations"] for b in range(0, B, max(1, B // 16)))
    want = sum_over_ranks(float(sum(sl.res[j].iterations for j, b in enumerate(mine) if b % max(1, B // 16) == 0)))
    return {"workload": "%d pairs of the rendered blurred sequence (%s keyframes), device-side LM, %d iterations per pair, pair b on "
                        "rank b %% N, one all-gather of the records" % (B, "packed" if packed else "float-gradient", iterations),
            "n_gpus": world, "sharding": "pairs (whole alignments)", "scaling": "weak" if weak else "strong", "pairs_per_rank": len(mine),
            "collective": "one all-gather of %d-double records: %s" % (sl.rec, getattr(sl.coll, "name", "rccl")), "ms_total": round(1e3 * best, 4),
            "lm_iterations": int(iters), "accepted": int(acc), "value": round(iters / best, 1), "unit": "pair LM iterations/s",
            "us_per_pair_iteration": round(1e6 * best / max(iters, 1.0), 4), "gather_check": bool(got == int(want))}


if __name__ == "__main__":
    import torch
    import mba_vo_amd as mbavo
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    HOSTN = int(sys.argv[3]) if len(sys.argv) > 3 else min(B, 16)
    ctx = mbavo.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    print(json.dumps(bench_line(mbavo, ctx, "cuda:0", B, IT, HOSTN)))
