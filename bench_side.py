"""bench_side.py -- bench.py's side legs, product code only (the checker legs are in bench_checks.py): the other BASELINE configs
at N = 1 (`single_gpu_configs`), the batch configs of an N > 1 run (`sharded_configs`), trackFrame on a rendered sequence, and the
canary child of the one-shot p2p collectives.  Their entries go to the DETAILS file (bench.py --details-out); the line carries a
handful of their numbers."""
import os
import sys

import numpy as np

from bench_core import (FP64_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, SIDE_CONFIGS, Runner, bounded_run, executed_fp64_flops)


def _counts_of(se, probs):
    """(valid pixels, S, problem) of every LOCAL problem of a sharded evaluation after one clean pass."""
    import torch
    se.evaluate_local(True)
    torch.cuda.synchronize()
    valid = se.valid.cpu().numpy()
    row, out = 0, []
    for p in probs:
        out.append((float(valid[row:row + p.F].sum()), p.S, p))
        row += p.F
    return out


def trackframe_config(M, ctx, dev):
    """BlurAwareDirectTracker::trackFrame on a GPU-rendered blurred sequence, reference-shaped configuration
    (blur_aware_direct_tracker.cpp:88-203,544-637): wall time of the mbavo_vo_track_frame calls, and the absolute trajectory
    error against the ground truth (product code only; the comparison with the CPU checker is bench_checks.trackframe_checker)."""
    from mba_vo_amd import sequence
    seq = sequence.make_sequence(ctx, H=480, W=640, M=8, device=dev)
    sequence.track_sequence(ctx, seq)  # warm-up: allocations, code objects
    runs = [sequence.track_sequence(ctx, seq) for _ in range(5)]
    per_frame = sorted(sum(f["seconds"] for f in r) / len(r) for r in runs)
    r0 = runs[0]
    gt_rel = sequence.gt_relative(ctx, seq)
    out = {
        "workload": "BlurAwareDirectTracker::trackFrame, 640x480, 4 levels, 30-px grid keypoints x 8-pixel pattern, k = 2, "
                    "S = 8, %d frames (GPU-rendered blurred sequence on a textured plane), LM loop on persistent evaluation "
                    "kernels" % len(r0),
        "ms_per_frame": round(1e3 * per_frame[len(per_frame) // 2], 4), "ms_per_frame_min": round(1e3 * per_frame[0], 4),
        "passes": len(runs), "frames": len(r0), "keyframes": int(sum(f["is_keyframe"] for f in r0)),
        "keypoints_level0": int(r0[0]["K0"]), "lm_trace_records": int(sum(f["num_trace"] for f in r0)),
        "poses_reproducible": bool(all(np.array_equal(a["T"], b["T"]) for r in runs[1:] for a, b in zip(r0, r))),
        "ate_gt": sequence.ate(r0, gt_rel),
        "ate_note": "RMSE over the frames of |t_est - t_gt| (metres of the synthetic scene; poses relative to the first "
                    "keyframe, no alignment): the tracker's accuracy on this sequence, product code only"}
    return out, seq, r0, gt_rel


def p2p_canary_child():
    """The one-shot p2p collectives between this run's ranks, in a process of their own (gloo carries the handles): exit code 0 iff
    set-up, 20 all-reduces and 20 all-gathers ran and returned the right numbers on this rank."""
    import torch
    import torch.distributed as dist
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MBAVO_CANARY_SHARED") == "1":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import mba_vo_amd as M
    from mba_vo_amd import shard
    dev = "cuda:%d" % local_rank
    ctx = M.capi.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    c = shard.P2PCollective(ctx, rank, world, max_doubles=1 << 12)
    n, m = 2437, 501
    ramp = torch.arange(n, dtype=torch.float64, device=dev) * 1e-3
    want = torch.zeros(n, dtype=torch.float64, device=dev)
    for r in range(world):
        want += (r + 1.0) + ramp
    ok = True
    for it in range(20):
        x = (rank + 1.0) + ramp
        c.allreduce(x, x, n)
        buf = torch.zeros(world * m, dtype=torch.float64, device=dev)
        buf[rank * m:(rank + 1) * m] = rank + 1.0 + it
        c.allgather(buf, m)
        torch.cuda.synchronize()
        ok = ok and bool(torch.allclose(x, want, rtol=1e-14, atol=0.0))
        ok = ok and all(bool((buf[r * m:(r + 1) * m] == r + 1.0 + it).all()) for r in range(world))
    c.close()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 3


def run_p2p_canary(shared_gpu):
    """this rank's canary child (p2p_canary_child): True iff it exited with 0 within its time"""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29511")) + 37)  # a rendezvous of its own, beside the parent's
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)                        # (rank 0 of the children hosts that store itself)
    if shared_gpu:
        env["MBAVO_CANARY_SHARED"] = "1"
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--p2p-canary"], env=env, stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, timeout=120)
    except subprocess.TimeoutExpired:
        return False, "timeout"
    return p.returncode == 0, "rc %d%s" % (p.returncode, (": " + p.stderr.decode(errors="replace").strip().splitlines()[-1][:200]) if p.returncode and p.stderr.strip() else "")


def sharded_configs(env):
    """N > 1 side configs (default workload only), bounded: {name: entry} on rank 0, {} elsewhere.
    N > 1: BASELINE configs[3] (512 pairs over the ranks: STRONG scaling, every GPU holds 512 / N pairs) in both shardings and
    both pair collectives, then the WEAK-scaling points north_star's "independent keyframe-pair alignments shard naturally
    across the 8 GPUs" asks for (512 pairs PER rank), bounded -- the driver's scaling run only launches the default workload,
    so the batch's scaling points ride in its line"""
    import torch
    import torch.distributed as dist
    M, ctx, dev, rank, world, args, coll, use_dist = env.M, env.ctx, env.dev, env.rank, env.world, env.args, env.coll, env.use_dist
    sync, max_over_ranks, sum_over_ranks, per_rank = env.sync, env.max_over_ranks, env.sum_over_ranks, env.per_rank
    comm_profile, reduction_check, collective_name = env.comm_profile, env.reduction_check, env.collective_name
    cfgs = {}
    NP = args.batch_pairs
    cfg_failed = False

    def batch_entry(r, n, dt, kms, kname, scaling, chk):
        loc, red = comm_profile(r)
        c = r.local_counts()
        ps = sum_over_ranks(sum(px * S for px, S, _ in c))
        kr, lr, rr = per_rank(kms), per_rank(loc), per_rank(red)
        return {"workload": r.desc, "sharding": r.mode, "collective": collective_name(r), "n_gpus": world,
                "value": round(ps * n / dt / 1e6, 3), "unit": "Mpixel-samples/s", "scaling": scaling, "steps": n,
                "ms_per_step": round(dt / n * 1e3, 5), "kernel": kname, "pairs_per_rank": r.se.n_live,
                "per_rank": {"kernel_ms": [round(v, 6) for v in kr], "local_evaluation_ms": [round(v, 6) for v in lr],
                             "collective_ms": [round(v, 6) for v in rr]},
                "reduction_check": chk, "collective_doubles": int(r.se.count)}

    # the headline workload in north_star's own words -- "a final RCCL all-reduce of the normal equations": frame r on rank r,
    # the rank's block scattered into the 6N x 6N system on the device (mbavo_eval_batch_merged), ONE in-place all-reduce of the
    # systems (the line's own step moves the packed blocks by an all-gather and leaves the scatter to the consumer)
    try:
        r = Runner(M, ctx, "c2_dense", dev, rank, world, True, 0, shard_mode="frames", coll=coll)
        n, dt, kms, kname = bounded_run(M, ctx, r, min_steps=60, sync=sync)
        dt = max_over_ranks(dt)
        e = batch_entry(r, n, dt, kms, kname, "weak", reduction_check(r))
        e.pop("pairs_per_rank", None)
        if rank == 0:
            cfgs["c2_dense_frames_allreduce_of_systems"] = e
        del r
        torch.cuda.empty_cache()
    except Exception as e:
        if rank == 0:
            cfgs["c2_dense_frames_allreduce_of_systems"] = {"error": repr(e)}
        cfg_failed = True
    for mode, fmt, pc in (("pairs", 0, "allgather"), ("pairs", 0, "allreduce"), ("keypoints", 0, None), ("pairs", 2, "allgather")):
        if max_over_ranks(float(cfg_failed)) != 0.0:
            break
        key = "c4_batch512_" + mode + ("_allreduce" if pc == "allreduce" else "") + ("_packed" if fmt == 2 else "")  # (2: packed keyframes)
        try:
            r = Runner(M, ctx, "c4_batch512", dev, rank, world, True, fmt, shard_mode=mode, coll=coll, pair_collective=pc or "allgather",
                       pairs=NP if NP != 512 else None)
            n, dt, kms, kname = bounded_run(M, ctx, r, min_steps=60, sync=sync)
            dt = max_over_ranks(dt)
            e = batch_entry(r, n, dt, kms, kname, "strong", reduction_check(r))
            if rank == 0:
                cfgs[key] = e
            del r
            torch.cuda.empty_cache()
        except Exception as e:
            if rank == 0:
                cfgs[key] = {"error": repr(e)}
            cfg_failed = True
            break  # the ranks may have diverged: no further collective configs
    # weak scaling of the evaluation: NP pairs PER rank (rank r renders and owns pairs b % N == r of an N * NP-pair sequence),
    # ONE in-place all-gather of the N * NP packed blocks.  No rank holds the whole workload, so the check is by samples:
    # rank 0 renders one pair of every other rank, evaluates it alone and compares the gathered block (1e-12: another tile
    # partition).
    if max_over_ranks(float(cfg_failed)) == 0.0:
        try:
            from mba_vo_amd import shard as sh, workloads as wl
            BT = NP * world
            mine = sh.pairs_of_rank(BT, rank, world)
            batch = wl.RenderedPairBatch(ctx, BT, S=8, k=4, device=dev, seed=1, pairs=mine, grad_fp16=2)
            r = Runner.__new__(Runner)
            r.M, r.ctx, r.name, r.world, r.rank, r.mode, r.sequential, r.wl, r.cost_only = M, ctx, "c4_batch512", world, rank, "pairs", False, wl, False
            r.dw, r.probs = batch, [batch.probs[b] for b in mine]
            r.desc = "%d pairs PER RANK of one rendered blurred sequence of %d (packed keyframes), pair b on rank b %% N" % (NP, BT)
            r.se = sh.ShardedEvaluation(ctx, batch.array, 4, rank, world, "pairs", dev, collective=coll, frames_per_pair=[1] * BT)
            n, dt, kms, kname = bounded_run(M, ctx, r, min_steps=60, sync=sync)
            dt = max_over_ranks(dt)
            r.se.step(True)
            torch.cuda.synchronize()
            chk = {"by": "samples: rank 0 re-renders one pair of every rank, evaluates it alone", "collective": collective_name(r)}
            if rank == 0:
                samples = [rr_ + world * ((NP // 2) if NP > 1 else 0) for rr_ in range(world)]
                sb = wl.RenderedPairBatch(ctx, BT, S=8, k=4, device=dev, seed=1, pairs=samples, grad_fp16=2)
                one = (M.capi.Problem * len(samples))(*[sb.array[b] for b in samples])
                fb1 = torch.zeros(len(samples) * sb.E, dtype=torch.float64, device=dev)
                M.capi.check(ctx.lib.mbavo_eval_batch(ctx.handle, len(samples), one, 4, 1, fb1.data_ptr(), None, None), "mbavo_eval_batch")
                torch.cuda.synchronize()
                worst = 0.0
                for i, b in enumerate(samples):
                    a, g = fb1.view(-1, sb.E)[i], r.se.blocks_of_pair(b)[0]
                    worst = max(worst, float((a - g).abs().max() / a.abs().max()))
                chk.update(max_rel_diff_vs_single_gpu=worst, ok=bool(worst <= 1e-12), sampled_pairs=samples)
                del sb
            r.local_counts = lambda se=r.se, pr=r.probs: _counts_of(se, pr)
            e = batch_entry(r, n, dt, kms, kname, "weak", chk)
            if rank == 0:
                cfgs["c4_batch512_pairs_weak_packed"] = e
            del r, batch
            torch.cuda.empty_cache()
        except Exception as e:
            if rank == 0:
                cfgs["c4_batch512_pairs_weak_packed"] = {"error": repr(e)}
            cfg_failed = True
    # whole alignments sharded: the device-side LM on every rank's own pairs, one all-gather of the records at the end;
    # strong (NP pairs in all) and weak (NP pairs per rank)
    for key, weak in (("lm_batch512_pairs", False), ("lm_batch_pairs_weak", True)):
        if max_over_ranks(float(cfg_failed)) != 0.0:  # (decided together: a rank that skipped would leave the others in a collective)
            break
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import lm_bench
            line = lm_bench.sharded_line(M, ctx, dev, rank, world, (dist.barrier if use_dist and world > 1 else (lambda: None)),
                                         max_over_ranks, sum_over_ranks, B=NP * (world if weak else 1), coll=coll, weak=weak)
            if rank == 0:
                cfgs[key] = line
        except Exception as e:
            if rank == 0:
                cfgs[key] = {"error": repr(e)}
            cfg_failed = True
    return cfgs


def single_gpu_configs(env):
    """The other BASELINE configs, bounded (N = 1 only): value, step time, dominant kernel time, both fractions; trackFrame; the
    batched LM.  Returns ({name: entry}, track) -- track = (sequence, per-frame results, ground truth) for the checker leg, or None."""
    import torch
    M, ctx, dev, args = env.M, env.ctx, env.dev, env.args
    cfgs, track = {}, None
    todo = [(n, False, False) for n in SIDE_CONFIGS if n != args.workload] + [("c5_1080p", True, False)]
    if args.workload == "c2_dense":
        todo.insert(0, ("c2_dense", False, True))
    # named extras: the keyframe in the two lossless compact formats (mbavo_problem.grad_fp16 = 1: half pairs, 2: packed words)
    todo += [("c3_batch64_shared", False, False), ("c4_batch512", 1, False), ("c4_batch512", 2, False), ("c3_batch64", 2, False),
             ("c2_dense", 2, False)]
    todo = [t + (False, 4) for t in todo]
    # the cost-only evaluation (half of every LM iteration) and the reference's default spline degree
    todo += [("c2_dense", False, False, True, 4), ("c3_batch64", False, False, True, 4), ("c2_semidense", False, False, True, 4),
             ("c2_dense", False, False, False, 2), ("c2_semidense", False, False, False, 2), ("c2_dense", False, False, True, 2)]
    for name, half, seq_levels, cost_only, kdeg in todo:
        key = name + ("_k2" if kdeg == 2 else "") + ("_packed" if int(half) == 2 else "_fp16grad" if half else "") + \
            ("_sequential" if seq_levels else "") + ("_cost_only" if cost_only else "")
        try:
            r = Runner(M, ctx, name, dev, 0, 1, False, half, sequential=seq_levels, cost_only=cost_only, k=kdeg)
            if seq_levels:
                # the step is timed WITHOUT events (an event pair costs a launch gap), the levels' kernels in a second run with
                # an event pair on EVERY launch: kernel_ms = the sum over the levels' dominant kernels (mean per launch x levels)
                n, dt, _, kname = bounded_run(M, ctx, r, every=0)
                _, _, kms_mean, _ = bounded_run(M, ctx, r, every=1, seconds=0.1)
                kms = kms_mean * len(r._seq)
            else:
                n, dt, kms, kname = bounded_run(M, ctx, r)
            c = r.local_counts()
            fl, nb, tf, gbs = r.figures(c, kms)
            ex, _ = executed_fp64_flops(name + ("_k2" if kdeg == 2 else "") + ("_cost_only" if cost_only else ""), kname)
            cfgs[key] = {"workload": r.desc, "flops_alg": "122 PS + 13 PX (cost-only)" if cost_only else "PS (363 + 48 k) + PX (2 E + 12 k + 13)", "value": round(sum(px * S for px, S, _ in c) * n / dt / 1e6, 3),
                         "unit": "Mpixel-samples/s", "steps": n, "ms_per_step": round(dt / n * 1e3, 5), "kernel": kname,
                         "kernel_ms": round(kms, 6), "frac": round(tf / FP64_PEAK_TFLOPS, 5),
                         "frac_executed": round(ex / (kms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5) if ex and kms > 0 and not seq_levels else None,
                         "step_frac": round(fl / (dt / n) / 1e12 / FP64_PEAK_TFLOPS, 5),
                         "hbm_frac_algorithmic": round(gbs / HBM_PEAK_GBS, 6)}
            if r.nbytes_upper != nb:  # pairs with their own images: compulsory = distinct taps; the no-reuse gather bound beside it
                cfgs[key]["hbm_frac_algorithmic_upper"] = round(r.nbytes_upper / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if kms > 0 else None
                cfgs[key]["distinct_taps"] = r.distinct_summary()
            if seq_levels:
                cfgs[key]["note"] = "kernel_ms: SUM over the four levels' dominant kernels (an event pair on every launch, in a run of its own); " \
                                    "frac: the four levels' flops over that sum; step_frac: the same flops over the whole sequential step " \
                                    "(timed without events)"
            del r
            torch.cuda.empty_cache()
        except Exception as e:  # a failing side config must not cost the headline line
            cfgs[key] = {"error": repr(e)}
    try:
        cfgs["trackframe_640x480"], *track = trackframe_config(M, ctx, dev)
    except Exception as e:
        cfgs["trackframe_640x480"] = {"error": repr(e)}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import lm_bench
        cfgs["lm_batch64"] = lm_bench.bench_line(M, ctx, dev)
    except Exception as e:
        cfgs["lm_batch64"] = {"error": repr(e)}
    try:  # the reference's default degree through the same loop
        cfgs["lm_batch64_k2"] = lm_bench.bench_line_k2(M, ctx, dev)
    except Exception as e:
        cfgs["lm_batch64_k2"] = {"error": repr(e)}
    try:  # configs[3]'s pairs through the same loop (device side only)
        cfgs["lm_batch512"] = lm_bench.bench_line(M, ctx, dev, B=512, host_pairs=0)
    except Exception as e:
        cfgs["lm_batch512"] = {"error": repr(e)}
    return cfgs, track
