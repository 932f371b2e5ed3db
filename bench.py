#!/usr/bin/env python3
"""bench.py -- throughput of the blur-aware tracking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one Gauss-Newton iteration of the hot path = one full H/g evaluation
(evaluate_cost_hessian_gradient, ba_tracker/spline_update_step.cpp:97-241) of every
problem of the workload, inputs resident in HBM, outputs (packed normal-equation blocks)
left in HBM.  Default workload = BASELINE.json configs[1]: one 640x480 keyframe pair, 4-level
pyramid, 8 blur samples, 4 control poses (cubic, k = 4), dense mode (every pixel of every level a
P=1 patch).  N > 1: one process per GPU (torch.distributed, backend nccl == RCCL), every rank aligns
its own independent pair (weak scaling) and the packed normal equations are summed with one
all-reduce per step over xGMI; value = pixel-samples of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (metric, value, roofline, cpu_baseline, ...).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == FP64 matrix (v_mfma_f64) peak, 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2_dense",
                    choices=["c2_dense", "c2_semidense", "c1_dense", "c3_batch64", "c4_batch512", "c5_1080p"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grad-fp16", action="store_true",
                    help="gradient pyramid stored as IEEE half pairs (BASELINE configs[4]: lossless for 8-bit images)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="HIP-event pair around the dominant kernel on every n-th timed step (events cost launch gaps)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the bounded CPU baseline sample")
    return ap.parse_args()


def build_workload(name, rank, world):
    from mba_vo_amd import workloads as wl
    seed = 1 + rank
    if name == "c2_dense":
        return wl.pyramid_pair(480, 640, 4, S=8, k=4, N=4, mode="dense", seed=seed), \
            "640x480 pair, 4-level pyramid, S=8 blur samples, N=4 control poses (k=4), dense P=1 (configs[1])"
    if name == "c2_semidense":
        return wl.pyramid_pair(480, 640, 4, S=8, k=4, N=4, mode="semidense", seed=seed), \
            "640x480 pair, 4-level pyramid, S=8, N=4, semi-dense 30px grid keypoints x 8-pixel pattern (configs[1], reference-shaped)"
    if name == "c1_dense":
        return wl.pyramid_pair(480, 640, 1, S=1, k=4, N=4, mode="dense", seed=seed), \
            "640x480 pair, 1 level, S=1 (sharp degenerate case), dense (configs[0])"
    if name == "c3_batch64":
        return wl.pair_batch(64, S=8, k=4, N=4, mode="semidense", seed=seed), \
            "batch of 64 independent 640x480 pairs, S=8, N=4, semi-dense (configs[2])"
    if name == "c4_batch512":
        per = max(1, 512 // world)
        return wl.pair_batch(per, S=8, k=4, N=4, mode="semidense", seed=seed), \
            "batch of 512 pairs sharded over %d GPU(s) (%d per rank), S=8, N=4, semi-dense (configs[3])" % (world, per)
    if name == "c5_1080p":
        return wl.pyramid_pair(1080, 1920, 1, S=16, k=4, N=6, mode="dense", seed=seed), \
            "1920x1080 pair, 1 level, S=16, N=6 control poses, dense (configs[4], fp32 gradient pyramid)"
    raise ValueError(name)


def measured_hbm_traffic(workload):
    """HBM bytes per launch of the fused kernel from the committed TCC counter passes (tools/hbm_traffic.sh ->
    profiles/r01_hbm_counters.json; the counters need rocprofv3, so they are collected outside this process).
    FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE under-counts 2x on gfx950 (calibrated in the same file on a
    256 MiB copy: 131084 KiB read), WRITE_SIZE is exact."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_counters.json" if workload == "c2_dense" else "r01_hbm_counters_%s.json" % workload)
    if not os.path.exists(path):
        return None
    try:
        h = json.load(open(path))
        rd = [v["mean"] for k, v in h.items() if k.startswith("bench|FETCH_SIZE|") and "k_fused" in k][0]
        wr = [v["mean"] for k, v in h.items() if k.startswith("bench|WRITE_SIZE|") and "k_fused" in k][0]
        return (2.0 * rd + wr) * 1024.0
    except Exception:
        return None


def cpu_baseline(probs, budget_s):
    """Oracle (plain-C port of the reference path) timed on the host cores on the same workload:
    one evaluation with all cores, repeated while the budget allows, plus one single-thread
    evaluation of the coarser levels for scale.  Returns (dict, frame_blocks of the last run)."""
    from oracle import binding as B
    B.build()
    # scalar port on ONE host core by default: the sandboxed hosts here expose many logical CPUs but give a
    # process ~1-2 cores of real CPU time, so a multi-threaded number would only measure the throttle.
    cores = max(1, int(os.environ.get("MBAVO_CPU_THREADS", "1")))
    E = B.packed_len(probs[0].k)
    plist, keeps = [], []
    for p in probs:
        op, keep = B.make_problem(p.S, p.F, p.K, p.P, p.k, p.N, p.H, p.W, p.ref, p.grad, p.cur, p.kp_xy, p.kp_z,
                                  p.pattern, p.intr, p.cap, p.exp, p.t0, p.dt, p.knots_t, p.knots_R, p.start_idx,
                                  p.huber)
        plist.append(op)
        keeps.append(keep)
    ps = sum(p.pixel_samples for p in probs)
    R = B.ref()
    if R is not None and hasattr(R, "ref_compute_pixel_jacobian_residual") and os.environ.get("MBAVO_CPU_BASELINE", "reference") == "reference":
        # kind "reference": the per-sample arithmetic is the reference's own code (spline functors,
        # compute_pixel_intensity<double>, Core::MatrixMatrixMultiply) compiled from its sources into oracle/_ref;
        # the kernels' launch geometry and block reductions cannot be compiled and are the oracle's restatement
        args = [dict(S=p.S, F=p.F, K=p.K, P=p.P, k=p.k, N=p.N, H=p.H, W=p.W, ref_img=p.ref, ref_dIxy=p.grad, cur_imgs=p.cur,
                     kp_xy=p.kp_xy, kp_z=p.kp_z, pattern=p.pattern, intr=p.intr, cap=p.cap, exp_t=p.exp, t0=p.t0, dt=p.dt,
                     knots_t=p.knots_t, knots_R=p.knots_R, huber_a=p.huber) for p in probs]
        t_all, reps, blocks = 0.0, 0, None
        while reps < 1 or (t_all + t_all / reps < budget_s and reps < 20):
            t0 = time.perf_counter()
            blocks = [B.evaluate_with_reference(a) for a in args]
            t_all += time.perf_counter() - t0
            reps += 1
        return dict(value=round(ps * reps / t_all / 1e6, 3), unit="Mpixel-samples/s", cores=1, kind="reference",
                    sample="%d full H/g evaluation(s) of the same workload (%d pixel-samples each) on ONE host core: "
                           "per-sample code = the reference's compute_pixel_intensity<double>, C2/C4 spline functors and "
                           "Core::MatrixMatrixMultiply compiled from its sources (oracle/_ref, g++ -O2 -ffp-contract=off); "
                           "kernel launch geometry, Huber and block reductions = oracle restatement; %.1f s"
                           % (reps, ps, t_all)), np.concatenate(blocks, 0)
    t_all, reps, blocks = 0.0, 0, None
    while reps < 1 or (t_all + t_all / reps < budget_s and reps < 20):
        t0 = time.perf_counter()
        blocks = [B.evaluate_fast(op, num_threads=cores)["frame_blocks"] for op in plist]
        t_all += time.perf_counter() - t0
        reps += 1
    value = ps * reps / t_all / 1e6
    return dict(value=round(value, 3), unit="Mpixel-samples/s", cores=cores, kind="port",
                sample="%d full H/g evaluation(s) of the same workload (%d pixel-samples each) with "
                       "oracle/mbavo_oracle.c orc_evaluate_fast, %d OpenMP threads, gcc -O2 -ffp-contract=off, %.1f s"
                       % (reps, ps, cores, t_all)), np.concatenate(blocks, 0)


def main():
    args = parse()
    # The one JSON line goes to the REAL stdout; everything else written to file descriptor 1 by this process or by the
    # libraries it loads goes to stderr.  RCCL prints a version banner to C stdout when the first communicator is
    # created, block-buffered when stdout is a pipe and flushed only at exit -- i.e. AFTER the JSON line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("MBAVO_BENCH_FORCE_DIST") == "1"  # the env switch tests the N > 1 code on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import mba_vo_amd as M
    from mba_vo_amd import workloads as wl
    if not os.path.exists(M.LIB_PATH):
        raise SystemExit("libmbavo.so missing: run python -c 'import __graft_entry__ as g; g.build()'")
    dev = "cuda:%d" % local_rank
    stream = torch.cuda.current_stream()
    ctx = M.capi.Context(local_rank, stream=stream.cuda_stream)

    probs, desc = build_workload(args.workload, rank, world)
    if args.grad_fp16:
        for p in probs:
            p.grad_fp16 = True
        desc += ", fp16 gradient pyramid"
    dw = wl.DeviceWorkload(probs, device=dev)

    # N > 1: the final sum of the packed normal equations over xGMI (RCCL), one all-reduce per step.  Default: in place,
    # ordered after the finalize kernel (+8 us per step measured with a 1-rank group).  MBAVO_BENCH_ALLREDUCE=async runs
    # it on the collective's own stream, overlapped with the next step through two output buffers (the stream
    # hand-offs then cost +23 us per step on one GPU, so it only pays when the collective itself is slower than that)
    bufs = [dw.frame_blocks, torch.zeros_like(dw.frame_blocks)]
    pending = [None, None]
    count = [0]

    sync_allreduce = os.environ.get("MBAVO_BENCH_ALLREDUCE", "sync") == "sync"

    def step():
        if sync_allreduce:  # in place on the default buffer, ordered by the collective's own stream semantics
            dw.step(ctx, True)
            if use_dist:
                dist.all_reduce(dw.frame_blocks, op=dist.ReduceOp.SUM)
            return
        b = count[0] & 1
        count[0] += 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        dw.step(ctx, True, out=bufs[b])
        if use_dist:
            pending[b] = dist.all_reduce(bufs[b], op=dist.ReduceOp.SUM, async_op=True)

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    ctx.lib.mbavo_profile(ctx.handle, args.time_every)  # HIP-event pair around the fused kernel of every n-th step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fused_ms, nlaunch = np.zeros(1), np.zeros(1, np.int32)
    M.capi.check(ctx.lib.mbavo_profile_read(ctx.handle, M.capi.dp(fused_ms), M.capi.ip(nlaunch)), "mbavo_profile_read")
    ctx.lib.mbavo_profile(ctx.handle, 0)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # units actually processed: in-bounds pixels x S (one clean evaluation, outside the timed region)
    dw.step(ctx, True)
    torch.cuda.synchronize()
    valid = dw.valid.cpu().numpy()
    fb_gpu = dw.frame_blocks.cpu().numpy().reshape(dw.nbf, dw.E)
    row, valid_px = 0, []
    for p in probs:
        valid_px.append(float(valid[row:row + p.F].sum()))
        row += p.F
    ps_rank = sum(v * p.S for v, p in zip(valid_px, probs))
    ps_launched = sum(p.pixel_samples for p in probs)
    tot = torch.tensor([ps_rank], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    ps_all = float(tot.item())

    if rank == 0:
        value = ps_all * args.steps / elapsed / 1e6
        flops = wl.algorithmic_flops(probs, valid_px)
        nbytes = wl.algorithmic_bytes(probs)
        k_ms = float(fused_ms[0]) / max(int(nlaunch[0]), 1)
        ach_tf = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        ach_gbs = nbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "Mpixel-samples/s per GN iteration (640x480, 4-lvl pyr, 8 blur samples)",
            "value": round(value, 3), "unit": "Mpixel-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "problems_per_rank": len(probs),
                       "pixel_samples_per_step_per_rank": ps_rank, "pixel_samples_launched": ps_launched,
                       "parallelism": "independent pairs per GPU + one RCCL all-reduce of the packed J^T J blocks per step" if world > 1 else "1 GPU"},
            "roofline": {"bound": "mfma", "achieved": round(ach_tf, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tf / FP64_PEAK_TFLOPS, 5), "traffic": measured_hbm_traffic(args.workload),
                         "kernel": "k_fused<4,true>", "kernel_ms": round(k_ms, 6), "launches_timed": int(nlaunch[0]),
                         "algorithmic_flops_per_launch": flops,
                         "note": "FP64 pipe is the binding roofline (FP64 vector peak == f64 MFMA peak, 78.6 TFLOP/s): "
                                 "intensity ~150 flop/B >> 9.8 flop/B balance; flops counted as the reference source "
                                 "writes them (SURVEY.md 8d), so CSE in the kernel raises this fraction"},
            "roofline_hbm": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": nbytes,
                             "traffic": measured_hbm_traffic(args.workload),
                             "note": "compulsory bytes only; compute-bound kernel, low by construction; traffic = "
                                     "(2*FETCH_SIZE + WRITE_SIZE) KiB from profiles/r01_hbm_counters*.json"},
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            cb, fb_cpu = cpu_baseline(probs, args.cpu_seconds)
            scale = np.abs(fb_cpu).max(axis=1, keepdims=True)
            cb["gpu_vs_cpu_max_rel_diff"] = float((np.abs(fb_gpu - fb_cpu) / scale).max())
            out["cpu_baseline"] = cb
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
