#!/usr/bin/env python3
"""bench.py -- throughput of the blur-aware tracking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one Gauss-Newton iteration of the hot path = one full H/g evaluation
(evaluate_cost_hessian_gradient, ba_tracker/spline_update_step.cpp:97-241) of every problem of the workload, inputs
resident in HBM, outputs (packed normal-equation blocks) left in HBM.  Default workload = BASELINE.json configs[1]: one
640x480 keyframe pair, 4-level pyramid, 8 blur samples, 4 control poses (cubic, k = 4), dense mode (every pixel of every
level a P=1 patch).

N > 1 (one process per GPU, launched by torch.distributed.run): ONE joint problem is sharded over the ranks --
  * c2_dense (default) and the other single-pair workloads: N blurred frames against the same keyframe on one spline
    segment, frame r on rank r (weak scaling: the per-GPU work is the N = 1 workload); every rank scatters its frame's
    packed blocks into the 6N x 6N normal equations on the device (mbavo_merge_device) and the partial systems are
    summed with ONE all-reduce per step over xGMI;
  * c4_batch512: every pair's keypoints are sharded over the ranks (strong scaling) and the 512 x E packed blocks are
    summed with ONE all-reduce per step
-- through the PRODUCT's collective (mbavo_allreduce_blocks on the context's own RCCL communicator); torch.distributed
(backend nccl == RCCL) is the rendezvous, the barrier and the max-over-ranks of the clock.  After the timed region the
reduced normal equations are compared with a single-GPU evaluation of the whole joint problem (rank 0; 1e-12).
value = pixel-samples of all ranks / max-over-ranks time.

The timed region (exactly K steps between barrier + synchronize) is repeated until >= 0.3 s have been timed and the
MEDIAN region is reported, so that K = 20 does not rest on 1 ms of GPU work.  At N = 1 every other BASELINE config is
then run for a bounded time and reported under "configs"; the CPU baseline (1 thread and all host threads) comes last.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == FP64 matrix (v_mfma_f64) peak: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz;
                          # one shared pipe (tools/micro/mfma_valu_overlap.hip; 75.2 TFLOP/s sustained by v_mfma_f64_16x16x4)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s

WORKLOADS = ["c2_dense", "c2_semidense", "c1_dense", "c3_batch64", "c4_batch512", "c5_1080p"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2_dense", choices=WORKLOADS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the bounded runs of the other BASELINE configs")
    ap.add_argument("--grad-fp16", action="store_true",
                    help="gradient pyramid stored as IEEE half pairs (BASELINE configs[4]: lossless for 8-bit images)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="HIP-event pair around the dominant kernel on every n-th timed step (events cost launch gaps)")
    ap.add_argument("--min-seconds", type=float, default=0.3, help="repeat the K-step region until this much was timed")
    ap.add_argument("--no-spin-sync", action="store_true", help="synchronize without polling the stream first (A/B of the bracket's own cost)")
    ap.add_argument("--max-repeats", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="budget of EACH bounded CPU baseline sample (1 and T threads)")
    return ap.parse_args()


def build_workload(name, frames=1, seed=1):
    """(list of Prob, description, sharding mode at N > 1)"""
    from mba_vo_amd import workloads as wl
    if name == "c2_dense":
        return wl.pyramid_pair(480, 640, 4, S=8, k=4, N=4, mode="dense", seed=seed, frames=frames), \
            "640x480 pair, 4-level pyramid, S=8 blur samples, N=4 control poses (k=4), dense P=1 (configs[1]); synthetic " \
            "band-limited noise keyframe, current image = shifted keyframe + noise", "frames"
    if name == "c2_semidense":
        return wl.pyramid_pair(480, 640, 4, S=8, k=4, N=4, mode="semidense", seed=seed, frames=frames), \
            "640x480 pair, 4-level pyramid, S=8, N=4, semi-dense 30px grid keypoints x 8-pixel pattern (configs[1], " \
            "reference-shaped)", "frames"
    if name == "c1_dense":
        return wl.pyramid_pair(480, 640, 1, S=1, k=4, N=4, mode="dense", seed=seed, frames=frames), \
            "640x480 pair, 1 level, S=1 (sharp degenerate case), dense (configs[0])", "frames"
    if name == "c3_batch64":
        return wl.pair_batch(64, S=8, k=4, N=4, mode="semidense", seed=seed), \
            "batch of 64 independent 640x480 pairs, S=8, N=4, semi-dense; one shared keyframe, every pair its own knots and " \
            "its own shifted-noise current image (configs[2] shape; not a rendered blurred sequence)", "keypoints"
    if name == "c4_batch512":
        return wl.pair_batch(512, S=8, k=4, N=4, mode="semidense", seed=seed), \
            "batch of 512 independent 640x480 pairs, S=8, N=4, semi-dense; one shared keyframe, every pair its own knots " \
            "and shifted-noise current image (configs[3] shape)", "keypoints"
    if name == "c5_1080p":
        return wl.pyramid_pair(1080, 1920, 1, S=16, k=4, N=6, mode="dense", seed=seed, frames=frames), \
            "1920x1080 pair, 1 level, S=16, N=6 control poses, dense (configs[4])", "frames"
    raise ValueError(name)


def committed_counters(kind, workload):
    """Counter extracts committed under profiles/ (the PMC passes need rocprofv3 and are collected outside this process,
    tools/hbm_traffic.sh / tools/pmc_fp64.sh): newest round first.  kind 'hbm_counters' | 'pmc_fp64'."""
    suffix = "" if workload == "c2_dense" else "_" + workload
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_%s%s.json" % (kind, suffix))), reverse=True):
        try:
            return json.load(open(path)), os.path.basename(path)
        except Exception:
            continue
    return None, None


def measured_hbm_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel from the committed TCC counter passes.  FETCH_SIZE / WRITE_SIZE are
    KiB; FETCH_SIZE under-counts 2x on gfx950 (calibrated in the same file on a 256 MiB copy), WRITE_SIZE is exact."""
    h, src = committed_counters("hbm_counters", workload)
    if h is None:
        return None, None
    base = kernel.split("<")[0]
    try:
        rd = [v["mean"] for k, v in h.items() if k.startswith("bench|FETCH_SIZE|") and base + "<" in k][0]
        wr = [v["mean"] for k, v in h.items() if k.startswith("bench|WRITE_SIZE|") and base + "<" in k][0]
        return (2.0 * rd + wr) * 1024.0, src
    except Exception:
        return None, None


def executed_fp64_flops(workload, kernel):
    """FP64 flops the dominant kernel EXECUTES per launch, from the committed SQ instruction counters
    (tools/pmc_fp64.sh): 64 lanes x (2 FMA + ADD + MUL + TRANS) + 512 x MFMA_MOPS_F64.  Bounded by the pipe, unlike the
    reference-flop count of SURVEY 8(d), which the kernel undercuts by CSE."""
    h, src = committed_counters("pmc_fp64", workload)
    if h is None:
        return None, None
    base = kernel.split("<")[0]
    for k, v in h.items():
        if base + "<" in k and "flops_fp64_per_launch" in v:
            return float(v["flops_fp64_per_launch"]), src
    return None, None


def issue_busy_fraction(workload, kernel, k_ms):
    """Share of the kernel's duration in which a SIMD's VALU / matrix issue port is busy, from the committed SQ counters
    (tools/pmc_all.sh): ((SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 cycles + SQ_VALU_MFMA_BUSY_CYCLES) / 1024 SIMDs / kernel
    cycles at the nominal 2.4 GHz.  Counts every vector instruction (fp32 bilinear, integer, moves), not only flops."""
    h, src = committed_counters("pmc_sq", workload)
    if h is None or k_ms <= 0:
        return None
    base = kernel.split("<")[0]
    for k, v in h.items():
        if base + "<" in k and "SQ_INSTS_VALU" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            cyc = (v["SQ_INSTS_VALU"] - v.get("SQ_INSTS_MFMA", 0.0)) * 4.0 + v["SQ_VALU_MFMA_BUSY_CYCLES"]
            return round(cyc / 1024.0 / (k_ms * 1e-3 * 2.4e9), 4)
    return None


def cpu_baseline(probs, budget_s):
    """The reference's per-sample code (oracle/_ref; kind "reference") or the oracle's fused port (kind "port") timed on
    the host cores on the SAME workload: one sample on 1 thread and one on all host threads, each bounded by `budget_s`
    (a whole number of full evaluations; at least one).  Returns (dict, frame_blocks of the last evaluation)."""
    from oracle import binding as B
    B.build()
    T = max(1, int(os.environ.get("MBAVO_CPU_THREADS", str(os.cpu_count() or 1))))
    ps = sum(p.pixel_samples for p in probs)
    R = B.ref()
    use_ref = R is not None and hasattr(R, "ref_compute_pixel_jacobian_residual") and \
        os.environ.get("MBAVO_CPU_BASELINE", "reference") == "reference"
    if use_ref:
        args = [dict(S=p.S, F=p.F, K=p.K, P=p.P, k=p.k, N=p.N, H=p.H, W=p.W, ref_img=p.ref, ref_dIxy=p.grad, cur_imgs=p.cur,
                     kp_xy=p.kp_xy, kp_z=p.kp_z, pattern=p.pattern, intr=p.intr, cap=p.cap, exp_t=p.exp, t0=p.t0, dt=p.dt,
                     knots_t=p.knots_t, knots_R=p.knots_R, huber_a=p.huber) for p in probs]
        run = lambda threads: [B.evaluate_with_reference(a, threads=threads) for a in args]
        kind = "reference"
        how = ("per-sample code = the reference's compute_pixel_intensity<double>, C2/C4 spline functors and "
               "Core::MatrixMatrixMultiply compiled from its sources (oracle/_ref, g++ -O2 -ffp-contract=off); kernel launch "
               "geometry, Huber and block reductions = oracle restatement; keypoint chunks of 4096 spread over the threads")
    else:
        plist, keeps = [], []
        for p in probs:
            op, keep = B.make_problem(p.S, p.F, p.K, p.P, p.k, p.N, p.H, p.W, p.ref, p.grad, p.cur, p.kp_xy, p.kp_z,
                                      p.pattern, p.intr, p.cap, p.exp, p.t0, p.dt, p.knots_t, p.knots_R, p.start_idx,
                                      p.huber)
            plist.append(op)
            keeps.append(keep)
        run = lambda threads: [B.evaluate_fast(op, num_threads=threads)["frame_blocks"] for op in plist]
        kind = "port"
        how = "oracle/mbavo_oracle.c orc_evaluate_fast (fused OpenMP port), gcc -O2 -ffp-contract=off"

    def sample(threads):
        t_all, reps, blocks = 0.0, 0, None
        while reps < 1 or (t_all + t_all / reps < budget_s and reps < 20):
            t0 = time.perf_counter()
            blocks = run(threads)
            t_all += time.perf_counter() - t0
            reps += 1
        return ps * reps / t_all / 1e6, reps, t_all, blocks

    v1, r1, t1, blocks = sample(1)
    out = dict(value=round(v1, 3), unit="Mpixel-samples/s", cores=1, kind=kind,
               sample="%d full H/g evaluation(s) of the same workload (%d pixel-samples each) on 1 thread, %.1f s; %s"
                      % (r1, ps, t1, how), host_logical_cpus=os.cpu_count())
    if T > 1:
        vT, rT, tT, blocks = sample(T)
        out["all_threads"] = dict(value=round(vT, 3), unit="Mpixel-samples/s", cores=T,
                                  sample="%d evaluation(s) on %d threads, %.1f s" % (rT, T, tT))
        if vT > v1:  # the better of the two is the quoted baseline, its thread count stated
            out.update(value=round(vT, 3), cores=T)
            out["single_thread"] = dict(value=round(v1, 3), cores=1)
            out["sample"] = "%d full H/g evaluation(s) of the same workload (%d pixel-samples each) on %d threads, %.1f s " \
                            "(1 thread: %.3f Mpixel-samples/s); %s" % (rT, ps, T, tT, v1, how)
    return out, np.concatenate(blocks, 0)


class Runner:
    """One workload resident on this rank's GPU: step(), unit counts, roofline figures."""

    def __init__(self, M, ctx, name, dev, rank, world, sharded, grad_fp16=False):
        from mba_vo_amd import shard, workloads as wl
        self.M, self.ctx, self.name, self.world, self.rank = M, ctx, name, world, rank
        self.probs, self.desc, self.mode = build_workload(name, frames=world if sharded else 1)
        if grad_fp16:
            for p in self.probs:
                p.grad_fp16 = True
            self.desc += ", fp16 gradient pyramid"
        self.dw = wl.DeviceWorkload(self.probs, device=dev)
        self.se = shard.ShardedEvaluation(ctx, self.dw.array, self.dw.k, rank, world, self.mode, dev) if sharded else None
        self.wl = wl

    def step(self):
        if self.se is not None:
            self.se.step(True)
        else:
            self.dw.step(self.ctx, True)

    def local_counts(self):
        """(valid pixels per local problem, S per local problem) after one clean evaluation."""
        import torch
        if self.se is not None:
            self.se.evaluate_local(True)
            torch.cuda.synchronize()
            valid = self.se.valid.cpu().numpy()
            row, out = 0, []
            for b in range(self.se.B):
                F = self.se.shards[b].F
                out.append((float(valid[row:row + F].sum()), self.probs[b].S, self.probs[b]))
                row += F
            return out
        self.dw.step(self.ctx, True)
        torch.cuda.synchronize()
        valid = self.dw.valid.cpu().numpy()
        row, out = 0, []
        for p in self.probs:
            out.append((float(valid[row:row + p.F].sum()), p.S, p))
            row += p.F
        return out

    def figures(self, counts, k_ms):
        """Algorithmic flops / bytes of THIS rank's launch and the derived rates."""
        from mba_vo_amd import synth
        flops = 0.0
        for px, S, p in counts:
            E = synth.packed_len(p.k)
            flops += px * S * (363 + 48 * p.k) + px * (2 * E + 12 * p.k + 13)
        nbytes = self.wl.algorithmic_bytes(self.probs, None if self.se is None else (self.mode, self.rank, self.world))
        ach_tf = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        ach_gbs = nbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        return flops, nbytes, ach_tf, ach_gbs


def launches_per_step(kernel):
    """What one evaluation enqueues, from the dominant kernel's label (Engine::last_kernel)."""
    if kernel.startswith("k_fused_sp<") and kernel.endswith(",true>"):
        return "1: k_fused_sp<.., ONE> (pose entries in its prologue, finalize by the last workgroup of a slot)"
    if kernel.startswith("k_fused<") and kernel.endswith(",true>"):
        return "2: k_fused<.., POSE> (pose entries in its prologue: inside kernel_ms, so frac is lower than the sample loop's) + k_finalize"
    return "3: k_pose_table + fused kernel + k_finalize"


def kernel_name(ctx):
    return ctx.lib.mbavo_last_kernel(ctx.handle).decode()


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
    use_dist = world > 1 or os.environ.get("MBAVO_BENCH_FORCE_DIST") == "1"  # the env switch runs the N > 1 code on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import mba_vo_amd as M
    from mba_vo_amd import shard
    if not os.path.exists(M.LIB_PATH):
        raise SystemExit("libmbavo.so missing: run python -c 'import __graft_entry__ as g; g.build()'")
    dev = "cuda:%d" % local_rank
    stream = torch.cuda.current_stream()
    ctx = M.capi.Context(local_rank, stream=stream.cuda_stream)
    rccl_ranks = 0
    if use_dist:
        rccl_ranks = shard.comm_init(ctx, rank, world, shard.torch_bcast(dev))

    def sync():
        # torch.cuda.synchronize() is the contract's bracket; the stream is polled first because the runtime's blocking wait
        # wakes up ~20 us after the last kernel retires -- 2.5 % of a 20-step region (measured: --steps 20 vs --steps 300)
        if not args.no_spin_sync:
            while not stream.query():
                pass
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    run = Runner(M, ctx, args.workload, dev, rank, world, use_dist, args.grad_fp16)

    for _ in range(args.warmup):
        run.step()
    sync()
    ctx.lib.mbavo_profile(ctx.handle, args.time_every)  # HIP-event pair on the dominant kernel's dispatch, every n-th step
    regions, total = [], 0.0
    while not regions or (total < args.min_seconds and len(regions) < args.max_repeats):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.step()
        sync()
        dt = max_over_ranks(time.perf_counter() - t0)  # the same number on every rank: all ranks repeat equally often
        regions.append(dt)
        total += dt
    fused_ms, nlaunch = np.zeros(1), np.zeros(1, np.int32)
    M.capi.check(ctx.lib.mbavo_profile_read(ctx.handle, M.capi.dp(fused_ms), M.capi.ip(nlaunch)), "mbavo_profile_read")
    ctx.lib.mbavo_profile(ctx.handle, 0)
    elapsed = statistics.median(regions)
    kernel = kernel_name(ctx)

    # N > 1: the reduced normal equations against a single-GPU evaluation of the whole joint problem (outside the timing)
    reduction = None
    if run.se is not None:
        run.se.step(True)
        torch.cuda.synchronize()
        got = run.se.reduced.clone()
        ref = run.se.reference()
        scale = float(ref.abs().max())
        diff = float((got - ref).abs().max()) / (scale if scale > 0 else 1.0)
        reduction = {"object": "merged [cost | g | H] systems" if run.mode == "frames" else "packed frame blocks",
                     "doubles": int(run.se.count), "max_rel_diff_vs_single_gpu": diff, "ok": bool(diff <= 1e-12),
                     "sharding": run.mode}

    counts = run.local_counts()
    ps_rank = sum(px * S for px, S, _ in counts)
    ps_launched = sum(p.pixel_samples for p in run.probs) / (world if run.se is not None else 1)
    ps_all = sum_over_ranks(ps_rank)

    out = None
    if rank == 0:
        k_ms = float(fused_ms[0]) / max(int(nlaunch[0]), 1)
        flops, nbytes, ach_tf, ach_gbs = run.figures(counts, k_ms)
        traffic, traffic_src = measured_hbm_traffic(args.workload, kernel)
        exe, exe_src = executed_fp64_flops(args.workload, kernel)
        per_step = [r / args.steps * 1e3 for r in regions]
        out = {
            "metric": "Mpixel-samples/s per GN iteration (640x480, 4-lvl pyr, 8 blur samples)" if args.workload.startswith("c2")
                      else "Mpixel-samples/s per GN iteration (%s)" % args.workload,
            "value": round(ps_all * args.steps / elapsed / 1e6, 3), "unit": "Mpixel-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong" if (run.se is not None and run.mode == "keypoints") else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": len(regions), "ms_per_step_min_max": [round(min(per_step), 5), round(max(per_step), 5)],
            "config": {"workload": run.desc, "name": args.workload, "problems_per_rank": len(run.probs),
                       "pixel_samples_per_step_per_rank": ps_rank, "pixel_samples_launched_per_rank": ps_launched,
                       "parallelism": ("joint problem sharded by %s over %d rank(s): evaluation -> %sONE mbavo_allreduce_blocks "
                                       "(RCCL, context's own communicator) of %d doubles per step"
                                       % (run.mode, world, "device merge -> " if run.mode == "frames" else "", run.se.count))
                       if run.se is not None else "1 GPU"},
            "roofline": {"bound": "fp64", "achieved": round(ach_tf, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tf / FP64_PEAK_TFLOPS, 5),
                         "frac_executed": round(exe / (k_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5) if exe and k_ms > 0 else None,
                         "executed_fp64_flops_per_launch": exe, "executed_source": exe_src,
                         "issue_busy_frac": issue_busy_fraction(args.workload, kernel, k_ms),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel, "kernel_ms": round(k_ms, 6), "launches_timed": int(nlaunch[0]),
                         "algorithmic_flops_per_launch": flops,
                         "step_frac": round(flops / (elapsed / args.steps) / 1e12 / FP64_PEAK_TFLOPS, 5) if elapsed > 0 else None,
                         "launches_per_step": launches_per_step(kernel),
                         "note": "binding roofline = the FP64 pipe (FP64 VALU and f64 MFMA share it; 78.6 TFLOP/s; this is "
                                 "the contract's 'mfma' bound): intensity ~150 flop/B >> 9.8 flop/B balance.  frac counts flops "
                                 "as the reference source writes them (SURVEY.md 8d) and can exceed 1 because the kernel "
                                 "applies CSE; frac_executed counts the FP64 flops the kernel issues (SQ counters) and cannot; "
                                 "issue_busy_frac = share of SIMD issue cycles taken by ANY vector / matrix instruction"},
            "roofline_hbm": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": nbytes,
                             "traffic": traffic,
                             "note": "compulsory bytes only; compute-bound kernel, low by construction; traffic = "
                                     "(2*FETCH_SIZE + WRITE_SIZE) KiB from the committed TCC counter passes"},
        }
        if use_dist:
            out["rccl_ranks"] = rccl_ranks
            out["reduction_check"] = reduction
    fb_gpu = None
    if rank == 0 and world == 1 and run.se is None:
        fb_gpu = run.dw.frame_blocks.cpu().numpy().reshape(run.dw.nbf, run.dw.E)

    # the other BASELINE configs, bounded (N = 1 only): value, step time, dominant kernel time, both fractions
    if rank == 0 and world == 1 and not use_dist and not args.no_configs:
        cfgs = {}
        todo = [(n, False) for n in WORKLOADS if n != args.workload] + [("c5_1080p", True)]
        for name, half in todo:
            key = name + ("_fp16grad" if half else "")
            try:
                r = Runner(M, ctx, name, dev, 0, 1, False, half)
                for _ in range(5):
                    r.step()
                torch.cuda.synchronize()
                ctx.lib.mbavo_profile(ctx.handle, 4)
                n, t0 = 0, time.perf_counter()
                while n < 40 or (time.perf_counter() - t0 < 0.25 and n < 4000):
                    for _ in range(20):
                        r.step()
                    torch.cuda.synchronize()
                    n += 20
                dt = time.perf_counter() - t0
                ms, nl = np.zeros(1), np.zeros(1, np.int32)
                M.capi.check(ctx.lib.mbavo_profile_read(ctx.handle, M.capi.dp(ms), M.capi.ip(nl)), "mbavo_profile_read")
                ctx.lib.mbavo_profile(ctx.handle, 0)
                kname = kernel_name(ctx)
                c = r.local_counts()
                kms = float(ms[0]) / max(int(nl[0]), 1)
                fl, nb, tf, gbs = r.figures(c, kms)
                ex, _ = executed_fp64_flops(name, kname)
                cfgs[key] = {"workload": r.desc, "value": round(sum(px * S for px, S, _ in c) * n / dt / 1e6, 3),
                             "unit": "Mpixel-samples/s", "steps": n, "ms_per_step": round(dt / n * 1e3, 5), "kernel": kname,
                             "kernel_ms": round(kms, 6), "frac": round(tf / FP64_PEAK_TFLOPS, 5),
                             "frac_executed": round(ex / (kms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5) if ex and kms > 0 else None,
                             "hbm_frac_algorithmic": round(gbs / HBM_PEAK_GBS, 6)}
                del r
                torch.cuda.empty_cache()
            except Exception as e:  # a failing side config must not cost the headline line
                cfgs[key] = {"error": repr(e)}
        # the caller of the path: BlurAwareDirectTracker::trackFrame on a GPU-rendered blurred sequence, reference-shaped
        # configuration (blur_aware_direct_tracker.cpp:88-203,544-637); wall time of the mbavo_vo_track_frame calls
        try:
            from mba_vo_amd import sequence
            seq = sequence.make_sequence(ctx, H=480, W=640, M=8, device=dev)
            sequence.track_sequence(ctx, seq)  # warm-up: allocations, code objects
            runs = [sequence.track_sequence(ctx, seq) for _ in range(5)]
            per_frame = sorted(sum(f["seconds"] for f in r) / len(r) for r in runs)
            r0 = runs[0]
            cfgs["trackframe_640x480"] = {
                "workload": "BlurAwareDirectTracker::trackFrame, 640x480, 4 levels, 30-px grid keypoints x 8-pixel pattern, k = 2, "
                            "S = 8, %d frames (GPU-rendered blurred sequence on a textured plane), host-driven LM loop on "
                            "persistent evaluation kernels" % len(r0),
                "ms_per_frame": round(1e3 * per_frame[len(per_frame) // 2], 4), "ms_per_frame_min": round(1e3 * per_frame[0], 4),
                "passes": len(runs), "frames": len(r0), "keyframes": int(sum(f["is_keyframe"] for f in r0)),
                "keypoints_level0": int(r0[0]["K0"]), "lm_trace_records": int(sum(f["num_trace"] for f in r0)),
                "poses_reproducible": bool(all(np.array_equal(a["T"], b["T"]) for r in runs[1:] for a, b in zip(r0, r)))}
        except Exception as e:
            cfgs["trackframe_640x480"] = {"error": repr(e)}
        out["configs"] = cfgs

    if rank == 0 and not args.no_cpu_baseline and world == 1 and fb_gpu is not None:  # rank 0 at N = 1 only
        cb, fb_cpu = cpu_baseline(run.probs, args.cpu_seconds)
        scale = np.abs(fb_cpu).max(axis=1, keepdims=True)
        cb["gpu_vs_cpu_max_rel_diff"] = float((np.abs(fb_gpu - fb_cpu) / scale).max())
        out["cpu_baseline"] = cb
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        ctx.lib.mbavo_comm_destroy(ctx.handle)
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
