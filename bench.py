#!/usr/bin/env python3
"""bench.py -- throughput of the blur-aware tracking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launches its own N ranks; or under torch.distributed.run)

A "step" is one Gauss-Newton iteration of the hot path = one full H/g evaluation (evaluate_cost_hessian_gradient,
ba_tracker/spline_update_step.cpp:97-241) of every problem of the workload, inputs resident in HBM, the merged [cost | g | H]
systems and the packed normal-equation blocks left in HBM.  Default workload = BASELINE.json configs[1]: one 640x480 keyframe
pair, 4-level pyramid, 8 blur samples, 4 control poses (cubic, k = 4), dense mode (every pixel of every level a P=1 patch).

Output: ONE compact JSON line on rank 0 (<= 6 KB: the contract's keys, `roofline`, `cpu_baseline`, a handful of side figures and
parity figures, numbers and short labels only) and a DETAILS file (--details-out, default profiles/bench_details_last.json, its
path in the line) with everything else: every side config, per-rank timings, both collectives' runs, the long-horizon report.
What the keys mean is written in README.md ("Reading the bench line"), not in the JSON.

N > 1 (one process per GPU): `python bench.py --gpus N` run WITHOUT a launcher re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; run under a launcher (WORLD_SIZE set)
it is one rank and WORLD_SIZE must equal --gpus.  The workload is sharded over the ranks --
  * c2_dense (default) and the other single-pair workloads: ONE joint problem of N blurred frames against the same keyframe on
    one spline segment, frame r on rank r (weak scaling: the per-GPU work is the N = 1 workload); every rank evaluates its
    frame's packed block into its slice of the result buffer and ONE in-place all-gather of equal slices per step over xGMI
    leaves every frame's block on every rank (--collective allreduce / --shard frames: an all-reduce, as BASELINE.json words it);
  * c4_batch512 / c3_batch64 (independent keyframe pairs): pair b on rank b % N, whole (strong scaling);
through the PRODUCT's collectives (RCCL on the context's communicator, then -- behind a canary child process per rank -- the
one-shot collectives over peer-mapped memory by the same timing procedure; the faster verified one is the line's step, `comm`
says which).  torch.distributed carries the rendezvous, the barrier and the max-over-ranks of the clock.  After the timed region
the reduced normal equations are compared with a single-GPU evaluation of the whole workload (rank 0; 1e-12).
value = pixel-samples of all ranks / max-over-ranks time.

The timed region (exactly K steps between barrier + synchronize) is repeated until >= --min-seconds have been timed and the MEDIAN
region is reported, so that K = 20 does not rest on 1 ms of GPU work.  The side configs, the CPU baseline and the parity legs run
after it (N = 1), bounded.
"""
import argparse
import json
import os
import statistics
import sys
import time

# the CPU baseline's OpenMP threads must SLEEP at the end of their share, not spin: the sandboxed hosts meter CPU time, and
# spinning waiters eat the quota of the threads still working (read by libgomp when it is first loaded)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np

from bench_core import (FP64_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, WORKLOADS, Runner, executed_fp64_flops, issue_busy_fraction,
                        kernel_name, launches_per_step, measured_hbm_traffic, rocprof_kernel_ms, stale_flags)

LINE_LIMIT = 6000  # bytes of the JSON line (the driver keeps an 8 KB tail of stdout; round 5's 26 KB line did not parse)
SHARED_COMMS = ("gloo", "p2p-shared")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2_dense", choices=WORKLOADS)
    ap.add_argument("--details-out", default=os.path.join("profiles", "bench_details_last.json"),
                    help="where rank 0 writes the full report (side configs, per-rank timings, notes); relative to the repo root")
    ap.add_argument("--shard", default=None, choices=["pairs", "keypoints", "frames", "frame_blocks"],
                    help="N > 1 sharding (default: the workload's own -- frames for a single pair, pairs for a batch of pairs)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "gloo", "p2p", "p2p-shared"],
                    help="N > 1 only.  rccl: one GPU per rank, the product's collectives on the context's RCCL communicator.  gloo: the "
                         "ranks SHARE the visible GPU(s) (rank r on GPU r %% device_count) and a gloo collective on a pinned host copy "
                         "stands in for RCCL (shard.HostStagedCollective); its timings are not scaling figures.  p2p: one GPU per rank, "
                         "the product's ONE-SHOT collectives over peer-mapped receive regions (csrc/p2p_comm.hip, no RCCL).  "
                         "p2p-shared: the same collectives between ranks that share the visible GPU(s)")
    ap.add_argument("--collective", default="allgather", choices=["allgather", "allreduce"],
                    help="pair sharding: ONE in-place all-gather of equal slices (default) or, as BASELINE.json words it, ONE "
                         "all-reduce of a send buffer that is zero outside the rank's slice (twice the bytes on the wire)")
    ap.add_argument("--batch-pairs", type=int, default=512, help="pairs of the N > 1 batch configs (tests shrink it)")
    ap.add_argument("--cost-only", action="store_true",
                    help="the cost-only evaluation (evaluate_cost_hessian_gradient with nullptr, nullptr: half of every LM iteration, "
                         "blur_aware_direct_tracker.cpp:863-880) as the timed step; flops_alg = 122 PS + 13 PX (SURVEY.md 8d)")
    ap.add_argument("--spline-k", type=int, default=4, choices=[2, 4], help="spline degree (N = k control poses); 2 is the reference's default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the bounded runs of the other BASELINE configs")
    ap.add_argument("--long-frames", type=int, default=120, help="frames of the long-horizon parity leg (0: skip it)")
    ap.add_argument("--grad-fp16", action="store_true",
                    help="gradient pyramid stored as IEEE half pairs (BASELINE configs[4]: lossless for 8-bit images)")
    ap.add_argument("--packed-keyframes", action="store_true",
                    help="keyframes as one word per pixel: intensity + both central differences (mbavo_problem.grad_fp16 = 2, lossless)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="HIP-event pair around the dominant kernel on every n-th timed step (events cost launch gaps)")
    ap.add_argument("--min-seconds", type=float, default=0.3, help="repeat the K-step region until this much was timed")
    ap.add_argument("--no-spin-sync", action="store_true", help="synchronize without polling the stream first (A/B of the bracket's own cost)")
    ap.add_argument("--max-repeats", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="budget of EACH bounded CPU baseline sample (1 and T threads)")
    ap.add_argument("--p2p-canary", action="store_true",
                    help="(internal) child process of an N > 1 run: sets the one-shot p2p collectives up between the ranks' GPUs and checks an "
                         "all-reduce and an all-gather, so that a platform that faults on peer-mapped memory costs a child, not the bench line")
    return ap.parse_args()


def fail_line(msg, **kw):
    """one-line JSON error object on stdout, exit code 2 (a launcher or driver can parse why there is no measurement)"""
    print(json.dumps(dict({"error": msg, "metric": None, "value": None}, **kw)), flush=True)
    raise SystemExit(2)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: this process becomes the launcher of N ranks of itself (one per GPU, or N
    ranks on the visible GPU(s) for the shared-GPU comm modes).  Rank 0's JSON line is the children's stdout, i.e. ours."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        fail_line("bench.py needs an MI355X: the HIP path has no CPU fallback", n_gpus_requested=args.gpus, gpus_visible=0)
    if have < args.gpus and args.comm not in SHARED_COMMS:
        fail_line("--gpus %d needs %d visible GPUs, %d found (--comm p2p-shared / gloo run the N > 1 path with ranks sharing a GPU)"
                  % (args.gpus, args.gpus, have), n_gpus_requested=args.gpus, gpus_visible=have)
    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and the p2p regions need it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.run(cmd, env=env, cwd=ROOT).returncode)


def short(s, n=60):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def compact_line(full, details_path):
    """The line the driver parses: the contract's keys with numbers and <= 60-character labels; everything else stays in `full`
    (the DETAILS file).  Six side figures and the parity figures ride along."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "repeats", "ms_per_step_min_max", "ms_per_step_incl_d2h", "ms_per_step_to_pinned_host", "to_pinned_host_same_bits")
    line = {k: (short(full[k], 100) if isinstance(full[k], str) else full[k]) for k in keys if k in full}
    c = full["config"]
    line["config"] = {"workload": c["name"], "shape": short(c["shape"], 100), "problems_per_rank": c["problems_per_rank"],
                      "pixel_samples_per_step_per_rank": c["pixel_samples_per_step_per_rank"], "parallelism": short(c["parallelism_short"], 100)}
    for name in ("roofline", "roofline_fp64", "roofline_hbm"):
        r = full.get(name)
        if not r:
            continue
        keep = ("bound", "pipe", "achieved", "peak", "unit", "frac", "frac_is", "frac_executed", "frac_executed_is", "frac_of_traffic", "frac_upper",
                "issue_busy_frac", "traffic", "kernel", "kernel_ms", "launches_timed", "kernel_ms_rocprofv3", "frac_rocprofv3", "rocprofv3_stale", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch",
                "executed_fp64_flops_per_launch", "step_frac", "launches_per_step", "counters_from", "stale", "kernel_source_sha")
        line[name] = {k: (short(r[k]) if isinstance(r[k], str) else r[k]) for k in keep if k in r and r[k] is not None or k == "traffic" and k in r}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample_short", "single_thread_value", "host_logical_cpus",
                                                   "gpu_vs_cpu_max_rel_diff") if k in cb}
        line["cpu_baseline"]["sample"] = line["cpu_baseline"].pop("sample_short")
    cfgs = full.get("configs") or {}

    def pick(name, key):
        v = cfgs.get(name)
        return v.get(key) if isinstance(v, dict) and "error" not in v else None
    side = {}
    if full["n_gpus"] == 1 and cfgs:
        lm64, lm512 = cfgs.get("lm_batch64") or {}, cfgs.get("lm_batch512") or {}
        side = {"trackframe_ms_per_frame": pick("trackframe_640x480", "ms_per_frame"),
                "lm_batch64_us_per_round": (lm64.get("device_svd") or {}).get("us_per_round"),
                "lm_batch512_us_per_round": (lm512.get("device_svd") or {}).get("us_per_round"),
                "c2_semidense_ms_per_step": pick("c2_semidense", "ms_per_step"),
                "c2_dense_sequential_ms_per_step": pick("c2_dense_sequential", "ms_per_step"),
                "c2_dense_cost_only_ms_per_step": pick("c2_dense_cost_only", "ms_per_step"),
                "c1_dense_frac": pick("c1_dense", "frac"), "c3_batch64_ms_per_step": pick("c3_batch64", "ms_per_step"),
                "c4_batch512_ms_per_step": pick("c4_batch512", "ms_per_step"), "c4_batch512_packed_ms_per_step": pick("c4_batch512_packed", "ms_per_step"),
                "c5_1080p_frac": pick("c5_1080p", "frac"), "c5_1080p_fp16grad_frac": pick("c5_1080p_fp16grad", "frac")}
    elif cfgs:
        side = {"c4_batch512_pairs_value": pick("c4_batch512_pairs", "value"), "c4_batch512_pairs_ms_per_step": pick("c4_batch512_pairs", "ms_per_step"),
                "c4_batch512_pairs_packed_value": pick("c4_batch512_pairs_packed", "value"),
                "c4_batch512_pairs_weak_packed_value": pick("c4_batch512_pairs_weak_packed", "value"),
                "c2_dense_frames_allreduce_of_systems_ms_per_step": pick("c2_dense_frames_allreduce_of_systems", "ms_per_step"),
                "lm_batch512_pairs_iterations_per_s": pick("lm_batch512_pairs", "value"), "lm_batch_pairs_weak_iterations_per_s": pick("lm_batch_pairs_weak", "value"),
                "configs_with_errors": sorted(k for k, v in cfgs.items() if isinstance(v, dict) and "error" in v)}
    if side:
        line["side"] = side
    if full.get("parity"):
        line["parity"] = full["parity"]
    if full["n_gpus"] > 1 or "reduction_check" in full:
        line["comm"] = short(full.get("comm"), 80)
        line["rccl_ranks"] = full.get("rccl_ranks")
        rc = full.get("reduction_check") or {}
        line["reduction_check"] = {"ok": rc.get("ok"), "max_rel_diff_vs_single_gpu": rc.get("max_rel_diff_vs_single_gpu"), "doubles": rc.get("doubles"),
                                   "sharding": rc.get("sharding"), "collective": short(rc.get("collective"), 80)}
        line["per_rank"] = {k: [round(v, 4) for v in full["per_rank"][k]] for k in ("kernel_ms", "local_evaluation_ms", "collective_ms")}
        for k in ("comm_profile_p2p", "comm_profile_rccl", "comm_profile_standin"):
            p = full.get(k)
            if isinstance(p, dict):
                line[k] = {kk: (short(p[kk], 120) if isinstance(p[kk], str) else p[kk]) for kk in ("ms_per_step", "value", "selected_as_the_step", "skipped", "error") if kk in p}
                if isinstance(p.get("reduction_check"), dict):
                    line[k]["ok"] = p["reduction_check"].get("ok")
    line["details"] = details_path
    return line


def main():
    """One rank of the bench.  Whatever goes wrong after the launch checks, rank 0 still prints ONE parseable line saying so (a
    driver that finds no line cannot tell a crash from a hang), then the exception propagates."""
    state = {"emit": None, "rank": int(os.environ.get("RANK", "0")), "printed": False}
    try:
        run(state)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 (reported, then re-raised)
        if state["emit"] is not None and state["rank"] == 0 and not state["printed"]:
            import traceback
            where = traceback.extract_tb(e.__traceback__)[-1]
            state["emit"]({"error": short(repr(e), 300), "where": "%s:%d" % (os.path.basename(where.filename), where.lineno), "metric": None, "value": None})
        raise


def run(state):
    args = parse()
    if args.p2p_canary:
        import bench_side
        raise SystemExit(bench_side.p2p_canary_child())
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and os.environ.get("MBAVO_BENCH_FORCE_DIST") != "1":
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and os.environ.get("MBAVO_BENCH_FORCE_DIST") != "1":
        if rank == 0:
            fail_line("WORLD_SIZE %d != --gpus %d: launch N ranks for --gpus N (or run `python bench.py --gpus N` without a launcher)" % (world, args.gpus))
        raise SystemExit(2)
    # The one JSON line goes to the REAL stdout; everything else written to file descriptor 1 by this process or by the
    # libraries it loads goes to stderr.  RCCL prints a version banner to C stdout when the first communicator is
    # created, block-buffered when stdout is a pipe and flushed only at exit -- i.e. AFTER the JSON line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())
    state["emit"], state["rank"] = emit, rank

    if not torch.cuda.is_available():
        if rank == 0:
            emit({"error": "bench.py needs an MI355X: the HIP path has no CPU fallback", "metric": None, "value": None})
        raise SystemExit(2)
    shared_gpu = args.comm in SHARED_COMMS  # the ranks share the visible GPU(s); gloo is the process group (see --comm)
    use_p2p = args.comm in ("p2p", "p2p-shared")
    if world > 1 and not shared_gpu and torch.cuda.device_count() < world:
        if rank == 0:
            emit({"error": "%d ranks but %d visible GPU(s) and --comm %s is not a shared-GPU mode" % (world, torch.cuda.device_count(), args.comm),
                  "metric": None, "value": None})
        raise SystemExit(2)
    if shared_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("MBAVO_BENCH_FORCE_DIST") == "1"  # the env switch runs the N > 1 code on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if shared_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import mba_vo_amd as M
    from mba_vo_amd import shard
    if not os.path.exists(M.LIB_PATH):
        raise SystemExit("libmbavo.so missing: run python -c 'import __graft_entry__ as g; g.build()'")
    dev = "cuda:%d" % local_rank
    stream = torch.cuda.current_stream()
    ctx = M.capi.Context(local_rank, stream=stream.cuda_stream)
    rccl_ranks, coll = 0, None
    if use_dist and use_p2p:
        coll = shard.P2PCollective(ctx, rank, world, max_doubles=1 << 18)
        rccl_ranks = world
    elif use_dist and shared_gpu:
        coll = shard.HostStagedCollective(ctx, rank, world)
    elif use_dist:
        rccl_ranks = shard.comm_init(ctx, rank, world, shard.torch_bcast(dev))
        coll = shard.RcclCollective(ctx)
    hdev = "cpu" if shared_gpu else dev  # where the helper collectives below keep their scalars (gloo: host tensors)

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
        t = torch.tensor([x], dtype=torch.float64, device=hdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=hdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def per_rank(x):
        """the value of every rank, in rank order (a list on every rank)"""
        if not use_dist:
            return [float(x)]
        mine = torch.tensor([x], dtype=torch.float64, device=hdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        return [float(t.item()) for t in every]

    def reduction_check(run):
        """the reduced normal equations against a single-GPU evaluation of the whole workload (outside the timing)"""
        run.se.step(True)
        torch.cuda.synchronize()
        got = run.se.reduced.clone()
        ref = run.se.reference()
        scale = float(ref.abs().max())
        diff = float((got - ref).abs().max()) / (scale if scale > 0 else 1.0)
        obj = {"frames": "merged [cost | g | H] systems", "frame_blocks": "packed frame blocks (every rank's frames in its slice, rank-major)",
               "keypoints": "packed frame blocks (partial sums over the ranks' keypoint bands)",
               "pairs": "packed frame blocks (disjoint slices, rank-major)"}[run.mode]
        return {"object": obj, "doubles": int(run.se.count), "max_rel_diff_vs_single_gpu": diff,
                "ok": bool(diff <= 1e-12), "bit_exact": bool(torch.equal(got, ref)), "sharding": run.mode,
                "collective": collective_name(run)}

    def collective_name(r):
        call = ("mbavo_allgather_blocks" if r.se.pair_collective == "allgather" else "mbavo_allreduce_blocks_to") if r.mode == "pairs" \
            else (("mbavo_allgather_blocks" if r.se.fb_allgather else "mbavo_allreduce_blocks_to") if r.mode == "frame_blocks" else "mbavo_allreduce_blocks")
        if use_p2p:
            name = {"mbavo_allgather_blocks": "mbavo_allgather_blocks_p2p", "mbavo_allreduce_blocks_to": "copy + mbavo_allreduce_blocks_p2p",
                    "mbavo_allreduce_blocks": "mbavo_allreduce_blocks_p2p"}[call]
            return name + " [one-shot over peer-mapped regions, no RCCL%s]" % (": ranks share one GPU" if shared_gpu else "")
        return call + (" [RCCL]" if not shared_gpu else " -> STAND-IN: gloo on a pinned host copy (ranks share one GPU)")

    def comm_profile(run, n=40):
        """Per-rank duration of the collective alone (events around it, every rank's own evaluation before it: the figure
        includes the wait for the slowest rank's evaluation) and of the local evaluation + merge alone."""
        se = run.se
        for _ in range(3):
            se.step(True)
        sync()
        ev = []
        for _ in range(n):
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record(stream)
            se.evaluate_local(True)
            b.record(stream)
            se.reduce()
            c.record(stream)
            ev.append((a, b, c))
        sync()
        loc = statistics.median(a.elapsed_time(b) for a, b, c in ev)
        red = statistics.median(b.elapsed_time(c) for a, b, c in ev)
        return loc, red

    fmt = 2 if args.packed_keyframes else int(args.grad_fp16)
    pairs = args.batch_pairs if args.batch_pairs != 512 else None
    run = Runner(M, ctx, args.workload, dev, rank, world, use_dist, fmt, shard_mode=args.shard, coll=coll, pair_collective=args.collective,
                 pairs=pairs, cost_only=args.cost_only, k=args.spline_k)
    # key of the committed counter extracts: the workload plus the variant of the evaluation
    wkey = args.workload + ("_k2" if args.spline_k == 2 else "") + ("_cost_only" if args.cost_only else "")

    def time_regions(r):
        """W warm-up steps, then the K-step region between barrier + synchronize, repeated until min_seconds were timed:
        (regions, their median, the dominant kernel's mean duration, its timed launches)"""
        for _ in range(args.warmup):
            r.step()
        sync()
        ctx.lib.mbavo_profile(ctx.handle, args.time_every)  # HIP-event pair on the dominant kernel's dispatch, every n-th step
        regs, total = [], 0.0
        while not regs or (total < args.min_seconds and len(regs) < args.max_repeats):
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                r.step()
            sync()
            dt = max_over_ranks(time.perf_counter() - t0)  # the same number on every rank: all ranks repeat equally often
            regs.append(dt)
            total += dt
        ms, nl = np.zeros(1), np.zeros(1, np.int32)
        M.capi.check(ctx.lib.mbavo_profile_read(ctx.handle, M.capi.dp(ms), M.capi.ip(nl)), "mbavo_profile_read")
        ctx.lib.mbavo_profile(ctx.handle, 0)
        return regs, statistics.median(regs), float(ms[0]) / max(int(nl[0]), 1), nl

    regions, elapsed, k_ms, nlaunch = time_regions(run)
    kernel = kernel_name(ctx)

    reduction = reduction_check(run) if run.se is not None else None
    k_ms_ranks = per_rank(k_ms)
    comm_ms = None
    if run.se is not None:
        loc, red = comm_profile(run)
        comm_ms = (per_rank(loc), per_rank(red))

    # the D2H-inclusive step: the packed blocks copied to pinned host memory after every evaluation, which is
    # what a host-side LM consumer of the blocks waits for (N = 1 only; never `value`)
    d2h_ms = to_host_ms = to_host_same = None
    if run.se is None:
        host = torch.empty(run.dw.frame_blocks.shape, dtype=torch.float64).pin_memory()
        for _ in range(3):
            run.step()
            host.copy_(run.dw.frame_blocks, non_blocking=True)
        torch.cuda.synchronize()
        n_d2h = max(20, min(args.steps, 200))
        t0 = time.perf_counter()
        for _ in range(n_d2h):
            run.step()
            host.copy_(run.dw.frame_blocks, non_blocking=True)
            while not stream.query():
                pass
        torch.cuda.synchronize()
        d2h_ms = (time.perf_counter() - t0) / n_d2h * 1e3
        # ... and without the copy: the finalize step stores the packed blocks, the merged systems and the valid counts straight into
        # PINNED HOST memory (the C ABI takes any device-accessible pointer; this is how the library's own LM loop gets its results),
        # the host polls the stream -- what a host-side consumer pays when it hands the library host buffers (never `value`)
        try:
            if args.cost_only or getattr(run.dw, "systems", None) is None:
                raise RuntimeError("merged H/g steps only")
            h_fb = torch.empty(run.dw.frame_blocks.shape, dtype=torch.float64).pin_memory()
            h_sys = torch.empty(run.dw.systems.shape, dtype=torch.float64).pin_memory()
            h_valid = torch.empty(run.dw.valid.shape, dtype=torch.float64).pin_memory()

            def host_step():
                M.capi.check(ctx.lib.mbavo_eval_batch_merged(ctx.handle, run.dw.B, run.dw.array, run.dw.k, h_fb.data_ptr(), h_sys.data_ptr(), None,
                                                             h_valid.data_ptr()), "mbavo_eval_batch_merged")
            for _ in range(3):
                host_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_d2h):
                host_step()
                while not stream.query():
                    pass
            to_host_ms = (time.perf_counter() - t0) / n_d2h * 1e3
            run.step()
            torch.cuda.synchronize()
            to_host_same = bool(torch.equal(h_fb, run.dw.frame_blocks.cpu()) and torch.equal(h_sys, run.dw.systems.cpu()))
        except Exception as e:  # (a side figure must not cost the line)
            to_host_ms, to_host_same = None, repr(e)

    counts = run.local_counts()
    ps_rank = sum(px * S for px, S, _ in counts)
    ps_launched = sum(p.pixel_samples for p in run.probs) / (world if run.se is not None else 1)
    ps_all = sum_over_ranks(ps_rank)

    def parallelism(r):
        return ("workload sharded by %s over %d rank(s): evaluation -> %sONE %s of %d doubles per step"
                % (r.mode, world, "device merge -> " if r.mode == "frames" else "", collective_name(r), r.se.count)) if r.se is not None else "1 GPU"

    def parallelism_short(r):
        return "%s over %d ranks, 1 %s / step" % (r.mode, world, collective_name(r).split(" ")[0] if "copy" not in collective_name(r) else "allreduce_blocks_p2p") \
            if r.se is not None else "1 GPU"

    out = None
    if rank == 0:
        flops, nbytes, ach_tf, ach_gbs = run.figures(counts, k_ms)
        traffic, traffic_src = measured_hbm_traffic(wkey, kernel)
        exe, exe_src = executed_fp64_flops(wkey, kernel)
        sha_now, stale = stale_flags([("hbm_counters", wkey), ("pmc_fp64", wkey), ("pmc_sq", wkey)])
        per_step = [r / args.steps * 1e3 for r in regions]
        rp_ms, rp_src, rp_stale = rocprof_kernel_ms(kernel) if wkey == "c2_dense" else (None, None, None)
        out = {
            "metric": "Mpixel-samples/s per GN iteration (640x480, 4-lvl pyr, 8 blur samples)" if wkey.startswith("c2") and wkey == args.workload
                      else "Mpixel-samples/s per GN iteration (%s)" % wkey,
            "value": round(ps_all * args.steps / elapsed / 1e6, 3), "unit": "Mpixel-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong" if (run.se is not None and run.mode in ("keypoints", "pairs")) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": len(regions), "ms_per_step_min_max": [round(min(per_step), 5), round(max(per_step), 5)],
            "config": {"workload": run.desc, "name": args.workload, "shape": run.shape, "problems_per_rank": len(run.probs) if run.se is None else run.se.n_live,
                       "pixel_samples_per_step_per_rank": ps_rank, "pixel_samples_launched_per_rank": ps_launched,
                       "parallelism": parallelism(run), "parallelism_short": parallelism_short(run)},
            "roofline": {"bound": "mfma", "pipe": "fp64: FP64 VALU and v_mfma_f64 share one 78.6 TF pipe",
                         "achieved": round(ach_tf, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tf / FP64_PEAK_TFLOPS, 5), "frac_is": "reference-flop equivalent (SURVEY 8d count); can pass 1",
                         "frac_executed": round(exe / (k_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5) if exe and k_ms > 0 else None,
                         "frac_executed_is": "FP64 flops issued (SQ counters) / duration / peak",
                         "executed_fp64_flops_per_launch": exe, "executed_source": exe_src,
                         "issue_busy_frac": issue_busy_fraction(wkey, kernel, k_ms),
                         "traffic": traffic, "traffic_source": traffic_src, "counters_from": "committed extracts: " + ", ".join(sorted(stale)) if stale else None,
                         "kernel_source_sha": sha_now, "counter_extracts_stale": stale, "stale": bool(any(stale.values())) if stale else None,
                         "kernel": kernel, "kernel_ms": round(k_ms, 6), "launches_timed": int(nlaunch[0]),
                         # the committed rocprofv3 summary of the same command beside the live event timing (an event pair lengthens the
                         # launch it is attached to by ~2 us, so the live figure -- the one frac uses -- errs low)
                         "kernel_ms_rocprofv3": None if rp_ms is None else round(rp_ms, 6), "rocprofv3_source": rp_src, "rocprofv3_stale": rp_stale,
                         "frac_rocprofv3": None if not rp_ms else round(flops / (rp_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5),
                         "algorithmic_flops_per_launch": flops,
                         "step_frac": round(flops / (elapsed / args.steps) / 1e12 / FP64_PEAK_TFLOPS, 5) if elapsed > 0 else None,
                         "launches_per_step": launches_per_step(kernel)},
            "roofline_hbm": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": nbytes, "traffic": traffic},
        }
        if args.workload in ("c3_batch64", "c4_batch512"):
            # pairs with their own images: the tap gather binds (no-taps ablation: 47 % of the 512-pair kernel, profiles/
            # r03_kfused_experiments.txt 3.), so the HBM figure leads and the FP64 one rides along
            out["roofline_fp64"] = out["roofline"]
            h = out.pop("roofline_hbm")
            h.update(kernel=kernel, kernel_ms=round(k_ms, 6), launches_timed=int(nlaunch[0]), traffic_source=traffic_src,
                     frac_of_traffic=round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and k_ms > 0 else None,
                     frac_upper=round(run.nbytes_upper / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if k_ms > 0 else None,
                     algorithmic_bytes_upper_per_launch=run.nbytes_upper, distinct_taps=run.distinct_summary())
            out["roofline"] = h
        if d2h_ms is not None:
            out["ms_per_step_incl_d2h"] = round(d2h_ms, 5)
        if to_host_ms is not None:
            out["ms_per_step_to_pinned_host"] = round(to_host_ms, 5)
        if to_host_same is not None:
            out["to_pinned_host_same_bits"] = to_host_same
        if use_dist:
            out["rccl_ranks"] = rccl_ranks
            out["comm"] = ("p2p one-shot collectives" + (", %d ranks on %d GPU(s): NOT a scaling measurement" % (world, torch.cuda.device_count()) if shared_gpu else "")) if use_p2p \
                else "rccl" if not shared_gpu else "gloo stand-in, %d ranks on %d GPU(s): NOT a scaling measurement" % (world, torch.cuda.device_count())
            out["reduction_check"] = reduction
            out["per_rank"] = {"kernel_ms": [round(v, 6) for v in k_ms_ranks],
                               "local_evaluation_ms": [round(v, 6) for v in comm_ms[0]],
                               "collective_ms": [round(v, 6) for v in comm_ms[1]]}
    # Both collectives in the N > 1 line.  With RCCL as the step's collective above, the SAME sharded evaluation once more through
    # the product's one-shot p2p collectives, timed by the same procedure -- behind a CANARY: the set-up and a few collectives
    # first run in a child process per rank, so that a platform that faults on peer-mapped memory costs a child and not this line.
    # Where the p2p run is verified (reduction check) and faster, IT is the line's step: value, ms_per_step and the collective
    # named in config.parallelism are its own, RCCL's figures move to comm_profile_rccl.
    # (--comm gloo, ranks sharing a GPU, runs the same selection with the gloo stand-in in RCCL's place: the mechanics on one GPU.)
    if use_dist and not use_p2p and world > 1 and os.environ.get("MBAVO_BENCH_P2P", "1") != "0":
        import bench_side
        p2p_line = None
        ok_mine, why = bench_side.run_p2p_canary(shared_gpu)
        canary = per_rank(1.0 if ok_mine else 0.0)
        if min(canary) < 1.0:
            p2p_line = {"skipped": "the canary child failed on rank(s) %s (rank %d: %s)" % ([i for i, v in enumerate(canary) if v < 1.0], rank, why)}
        else:
            try:
                # (P2PCollective raises on EVERY rank if any rank's region cannot be created or mapped: the ranks stay in step)
                c2 = shard.P2PCollective(ctx, rank, world, max_doubles=max(int(run.se.count), 1 << 12))
                r2 = Runner(M, ctx, args.workload, dev, rank, world, True, fmt, shard_mode=args.shard, coll=c2, pair_collective=args.collective,
                            pairs=pairs, cost_only=args.cost_only, k=args.spline_k)
                use_p2p = True  # (collective_name / reduction_check label what they describe)
                loc2, red2 = comm_profile(r2)
                chk2 = reduction_check(r2)
                regions2, elapsed2, k_ms2, nl2 = time_regions(r2)
                lr2, rr2, kr2 = per_rank(loc2), per_rank(red2), per_rank(k_ms2)
                par2, par2s = parallelism(r2), parallelism_short(r2)
                use_p2p = False
                p2p_line = {"collective": chk2["collective"], "ms_per_step": round(elapsed2 / args.steps * 1e3, 5), "steps": args.steps,
                            "repeats": len(regions2),
                            "per_rank": {"kernel_ms": [round(v, 6) for v in kr2], "local_evaluation_ms": [round(v, 6) for v in lr2],
                                         "collective_ms": [round(v, 6) for v in rr2]},
                            "reduction_check": chk2}
                faster = bool(chk2["ok"]) and elapsed2 < elapsed  # (elapsed: max over the ranks -> the same decision on every rank)
                if faster and rank == 0:
                    out["comm_profile_rccl" if not shared_gpu else "comm_profile_standin"] = {
                        "collective": out["reduction_check"]["collective"], "ms_per_step": out["ms_per_step"], "value": out["value"],
                        "per_rank": out["per_rank"], "reduction_check": out["reduction_check"]}
                    per2 = [r_ / args.steps * 1e3 for r_ in regions2]
                    out.update({"value": round(ps_all * args.steps / elapsed2 / 1e6, 3), "ms_per_step": round(elapsed2 / args.steps * 1e3, 5),
                                "repeats": len(regions2), "ms_per_step_min_max": [round(min(per2), 5), round(max(per2), 5)],
                                "reduction_check": chk2})
                    out["per_rank"] = dict(out["per_rank"], **p2p_line["per_rank"])
                    out["config"]["parallelism"], out["config"]["parallelism_short"] = par2, par2s
                    out["comm"] = "p2p one-shot collectives (selected: verified against the single-GPU evaluation and faster than %s, " \
                                  "whose run of the same step is in comm_profile_%s)" % (("RCCL", "rccl") if not shared_gpu else ("the gloo stand-in", "standin"))
                    out["roofline"]["step_frac"] = round(out["roofline"]["algorithmic_flops_per_launch"] / (elapsed2 / args.steps) / 1e12 / FP64_PEAK_TFLOPS, 5) \
                        if "algorithmic_flops_per_launch" in out["roofline"] else out["roofline"].get("step_frac")
                p2p_line["selected_as_the_step"] = faster
                c2.close()
                del r2
            except Exception as e:
                use_p2p = False
                p2p_line = {"error": repr(e)}
        if rank == 0:
            out["comm_profile_p2p"] = p2p_line
    fb_gpu = None
    if rank == 0 and world == 1 and run.se is None:
        run.step()
        torch.cuda.synchronize()
        fb_gpu = run.dw.frame_blocks.cpu().numpy().reshape(run.dw.nbf, run.dw.E)

    class Env:  # what the side legs need of this run
        pass
    env = Env()
    env.M, env.ctx, env.dev, env.rank, env.world, env.args, env.coll, env.use_dist = M, ctx, dev, rank, world, args, coll, use_dist
    env.sync, env.max_over_ranks, env.sum_over_ranks, env.per_rank = sync, max_over_ranks, sum_over_ranks, per_rank
    env.comm_profile, env.reduction_check, env.collective_name = comm_profile, reduction_check, collective_name

    track = None
    if use_dist and not args.no_configs and args.workload == "c2_dense":
        import bench_side
        cfgs = bench_side.sharded_configs(env)  # (every rank: the configs run collectives)
        if rank == 0:
            out["configs"] = cfgs
    if rank == 0 and world == 1 and not use_dist and not args.no_configs:
        import bench_side
        out["configs"], track = bench_side.single_gpu_configs(env)

    if rank == 0 and not args.no_cpu_baseline and world == 1 and fb_gpu is not None:  # rank 0 at N = 1 only, behind every timing
        import bench_checks  # the ONLY place the CPU checker (oracle/) enters: a reported baseline and the parity figures
        host_probs = run.probs if not hasattr(run.dw, "host_problem") else [run.dw.host_problem(b) for b in range(run.dw.B)]
        out["cpu_baseline"], out["parity"] = bench_checks.run(ctx, host_probs, fb_gpu, track, args.cpu_seconds, args.long_frames)
    if rank == 0:
        details = args.details_out if os.path.isabs(args.details_out) else os.path.join(ROOT, args.details_out)
        try:
            os.makedirs(os.path.dirname(details), exist_ok=True)
            with open(details, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
            shown = args.details_out
        except OSError as e:
            shown = "not written: %r" % (e,)
        line = compact_line(out, shown)
        text = json.dumps(line)
        if len(text) > LINE_LIMIT:  # never again a line the driver cannot parse: drop the optional blocks, largest first
            for k in ("side", "comm_profile_standin", "comm_profile_rccl", "comm_profile_p2p", "parity", "roofline_fp64", "roofline_hbm", "per_rank"):
                line.pop(k, None)
                text = json.dumps(line)
                if len(text) <= LINE_LIMIT:
                    break
        sys.stdout.flush()
        os.write(real_stdout, (text + "\n").encode())
        state["printed"] = True
    if use_dist:
        dist.barrier()
        ctx.lib.mbavo_comm_destroy(ctx.handle)
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
