#!/usr/bin/env python3
"""bench.py -- throughput of the blur-aware tracking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one Gauss-Newton iteration of the hot path = one full H/g evaluation
(evaluate_cost_hessian_gradient, ba_tracker/spline_update_step.cpp:97-241) of every problem of the workload, inputs
resident in HBM, outputs (packed normal-equation blocks) left in HBM.  Default workload = BASELINE.json configs[1]: one
640x480 keyframe pair, 4-level pyramid, 8 blur samples, 4 control poses (cubic, k = 4), dense mode (every pixel of every
level a P=1 patch).

N > 1 (one process per GPU, launched by torch.distributed.run): the workload is sharded over the ranks --
  * c2_dense (default) and the other single-pair workloads: ONE joint problem of N blurred frames against the same
    keyframe on one spline segment, frame r on rank r (weak scaling: the per-GPU work is the N = 1 workload); every rank
    evaluates its frame's packed block straight into its slice of the result buffer and ONE in-place all-gather of equal
    slices per step over xGMI leaves every frame's block on every rank (--collective allreduce: an out-of-place all-reduce
    of a send buffer that is zero elsewhere, as BASELINE.json words it; --shard frames: the 6N x 6N systems merged on the
    device and summed by an all-reduce);
  * c4_batch512 / c3_batch64 (independent keyframe pairs): pair b on rank b % N, whole (--shard pairs, the default;
    strong scaling): every rank evaluates its pairs into its slice of a zero B x E send buffer and ONE out-of-place
    all-reduce leaves every pair's packed blocks on every rank; --shard keypoints splits every pair's keypoints instead
-- through the PRODUCT's collective (mbavo_allreduce_blocks[_to] on the context's own RCCL communicator);
torch.distributed (backend nccl == RCCL) is the rendezvous, the barrier and the max-over-ranks of the clock.  After the
timed region the reduced normal equations are compared with a single-GPU evaluation of the whole workload (rank 0;
1e-12; pairs: bit-exact).  value = pixel-samples of all ranks / max-over-ranks time.  At N > 1 the line also carries, per
rank, the dominant kernel's duration and the all-reduce's, and under "configs" the 512-pair batch in both shardings.
The step's collective at N > 1 is SELECTED between the product's two: the RCCL run above, then -- after a canary child process per
rank has set the one-shot p2p collectives up and checked them, so that a platform fault costs a child, not the line -- the same
step through mbavo_all*_blocks_p2p by the same timing procedure; verified (reduction check) and faster, it is the line's step
(`comm`, config.parallelism and comm_profile_p2p.selected_as_the_step say so; the other run's figures are in comm_profile_rccl).

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

# the CPU baseline's OpenMP threads must SLEEP at the end of their share, not spin: the sandboxed hosts meter CPU time, and
# spinning waiters eat the quota of the threads still working (read by libgomp when it is first loaded)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == FP64 matrix (v_mfma_f64) peak: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz;
                          # one shared pipe (tools/micro/mfma_valu_overlap.hip; 75.2 TFLOP/s sustained by v_mfma_f64_16x16x4)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s

WORKLOADS = ["c2_dense", "c2_semidense", "c1_dense", "c3_batch64", "c4_batch512", "c5_1080p",
             "c3_batch64_shared", "c4_batch512_shared"]
SIDE_CONFIGS = ["c2_semidense", "c1_dense", "c3_batch64", "c4_batch512", "c5_1080p"]  # + c5 fp16, the named extras below


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2_dense", choices=WORKLOADS)
    ap.add_argument("--shard", default=None, choices=["pairs", "keypoints", "frames", "frame_blocks"],
                    help="N > 1 sharding (default: the workload's own -- frames for a single pair, pairs for a batch of pairs)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "gloo", "p2p", "p2p-shared"],
                    help="N > 1 only.  rccl: one GPU per rank, the product's collectives on the context's RCCL communicator.  gloo: the "
                         "ranks SHARE the visible GPU(s) (rank r on GPU r %% device_count) and a gloo collective on a pinned host copy "
                         "stands in for RCCL (shard.HostStagedCollective) -- executes every line of the N > 1 path on a one-GPU box "
                         "except ncclAllReduce / ncclAllGather themselves; its timings are not scaling figures.  p2p: one GPU per rank, the product's "
                         "ONE-SHOT collectives over peer-mapped receive regions (csrc/p2p_comm.hip, no RCCL) as the collective of the step.  "
                         "p2p-shared: the same collectives between ranks that share the visible GPU(s) (gloo only carries the rendezvous)")
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


def build_workload(name, frames=1, seed=1, ctx=None, dev="cuda:0", grad_fp16=False, pairs=None, k=4):
    """(list of Prob or a device-resident RenderedPairBatch, description, sharding mode at N > 1)"""
    from mba_vo_amd import workloads as wl
    kd = "" if k == 4 else "; LINEAR spline k = 2 on N = 2 control poses (the reference's default degree, blur_aware_direct_tracker.h:50)"
    if name == "c2_dense":
        return wl.pyramid_pair(480, 640, 4, S=8, k=k, N=k, mode="dense", seed=seed, frames=frames), \
            "640x480 pair, 4-level pyramid, S=8 blur samples, N=%d control poses (k=%d), dense P=1 (configs[1]); synthetic " \
            "band-limited noise keyframe, current image = shifted keyframe + noise" % (k, k), "frames"
    if name == "c2_semidense":
        return wl.pyramid_pair(480, 640, 4, S=8, k=k, N=k, mode="semidense", seed=seed, frames=frames), \
            "640x480 pair, 4-level pyramid, S=8, N=%d, semi-dense 30px grid keypoints x 8-pixel pattern (configs[1], " \
            "reference-shaped)" % k + kd, "frames"
    if name == "c1_dense":
        return wl.pyramid_pair(480, 640, 1, S=1, k=k, N=k, mode="dense", seed=seed, frames=frames), \
            "640x480 pair, 1 level, S=1 (sharp degenerate case), dense (configs[0])" + kd, "frames"
    if name in ("c3_batch64", "c4_batch512"):
        B = pairs if pairs else (64 if name == "c3_batch64" else 512)
        return wl.RenderedPairBatch(ctx, B, S=8, k=k, device=dev, seed=seed, grad_fp16=grad_fp16), \
            "batch of %d independent 640x480 pairs = %d consecutive frames of ONE GPU-rendered synthetic blurred sequence " \
            "(textured plane, camera on a ground-truth spline; generate_synthetic_data.cpp:127-214): every pair has its OWN " \
            "keyframe (sharp rendering), gradient image, grid-selected keypoints x 8-pixel pattern with depths from its own " \
            "z-map, motion-blurred current image and control knots; S=8, N=4 (configs[%d])" % (B, B, 2 if B == 64 else 3), "pairs"
    if name in ("c3_batch64_shared", "c4_batch512_shared"):
        B = 64 if name.startswith("c3") else 512
        return wl.pair_batch(B, S=8, k=4, N=4, mode="semidense", seed=seed), \
            "NAMED EXTRA, not configs[%d]: %d pairs that share ONE keyframe / gradient image / keypoint set (L2-resident), " \
            "every pair its own knots and shifted-noise current image" % (2 if B == 64 else 3, B), "pairs"
    if name == "c5_1080p":
        return wl.pyramid_pair(1080, 1920, 1, S=16, k=4, N=6, mode="dense", seed=seed, frames=frames), \
            "1920x1080 pair, 1 level, S=16, N=6 control poses, dense (configs[4])", "frames"
    raise ValueError(name)


def committed_counters(kind, workload):
    """Counter extracts committed under profiles/ (the PMC passes need rocprofv3 and are collected outside this process,
    tools/hbm_traffic.sh / tools/pmc_fp64.sh / tools/pmc_all.sh): newest round first.  Every extract carries the hash of the
    kernel sources it was collected at (`_source_sha`, mba_vo_amd.capi.kernel_source_sha)."""
    suffix = "" if workload == "c2_dense" else "_" + workload
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_%s%s.json" % (kind, suffix))), reverse=True):
        try:
            return json.load(open(path)), os.path.basename(path)
        except Exception:
            continue
    return None, None


def stale_flags(sources):
    """{file: True/False}: was the committed extract collected at another revision of the kernel sources than the one
    this process runs?  (True also for extracts of earlier rounds that carry no hash.)"""
    from mba_vo_amd import capi
    now = capi.kernel_source_sha()
    out = {}
    for kind, workload in sources:
        h, src = committed_counters(kind, workload)
        if h is not None:
            out[src] = bool(h.get("_source_sha") != now)
    return now, out


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
        if isinstance(v, dict) and base + "<" in k and "flops_fp64_per_launch" in v:
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
        if isinstance(v, dict) and base + "<" in k and "SQ_INSTS_VALU" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
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
    # thread counts of the all-threads sample: every logical CPU and, because the sandboxed hosts hand a process a CPU-time
    # quota far below their logical CPU count (256 OpenMP threads ran SLOWER than one there), the cgroup's quota and 16
    cands = {T}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cands.add(max(1, min(T, -(-int(q) // int(per)))))
    except Exception:
        pass
    if T > 16:
        cands.add(16)
    cands = sorted(c for c in cands if c > 1)
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
               "geometry, Huber and block reductions = oracle restatement; chunks of 64 keypoints spread over OpenMP threads INSIDE the "
               "compiled code (oracle/ref_shim.cpp: ref_evaluate_omp), per-thread frame blocks added in thread order")
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

    def sample(threads, budget=None):
        budget = budget_s if budget is None else budget
        t_all, reps, blocks = 0.0, 0, None
        while reps < 1 or (t_all + t_all / reps < budget and reps < 20):
            t0 = time.perf_counter()
            blocks = run(threads)
            t_all += time.perf_counter() - t0
            reps += 1
        return ps * reps / t_all / 1e6, reps, t_all, blocks

    v1, r1, t1, blocks = sample(1)
    out = dict(value=round(v1, 3), unit="Mpixel-samples/s", cores=1, kind=kind,
               sample="%d full H/g evaluation(s) of the same workload (%d pixel-samples each) on 1 thread, %.1f s; %s"
                      % (r1, ps, t1, how), host_logical_cpus=os.cpu_count())
    if cands:
        best = None
        tried = {}
        for c in cands:  # the budget is shared; the best count is the quoted one
            vc, rc_, tc, blk = sample(c, budget_s / len(cands))
            tried[c] = round(vc, 3)
            if best is None or vc > best[0]:
                best = (vc, rc_, tc, blk, c)
        vT, rT, tT, blocks, T = best
        try:
            usable = len(os.sched_getaffinity(0))
        except Exception:
            usable = None
        out["all_threads"] = dict(value=round(vT, 3), unit="Mpixel-samples/s", cores=T,
                                  sample="%d evaluation(s) on %d threads, %.1f s" % (rT, T, tT),
                                  speedup_over_1_thread=round(vT / v1, 2), cpus_in_affinity_mask=usable, thread_counts_tried=tried,
                                  note="a stated baseline, not a tuned one: an OpenMP loop over keypoint chunks inside the compiled "
                                       "reference code (round 4; a Python thread pool around it before).  The sandboxed host gives "
                                       "this process a fraction of its logical CPUs' real time, so the speed-up over 1 thread is "
                                       "bounded by the sandbox's CPU quota, not by the code")
        if vT > v1:  # the better of the two is the quoted baseline, its thread count stated
            out.update(value=round(vT, 3), cores=T)
            out["single_thread"] = dict(value=round(v1, 3), cores=1)
            out["sample"] = "%d full H/g evaluation(s) of the same workload (%d pixel-samples each) on %d threads, %.1f s " \
                            "(1 thread: %.3f Mpixel-samples/s); %s" % (rT, ps, T, tT, v1, how)
    return out, np.concatenate(blocks, 0)


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


class Runner:
    """One workload resident on this rank's GPU: step(), unit counts, roofline figures."""

    def __init__(self, M, ctx, name, dev, rank, world, sharded, grad_fp16=False, shard_mode=None, sequential=False, coll=None,
                 pair_collective="allgather", pairs=None, cost_only=False, k=4):
        from mba_vo_amd import shard, workloads as wl
        self.M, self.ctx, self.name, self.world, self.rank = M, ctx, name, world, rank
        self.cost_only = bool(cost_only)
        built, self.desc, self.mode = build_workload(name, frames=world if sharded else 1, ctx=ctx, dev=dev, grad_fp16=grad_fp16, pairs=pairs, k=k)
        if cost_only:
            self.desc += "; COST-ONLY evaluation (no Jacobians, no H / g: the candidate pass of an LM iteration)"
        elif not sharded and not sequential:
            self.desc += "; the step ENDS IN THE REFERENCE'S UNIT: merged [cost | g | H] per problem on the device (mbavo_eval_batch_merged: " \
                         "merge_hessian_gradient_cost inside the finalize step), packed frame blocks beside it"
        if shard_mode is not None:
            self.mode = shard_mode
        elif self.mode == "frames":
            self.mode = "frame_blocks"  # (the packed blocks are summed, no merge kernel in the step; --shard frames: merged systems)
        if isinstance(built, wl.RenderedPairBatch):
            self.dw, self.probs = built, built.probs
            built.count_distinct_taps(ctx)  # compulsory bytes = the DISTINCT tap locations (host count, actual knots)
            if grad_fp16:
                self.desc += ", packed keyframes (one word per pixel: intensity + both differences)" if int(grad_fp16) == 2 else ", fp16 gradient images"
        else:
            self.probs = built
            if grad_fp16:
                for p in self.probs:
                    p.grad_fp16 = int(grad_fp16)
                self.desc += ", packed keyframe pyramid (one word per pixel: intensity + both differences)" if int(grad_fp16) == 2 else ", fp16 gradient pyramid"
            self.dw = wl.DeviceWorkload(self.probs, device=dev)
        self.se = shard.ShardedEvaluation(ctx, self.dw.array, self.dw.k, rank, world, self.mode, dev, collective=coll,
                                          pair_collective=pair_collective) if sharded else None
        self.wl = wl
        # the four pyramid levels one after the other, as blur_aware_direct_tracker.cpp:571-575 runs them (an LM loop cannot
        # evaluate a finer level before the coarser one has converged): one mbavo_eval_batch call per problem
        self.sequential = sequential
        if sequential:
            import ctypes as C
            self.desc += "; the levels evaluated ONE AFTER THE OTHER, coarse to fine (one launch sequence per level)"
            self._seq = []
            rows = np.cumsum([0] + [p.F for p in self.probs])
            for b in reversed(range(self.dw.B)):
                one = (M.capi.Problem * 1)()
                C.memmove(C.byref(one[0]), C.byref(self.dw.array[b]), C.sizeof(M.capi.Problem))
                self._seq.append((one, int(rows[b])))

    def step(self):
        if self.cost_only:
            self.dw.step(self.ctx, False)
        elif self.se is not None:
            self.se.step(True)
        elif self.sequential:
            lib, dw = self.ctx.lib, self.dw
            for one, row in self._seq:
                rc = lib.mbavo_eval_batch(self.ctx.handle, 1, one, dw.k, 1, dw.frame_blocks.data_ptr() + 8 * row * dw.E, None,
                                          dw.valid.data_ptr() + 8 * row)
                if rc != 0:
                    raise RuntimeError("mbavo_eval_batch failed: %d" % rc)
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
        self.step()
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
            # H/g evaluation PS (363 + 48 k) + PX (2 E + 12 k + 13); cost-only 122 PS + 13 PX
            flops += (px * S * 122 + px * 13) if self.cost_only else (px * S * (363 + 48 * p.k) + px * (2 * E + 12 * p.k + 13))
        sh = None if self.se is None else (self.mode.replace("frame_blocks", "frames"), self.rank, self.world)
        nbytes = self.wl.algorithmic_bytes(self.probs, sh)
        self.nbytes_upper = self.wl.algorithmic_bytes(self.probs, sh, upper=True)  # (== nbytes unless the pairs carry a distinct-tap count)
        ach_tf = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        ach_gbs = nbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        return flops, nbytes, ach_tf, ach_gbs

    def distinct_summary(self):
        """per pixel-sample: distinct keyframe pixels tapped, 128-byte lines touched (pairs with their own images only)"""
        d = [(p.distinct, p.pixel_samples) for p in self.probs if getattr(p, "distinct", None)]
        if not d:
            return None
        ps = float(sum(n for _, n in d))
        return {"pairs_counted": len(d), "distinct_keyframe_pixels_per_pair": round(sum(x[0] for x, _ in d) / len(d), 1),
                "distinct_current_pixels_per_pair": round(sum(x[1] for x, _ in d) / len(d), 1),
                "lines_128B_touched_per_pair": round(sum(x[2] for x, _ in d) / len(d), 1),
                "compulsory_bytes_per_pixel_sample": round(sum(p.image_bytes for p in self.probs if getattr(p, "distinct", None)) / ps, 3),
                "gather_bound_bytes_per_pixel_sample": round(sum(p.image_bytes_upper for p in self.probs if getattr(p, "distinct", None)) / ps, 3),
                "line_granular_bytes_per_pixel_sample": round(128.0 * sum(x[2] for x, _ in d) / ps, 3)}


def launches_per_step(kernel):
    """What one evaluation enqueues, from the dominant kernel's label (Engine::last_kernel)."""
    if kernel.startswith("k_fused_sp<") and kernel.endswith(",true>"):
        return "1: k_fused_sp<.., ONE> (pose entries in its prologue, finalize by the last workgroup of a slot)"
    if kernel.startswith("k_fused<") and kernel.endswith(",true>"):
        return "2: k_fused<.., POSE> (pose entries in its prologue: inside kernel_ms, so frac is lower than the sample loop's) + k_finalize"
    return "3: k_pose_table + fused kernel + k_finalize"


def kernel_name(ctx):
    return ctx.lib.mbavo_last_kernel(ctx.handle).decode()


def bounded_run(M, ctx, r, min_steps=40, seconds=0.25, max_steps=4000, sync=None, every=4):
    """(steps, seconds, kernel_ms, kernel name) of a bounded timing run of Runner r (side configs)."""
    import torch
    sync = sync or torch.cuda.synchronize
    for _ in range(5):
        r.step()
    sync()
    ctx.lib.mbavo_profile(ctx.handle, every)
    n, t0 = 0, time.perf_counter()
    while n < min_steps or (time.perf_counter() - t0 < seconds and n < max_steps):
        for _ in range(20):
            r.step()
        sync()
        n += 20
        if r.se is not None and n >= min_steps:  # every rank must leave the loop after the same number of steps
            break
    dt = time.perf_counter() - t0
    ms, nl = np.zeros(1), np.zeros(1, np.int32)
    M.capi.check(ctx.lib.mbavo_profile_read(ctx.handle, M.capi.dp(ms), M.capi.ip(nl)), "mbavo_profile_read")
    ctx.lib.mbavo_profile(ctx.handle, 0)
    return n, dt, float(ms[0]) / max(int(nl[0]), 1), kernel_name(ctx)


def trackframe_config(M, ctx, dev):
    """BlurAwareDirectTracker::trackFrame on a GPU-rendered blurred sequence, reference-shaped configuration
    (blur_aware_direct_tracker.cpp:88-203,544-637): wall time of the mbavo_vo_track_frame calls, and the absolute trajectory
    error against the ground truth (product code only; the oracle comparison is in the checker leg)."""
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


def trackframe_checker(ctx, seq, got, gt_rel):
    """Checker leg (beside cpu_baseline): the CPU oracle's trackFrame on the SAME rendered sequence -- knot start indices
    and keyframe decisions must be identical, |ATE_gt(gpu) - ATE_gt(oracle)| <= 1e-5 (BASELINE.json north_star)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frontend
    from oracle import binding as B
    from mba_vo_amd import sequence
    t0 = time.perf_counter()
    want = frontend.run_oracle_vo(B, seq, sequence.REFERENCE_CFG)
    dt = time.perf_counter() - t0
    ate_o = float(np.sqrt(np.mean([np.sum((w["T"][:3] - g[:3]) ** 2) for w, g in zip(want, gt_rel)])))
    ate_g = sequence.ate(got, gt_rel)
    return {"ate_gt_oracle": ate_o, "abs_delta_ate_vs_oracle": abs(ate_g - ate_o),
            "start_idx_equal": bool(all(a["start_idx"] == b["start_idx"] for a, b in zip(got, want))),
            "keyframe_decisions_equal": bool(all(a["is_keyframe"] == b["is_keyframe"] for a, b in zip(got, want))),
            "trace_lengths_equal": bool(all(a["num_trace"] == b["num_trace"] for a, b in zip(got, want))),
            "max_abs_pose_diff": float(max(np.abs(a["T"] - b["T"]).max() for a, b in zip(got, want))),
            "oracle_ms_per_frame_1_thread": round(1e3 * dt / len(want), 3), "within_1e-5": bool(abs(ate_g - ate_o) <= 1e-5),
            "long_horizon": trackframe_long_horizon(ctx)}


def trackframe_long_horizon(ctx, frames=120):
    """Checker leg: trackFrame over `frames` rendered 640x480 frames on a bounded trajectory (synth.loop_spline, ~40 % keyframes)
    against the oracle, free-running and TEACHER-FORCED (the HIP tracker put into the oracle's state before every frame); the
    300-frame statistics and what they mean are in profiles/r05_long_horizon.txt and tests/test_gpu_horizon.py."""
    import frontend
    import horizon
    import mba_vo_amd as M
    from oracle import binding as B
    from mba_vo_amd import sequence
    seq = sequence.make_sequence(ctx, H=480, W=640, M=frames, trajectory="loop")
    cfg = dict(sequence.REFERENCE_CFG)
    t0 = time.perf_counter()
    want = frontend.run_oracle_vo(B, seq, cfg)
    dt = time.perf_counter() - t0
    gt = frontend.gt_relative(B, seq)
    free = horizon.compare(frontend.run_gpu_vo(M, ctx, seq, cfg), want, gt, min_step_quality=cfg["min_quality"])
    tf = horizon.compare(frontend.run_gpu_vo(M, ctx, seq, cfg, teacher=want), want, gt, min_step_quality=cfg["min_quality"])
    pick = lambda st: {"first_discrete_divergence_frame": st["first_discrete_divergence"], "first_pose_divergence_frame": st["first_pose_divergence"],
                       "max_abs_pose_diff": st["max_abs_pose_diff"], "abs_delta_ate": st["abs_delta_ate"],
                       "abs_delta_ate_50_frame_windows_max": st["abs_delta_ate_windows_max"], "ate_gt_gpu": st["ate_gt_gpu"], "ate_gt_oracle": st["ate_gt_oracle"]}
    return {"frames": frames + 1, "keyframes_oracle": free["keyframes_oracle"], "lm_records_oracle": free["lm_records_oracle"],
            "oracle_seconds_1_thread": round(dt, 2), "free_running": pick(free), "teacher_forced": pick(tf),
            "teacher_forced_all_discrete_results_identical": tf["first_discrete_divergence"] is None,
            "teacher_forced_within_1e-5": bool(tf["abs_delta_ate"] <= 1e-5 and (tf["abs_delta_ate_windows_max"] or 0) <= 1e-5),
            "note": "free-running, a rounding-level difference grows ~1.4x per frame (feedback through the velocity prediction and an LM "
                    "loop stopped at finite tolerance) until a discrete decision flips and the runs decorrelate to the tracker's own drift; "
                    "the oracle's own FMA-contracted build leaves the pinned oracle at frame 1 (profiles/r05_long_horizon.txt). "
                    "Teacher-forced, every frame is a one-step comparison from identical inputs."}


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
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--p2p-canary"], env=env, stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, timeout=240)
    except subprocess.TimeoutExpired:
        return False, "timeout"
    return p.returncode == 0, "rc %d%s" % (p.returncode, (": " + p.stderr.decode(errors="replace").strip().splitlines()[-1][:200]) if p.returncode and p.stderr.strip() else "")


def main():
    args = parse()
    if args.p2p_canary:
        raise SystemExit(p2p_canary_child())
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
    shared_gpu = args.comm in ("gloo", "p2p-shared")  # the ranks share the visible GPU(s); gloo is the process group (see --comm)
    use_p2p = args.comm in ("p2p", "p2p-shared")
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
        obj = {"frames": "merged [cost | g | H] systems", "frame_blocks": "packed frame blocks (every rank's frames in its slice, rank-major)", "keypoints": "packed frame blocks (partial sums over the ranks' keypoint bands)",
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
        """Per-rank duration of the all-reduce alone (events around the collective, every rank's own evaluation before it:
        the figure includes the wait for the slowest rank's evaluation) and of the local evaluation + merge alone."""
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

    run = Runner(M, ctx, args.workload, dev, rank, world, use_dist, 2 if args.packed_keyframes else int(args.grad_fp16), shard_mode=args.shard,
                 coll=coll, pair_collective=args.collective, pairs=args.batch_pairs if args.batch_pairs != 512 else None,
                 cost_only=args.cost_only, k=args.spline_k)
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
    d2h_ms = None
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

    counts = run.local_counts()
    ps_rank = sum(px * S for px, S, _ in counts)
    ps_launched = sum(p.pixel_samples for p in run.probs) / (world if run.se is not None else 1)
    ps_all = sum_over_ranks(ps_rank)

    out = None
    if rank == 0:
        flops, nbytes, ach_tf, ach_gbs = run.figures(counts, k_ms)
        traffic, traffic_src = measured_hbm_traffic(wkey, kernel)
        exe, exe_src = executed_fp64_flops(wkey, kernel)
        sha_now, stale = stale_flags([("hbm_counters", wkey), ("pmc_fp64", wkey), ("pmc_sq", wkey)])
        per_step = [r / args.steps * 1e3 for r in regions]
        out = {
            "metric": "Mpixel-samples/s per GN iteration (640x480, 4-lvl pyr, 8 blur samples)" if wkey.startswith("c2") and wkey == args.workload
                      else "Mpixel-samples/s per GN iteration (%s)" % wkey,
            "value": round(ps_all * args.steps / elapsed / 1e6, 3), "unit": "Mpixel-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong" if (run.se is not None and run.mode in ("keypoints", "pairs")) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": len(regions), "ms_per_step_min_max": [round(min(per_step), 5), round(max(per_step), 5)],
            "config": {"workload": run.desc, "name": args.workload, "problems_per_rank": len(run.probs) if run.se is None else run.se.n_live,
                       "pixel_samples_per_step_per_rank": ps_rank, "pixel_samples_launched_per_rank": ps_launched,
                       "parallelism": ("workload sharded by %s over %d rank(s): evaluation -> %sONE %s of %d doubles per step"
                                       % (run.mode, world, "device merge -> " if run.mode == "frames" else "", collective_name(run), run.se.count))
                       if run.se is not None else "1 GPU"},
            "roofline": {"bound": "fp64",
                         "frac_executed": round(exe / (k_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5) if exe and k_ms > 0 else None,
                         "frac_executed_label": "UTILISATION of the FP64 pipe: the FP64 flops the kernel actually issues (SQ counters of the committed "
                                                "extract) / kernel duration / peak; bounded by 1 -- the figure to judge the kernel by",
                         "executed_fp64_flops_per_launch": exe, "executed_source": exe_src,
                         "achieved": round(ach_tf, 4), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tf / FP64_PEAK_TFLOPS, 5),
                         "frac_label": "REFERENCE-FLOP EQUIVALENT (the contract's achieved / peak on SURVEY.md 8(d)'s algorithmic count, i.e. the "
                                       "reference's arithmetic as written, which the kernel undercuts by CSE): NOT a utilisation, can exceed 1 "
                                       "(1.37-1.41 on configs[4])",
                         "issue_busy_frac": issue_busy_fraction(wkey, kernel, k_ms),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_source_sha": sha_now, "counter_extracts_stale": stale, "stale": bool(any(stale.values())) if stale else None,
                         "kernel": kernel, "kernel_ms": round(k_ms, 6), "launches_timed": int(nlaunch[0]),
                         "algorithmic_flops_per_launch": flops,
                         "step_frac": round(flops / (elapsed / args.steps) / 1e12 / FP64_PEAK_TFLOPS, 5) if elapsed > 0 else None,
                         "launches_per_step": launches_per_step(kernel),
                         "note": "binding roofline = the FP64 pipe (FP64 VALU and f64 MFMA share it; 78.6 TFLOP/s; this is "
                                 "the contract's 'mfma' bound): intensity ~150 flop/B >> 9.8 flop/B balance.  frac counts flops "
                                 "as the reference source writes them (SURVEY.md 8d) and can exceed 1 because the kernel "
                                 "applies CSE; frac_executed counts the FP64 flops the kernel issues (SQ counters) and cannot; "
                                 "issue_busy_frac = share of SIMD issue cycles taken by ANY vector / matrix instruction; "
                                 "traffic, frac_executed and issue_busy_frac are read from committed counter extracts: "
                                 "`stale` says whether any of them was collected at another revision of the kernel sources"},
            "roofline_hbm": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": nbytes,
                             "traffic": traffic,
                             "note": "compulsory bytes only; compute-bound kernel, low by construction; traffic = "
                                     "(2*FETCH_SIZE + WRITE_SIZE) KiB from the committed TCC counter passes"},
        }
        if args.workload in ("c3_batch64", "c4_batch512"):
            # pairs with their own images: the tap gather binds (no-taps ablation: 47 % of the 512-pair kernel, profiles/
            # r03_kfused_experiments.txt 3.), so the HBM figure leads and the FP64 one rides along
            out["roofline_fp64"] = out["roofline"]
            h = out.pop("roofline_hbm")
            h.update(kernel=kernel, kernel_ms=round(k_ms, 6), launches_timed=int(nlaunch[0]), traffic_source=traffic_src,
                     frac_of_traffic=round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and k_ms > 0 else None,
                     frac_upper=round(run.nbytes_upper / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if k_ms > 0 else None,
                     algorithmic_bytes_upper_per_launch=run.nbytes_upper, distinct_taps=run.distinct_summary(),
                     note="achieved = COMPULSORY bytes as SURVEY.md 8(d) defines them -- the bytes of the DISTINCT tap locations of "
                          "every pair (counted on the host for the actual keypoints and knots, workloads.count_distinct_taps), the "
                          "current pixels, keypoints, pose tables, packed blocks -- over the kernel's duration; frac_upper = the same "
                          "with the no-reuse gather bound (36 B per pixel-sample) in their place; traffic = (2*FETCH_SIZE + "
                          "WRITE_SIZE) KiB from the committed TCC counter pass -- every touched 128-byte line of a sparse gather in "
                          "row-major images (distinct_taps.line_granular_bytes_per_pixel_sample); frac_of_traffic = traffic / "
                          "duration / peak")
            out["roofline"] = h
        if d2h_ms is not None:
            out["ms_per_step_incl_d2h"] = round(d2h_ms, 5)
            out["d2h_note"] = "step + copy of the %d packed doubles to pinned host memory + wait (what a host LM loop pays per " \
                              "evaluation; never `value`)" % int(run.dw.frame_blocks.numel())
        if use_dist:
            out["rccl_ranks"] = rccl_ranks
            out["comm"] = ("p2p one-shot collectives" + (", %d ranks on %d GPU(s): NOT a scaling measurement" % (world, torch.cuda.device_count()) if shared_gpu else "")) if use_p2p \
                else "rccl" if not shared_gpu else "gloo stand-in, %d ranks on %d GPU(s): NOT a scaling measurement" % (world, torch.cuda.device_count())
            out["reduction_check"] = reduction
            out["per_rank"] = {"kernel_ms": [round(v, 6) for v in k_ms_ranks],
                               "local_evaluation_ms": [round(v, 6) for v in comm_ms[0]],
                               "collective_ms": [round(v, 6) for v in comm_ms[1]],
                               "note": "kernel_ms: the dominant kernel's dispatch timestamps inside the timed region; "
                                       "local_evaluation_ms / collective_ms: event pairs around the rank's evaluation (+ merge) "
                                       "and around the collective in a separate pass of 40 steps -- the collective's figure "
                                       "includes waiting for the slowest rank"}
    # Both collectives in the N > 1 line.  With RCCL as the step's collective above, the SAME sharded
    # evaluation once more through the product's one-shot p2p collectives, timed by the same procedure -- behind a CANARY: the
    # set-up and a few collectives first run in a child process per rank, so that a platform that faults on peer-mapped memory
    # costs a child and not this line.  Where the p2p run is verified (reduction check) and faster, IT is the line's step: value,
    # ms_per_step and the collective named in config.parallelism are its own, RCCL's figures move to comm_profile_rccl.
    # (--comm gloo, ranks sharing a GPU, runs the same selection with the gloo stand-in in RCCL's place: the mechanics on one GPU.)
    if use_dist and not use_p2p and world > 1 and os.environ.get("MBAVO_BENCH_P2P", "1") != "0":
        p2p_line = None
        ok_mine, why = run_p2p_canary(shared_gpu)
        canary = per_rank(1.0 if ok_mine else 0.0)
        if min(canary) < 1.0:
            p2p_line = {"skipped": "the canary child failed on rank(s) %s (rank %d: %s)" % ([i for i, v in enumerate(canary) if v < 1.0], rank, why)}
        else:
            try:
                # (P2PCollective raises on EVERY rank if any rank's region cannot be created or mapped: the ranks stay in step)
                c2 = shard.P2PCollective(ctx, rank, world, max_doubles=max(int(run.se.count), 1 << 12))
                r2 = Runner(M, ctx, args.workload, dev, rank, world, True, 2 if args.packed_keyframes else int(args.grad_fp16), shard_mode=args.shard,
                            coll=c2, pair_collective=args.collective, pairs=args.batch_pairs if args.batch_pairs != 512 else None,
                            cost_only=args.cost_only, k=args.spline_k)
                use_p2p = True  # (collective_name / reduction_check label what they describe)
                loc2, red2 = comm_profile(r2)
                chk2 = reduction_check(r2)
                regions2, elapsed2, k_ms2, nl2 = time_regions(r2)
                lr2, rr2, kr2 = per_rank(loc2), per_rank(red2), per_rank(k_ms2)
                par2 = "workload sharded by %s over %d rank(s): evaluation -> %sONE %s of %d doubles per step" \
                       % (r2.mode, world, "device merge -> " if r2.mode == "frames" else "", collective_name(r2), r2.se.count)
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
                    out["config"]["parallelism"] = par2
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

    cfgs = {}
    # N > 1: BASELINE configs[3] (512 pairs over the ranks: STRONG scaling, every GPU holds 512 / N pairs) in both shardings and
    # both pair collectives, then the WEAK-scaling points north_star's "independent keyframe-pair alignments shard naturally
    # across the 8 GPUs" asks for (512 pairs PER rank), bounded -- the driver's scaling run only launches the default workload,
    # so the batch's scaling points ride in its line
    if use_dist and not args.no_configs and args.workload == "c2_dense":
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
        if rank == 0:
            out["configs"] = cfgs

    # the other BASELINE configs, bounded (N = 1 only): value, step time, dominant kernel time, both fractions
    track = None
    if rank == 0 and world == 1 and not use_dist and not args.no_configs:
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
        out["configs"] = cfgs

    if rank == 0 and not args.no_cpu_baseline and world == 1 and fb_gpu is not None:  # rank 0 at N = 1 only
        host_probs = run.probs if not hasattr(run.dw, "host_problem") else [run.dw.host_problem(b) for b in range(run.dw.B)]
        cb, fb_cpu = cpu_baseline(host_probs, args.cpu_seconds)
        scale = np.abs(fb_cpu).max(axis=1, keepdims=True)
        cb["gpu_vs_cpu_max_rel_diff"] = float((np.abs(fb_gpu - fb_cpu) / scale).max())
        if track:  # the caller of the path against the oracle's trackFrame on the same sequence
            try:
                cb["trackframe_vs_oracle"] = trackframe_checker(ctx, *track)
            except Exception as e:
                cb["trackframe_vs_oracle"] = {"error": repr(e)}
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
