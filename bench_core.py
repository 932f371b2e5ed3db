"""bench_core.py -- what bench.py and bench_side.py share: the BASELINE workloads as device-resident Runner objects, the
algorithmic flop / byte counts of SURVEY.md 8(d), the committed counter extracts (profiles/rNN_*.json) and the bounded timing
loop of the side configs.  Product code only: nothing here imports, loads or executes the CPU checker."""
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == FP64 matrix (v_mfma_f64) peak: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz;
                          # one shared pipe (tools/micro/mfma_valu_overlap.hip; 75.2 TFLOP/s sustained by v_mfma_f64_16x16x4)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s

WORKLOADS = ["c2_dense", "c2_semidense", "c1_dense", "c3_batch64", "c4_batch512", "c5_1080p",
             "c3_batch64_shared", "c4_batch512_shared"]
SIDE_CONFIGS = ["c2_semidense", "c1_dense", "c3_batch64", "c4_batch512", "c5_1080p"]  # + c5 fp16, the named extras below



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


def short_shape(name, k=4, pairs=None):
    """<= 100 characters: the workload's shape for the bench line (the long description goes to the details file)"""
    B = pairs if pairs else (64 if name.startswith("c3") else 512)
    return {"c2_dense": "640x480 pair, 4-level pyramid, S=8, N=%d (k=%d), dense P=1 (configs[1])" % (k, k),
            "c2_semidense": "640x480 pair, 4 levels, S=8, N=%d, 30px-grid keypoints x 8-px pattern (configs[1])" % k,
            "c1_dense": "640x480 pair, 1 level, S=1, dense (configs[0])",
            "c3_batch64": "%d rendered 640x480 pairs, own keyframes, S=8, N=4 (configs[2])" % B,
            "c4_batch512": "%d rendered 640x480 pairs, own keyframes, S=8, N=4 (configs[3])" % B,
            "c3_batch64_shared": "64 pairs sharing one keyframe (named extra)", "c4_batch512_shared": "512 pairs sharing one keyframe (named extra)",
            "c5_1080p": "1920x1080 pair, 1 level, S=16, N=6, dense (configs[4])"}[name]


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


def rocprof_kernel_ms(kernel):
    """(mean duration [ms] of `kernel` in the newest committed `rocprofv3 --kernel-trace --stats` summary of the bench command,
    its file, collected at another revision of the kernel sources?) -- profiles/rNN_kernel_stats.csv next to the line bench.py
    printed under the profiler (rNN_bench_under_rocprof.json carries the sources' hash; tools/profile.sh writes both)."""
    import csv
    from mba_vo_amd import capi
    want = kernel.replace("true", "1").replace("false", "0").replace(" ", "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_kernel_stats.csv")), reverse=True):
        try:
            for row in csv.DictReader(open(path)):
                name = row["Name"].split("(")[0].replace("void ", "").replace("mbavo::", "").replace(" ", "")
                name = name.replace("true", "1").replace("false", "0")
                if name == want:
                    sha = None
                    try:
                        line = json.load(open(path.replace("_kernel_stats.csv", "_bench_under_rocprof.json")))
                        sha = line["roofline"].get("kernel_source_sha")
                    except Exception:
                        pass
                    return float(row["AverageNs"]) * 1e-6, os.path.basename(path), bool(sha != capi.kernel_source_sha())
        except Exception:
            continue
    return None, None, None


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
        return round((2.0 * rd + wr) * 1024.0), src
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


class Runner:
    """One workload resident on this rank's GPU: step(), unit counts, roofline figures."""

    def __init__(self, M, ctx, name, dev, rank, world, sharded, grad_fp16=False, shard_mode=None, sequential=False, coll=None,
                 pair_collective="allgather", pairs=None, cost_only=False, k=4):
        from mba_vo_amd import shard, workloads as wl
        self.M, self.ctx, self.name, self.world, self.rank = M, ctx, name, world, rank
        self.cost_only = bool(cost_only)
        built, self.desc, self.mode = build_workload(name, frames=world if sharded else 1, ctx=ctx, dev=dev, grad_fp16=grad_fp16, pairs=pairs, k=k)
        self.shape = short_shape(name, k, pairs) + ("; cost-only" if cost_only else "") + \
            ("; packed keyframes" if int(grad_fp16) == 2 else "; fp16 gradients" if grad_fp16 else "")
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
        return "1: k_fused_sp<..,ONE> (pose prologue, ticket finalize)"
    if kernel.startswith("k_fused<") and kernel.endswith(",true>"):
        return "2: k_fused<..,POSE> (pose prologue inside) + k_finalize"
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
