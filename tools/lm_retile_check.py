"""Re-tiled late slots of the batched LM (lm_batch.hip "RE-TILING") against the single layout (MBAVO_LM_RETILE=0) over seeds and batch
sizes of rendered 640x480 pairs: per-pair (iterations, accepted, rejected, invalid, outliers) must be identical, final costs and knots
agree to rounding.  Usage (GPU box): python tools/lm_retile_check.py [seeds] [batch sizes ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import mba_vo_amd as M
from mba_vo_amd import workloads

M.load()
capi = M.capi
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sizes = [int(a) for a in sys.argv[2:]] or [136, 200, 300, 512]
worst_cost, worst_knot, pairs, bad = 0.0, 0.0, 0, 0
for B in sizes:
    for seed in range(1, seeds + 1):
        for fmt in (0, 2):
            batch = workloads.RenderedPairBatch(ctx, B, H=480, W=640, S=8, k=4, seed=seed, grad_fp16=fmt)
            out = {}
            for mode in ("1", "0"):
                os.environ["MBAVO_LM_RETILE"] = mode
                ctx.lib.mbavo_reload_env()  # (the library scans the environment once per process)
                batch.reset_knots()
                o = capi.LmBatchOpts()
                o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = 4, 10, 5
                o.solver_type, o.sync_every = 0, 0
                o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = 0.5, 1e-6 if seed % 2 else 0.0, 3.0
                res = (capi.LmBatchResult * B)()
                rc = ctx.lib.mbavo_lm_batch(ctx.handle, B, batch.array, C.byref(o), res, None, 0)
                assert rc == 0, rc
                torch.cuda.synchronize()
                knots = np.stack([np.concatenate([h["dkt"].cpu().numpy().ravel(), h["dkR"].cpu().numpy().ravel()]) for h in batch._host])
                out[mode] = ([(r.iterations, r.accepted, r.rejected, r.invalid, r.num_outliers) for r in res], np.array([r.final_cost for r in res]), knots)
            same = sum(a == b for a, b in zip(out["1"][0], out["0"][0]))
            dc = float(np.max(np.abs(out["1"][1] - out["0"][1]) / np.maximum(np.abs(out["0"][1]), 1e-300)))
            dk = float(np.max(np.abs(out["1"][2] - out["0"][2])))
            pairs += B; bad += B - same
            worst_cost, worst_knot = max(worst_cost, dc), max(worst_knot, dk)
            print("B=%d seed=%d %s: %d / %d pairs with identical counts, final cost rel. diff %.2e, knots %.2e, iterations %d..%d" % (
                B, seed, "packed" if fmt == 2 else "float", same, B, dc, dk, min(c[0] for c in out["1"][0]), max(c[0] for c in out["1"][0])), flush=True)
print("TOTAL %d pairs, %d with differing counts, worst final-cost difference %.2e, worst knot difference %.2e" % (pairs, bad, worst_cost, worst_knot))
