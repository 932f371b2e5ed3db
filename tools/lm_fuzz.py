"""Seeded sweep of the batched LM (mbavo_lm_batch) in three solver configurations of type 0 -- the default (LDL^T in
registers, refined in double-double above a pivot ratio of 1e8), the plain stand-in only (MBAVO_LM_REFINE=0) and the Jacobi
solvers only (MBAVO_FAST_SOLVE=0) -- against the host-driven loop (mbavo_optimize_trajectory) pair by pair: how many pairs
keep the identical (iteration, kind, outlier count) record sequence, and the largest relative difference of the final costs.
Usage (GPU box): python tools/lm_fuzz.py [first_seed] [count] [pairs_per_seed]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import mba_vo_amd as M
from mba_vo_amd import workloads
import test_gpu_lm_batch as T

M.load()
capi = M.capi
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
CONFIGS = (("default", {}), ("plain stand-in", {"MBAVO_LM_REFINE": "0"}), ("Jacobi only", {"MBAVO_FAST_SOLVE": "0"}))
# (k, N, F): n = 24, 18, 12 (LDL^T in registers), 36 with a knot without data (singular: Jacobi in every configuration), 36 and 30 with
# data on every knot (the workgroup LDL^T)
SHAPES = ((4, 4, 1), (2, 3, 2), (2, 2, 1), (4, 6, 2), (4, 6, 3), (2, 5, 4))


def device(dw, k, B):
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = k, T.OPTS["max_it"], T.OPTS["max_nonmono"]
    o.solver_type, o.sync_every = 0, 0
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = T.OPTS["min_q"], T.OPTS["min_dec"], T.OPTS["chi"]
    cap = 64
    res = (capi.LmBatchResult * B)()
    trace = (capi.TraceRec * (B * cap))()
    rc = ctx.lib.mbavo_lm_batch(ctx.handle, B, dw.array, C.byref(o), res, trace, cap)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return [([(t.iter, t.kind, t.num_outliers) for t in trace[b * cap:b * cap + res[b].num_trace]], res[b].final_cost) for b in range(B)]


tot = {name: [0, 0, 0.0] for name, _ in CONFIGS}  # pairs, pairs with identical records, worst final-cost difference among those
for seed in range(first, first + count):
    k, N, F = SHAPES[seed % len(SHAPES)]
    probs = T._scene(B, k, N, F, seed=seed)
    host = None
    for name, env in CONFIGS:
        for key in ("MBAVO_LM_REFINE", "MBAVO_FAST_SOLVE"):
            os.environ.pop(key, None)
        os.environ.update(env)
        ctx.lib.mbavo_reload_env()  # (the library scans the environment once per process)
        dw = workloads.DeviceWorkload(probs)
        if host is None:
            host = []
            for b, p in enumerate(probs):
                kt, kR, cost, tr = T._host_lm(M, ctx, dw, b, p, 0)
                host.append(([(t[0], t[1], t[2]) for t in tr], cost))
        got = device(dw, k, B)
        for b in range(B):
            tot[name][0] += 1
            if got[b][0] == host[b][0]:
                tot[name][1] += 1
                tot[name][2] = max(tot[name][2], abs(got[b][1] - host[b][1]) / max(1.0, abs(host[b][1])))
            else:
                print("DIFF", name, "seed", seed, "pair", b, "(k, N, F) =", (k, N, F), "records", len(got[b][0]), "vs host", len(host[b][0]),
                      "final cost %.9g vs %.9g" % (got[b][1], host[b][1]))
for name, _ in CONFIGS:
    n, same, worst = tot[name]
    print("%-15s %4d pairs, %4d with the host loop's record sequence, largest final-cost difference among those %.2e" % (name, n, same, worst))
