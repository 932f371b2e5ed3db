"""LM-iteration throughput on a batch of keyframe pairs (BASELINE configs[2]/[3]): device-side batched LM
(mbavo_lm_batch) against the host-driven loop run pair by pair (mbavo_optimize_trajectory, one level).
Usage: python tools/lm_bench.py [B] [max_iterations] [host_pairs]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import mba_vo_amd as mbavo
from mba_vo_amd import workloads

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
HOSTN = int(sys.argv[3]) if len(sys.argv) > 3 else min(B, 16)
capi = mbavo.capi
ctx = capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
probs = workloads.pair_batch(B, mode="semidense", seed=1)
out = {"B": B, "max_iterations": IT, "K": probs[0].K, "P": probs[0].P, "S": probs[0].S}
for solver, name in ((0, "svd"), (1, "ldlt")):
    dw = workloads.DeviceWorkload(probs)
    o = capi.LmBatchOpts()
    o.spline_deg_k, o.max_num_iterations, o.max_consecutive_nonmonotonic_steps = 4, IT, 5
    o.solver_type, o.sync_every = solver, 4
    o.min_step_quality, o.min_abs_cost_decrease, o.max_chi_square_error = 0.5, 0.0, 3.0  # run all IT iterations
    res = (capi.LmBatchResult * B)()
    times = []
    for rep in range(3):
        dw = workloads.DeviceWorkload(probs)  # fresh knots
        torch.cuda.synchronize()
        t = time.perf_counter()
        rc = ctx.lib.mbavo_lm_batch(ctx.handle, B, dw.array, C.byref(o), res, None, 0)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
        assert rc == 0, rc
    iters = sum(r.iterations for r in res)
    out["device_%s" % name] = {"ms_total": 1e3 * min(times), "lm_iterations": iters, "us_per_pair_iteration": 1e6 * min(times) / max(iters, 1),
                               "accepted": sum(r.accepted for r in res), "rejected": sum(r.rejected for r in res)}
    # host-driven loop, pair by pair
    lv = (capi.Level * 1)()
    t_host, it_host = 0.0, 0
    for b in range(HOSTN):
        p, a = probs[b], dw.array[b]
        q = lv[0]
        q.H, q.W, q.K, q.P, q.S = p.H, p.W, p.K, p.P, p.S
        q.d_ref_img, q.d_ref_dIxy, q.d_cur_imgs = a.d_ref_img, a.d_ref_dIxy, a.d_cur_imgs
        q.d_kp_xy, q.d_kp_z, q.d_pattern = a.d_kp_xy, a.d_kp_z, a.d_pattern
        to = capi.TrackOpts()
        to.num_levels, to.spline_deg_k, to.max_num_iterations, to.max_consecutive_nonmonotonic_steps, to.solver_type = 1, 4, IT, 5, solver
        for i in range(4):
            to.intrinsics[i] = float(p.intr[i])
        to.huber_k, to.min_step_quality, to.min_abs_cost_decrease, to.max_chi_square_error = p.huber, 0.5, 0.0, 3.0
        kt, kR = p.knots_t.copy(), p.knots_R.copy()
        start, cost = np.zeros(1, np.int32), np.zeros(1)
        trace = (capi.TraceRec * 64)()
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = ctx.lib.mbavo_optimize_trajectory(ctx.handle, C.byref(to), lv, 1, capi.dp(p.cap), capi.dp(p.exp), p.t0, p.dt,
                                              capi.dp(kt), capi.dp(kR), p.N, capi.ip(start), capi.dp(cost), trace, 64)
        t_host += time.perf_counter() - t
        it_host += max(r.iter for r in trace[:n])
    out["host_%s" % name] = {"pairs_timed": HOSTN, "ms_total": 1e3 * t_host, "lm_iterations": it_host,
                             "us_per_pair_iteration": 1e6 * t_host / max(it_host, 1)}
print(json.dumps(out))
