"""Per-step time of the headline workload against the time since the process' first GPU work, on a fresh box: how long the clocks take
to come up (30 ms on the box measured: 194 -> 41.7 -> 36.5 us per step), i.e. whether a short timed region right after W warm-up steps can
sit on the ramp.  Usage (GPU box): python tools/clock_ramp.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
import mba_vo_amd as M
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
r = bench.Runner(M, ctx, "c2_dense", "cuda:0", 0, 1, False)
torch.cuda.synchronize()
t_start = time.perf_counter()
out = []
while time.perf_counter() - t_start < 3.0:
    t0 = time.perf_counter()
    for _ in range(20):
        r.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out.append((t1 - t_start, (t1 - t0) / 20 * 1e6))
for i in list(range(0, 40, 2)) + list(range(40, len(out), max(1, len(out) // 40))):
    print("%.3f s  %.2f us/step" % out[i])
