"""Step time of the cost-only pass and of the k = 2 (linear spline, the reference's default degree) H/g pass on the
dense configs[1] workload.  Usage: python tools/mode_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mba_vo_amd as M
from mba_vo_amd import workloads as wl
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for k, N in ((4, 4), (2, 2)):
    probs = wl.pyramid_pair(480, 640, 4, S=8, k=k, N=N, mode="dense", seed=1)
    dw = wl.DeviceWorkload(probs)
    ps = sum(p.pixel_samples for p in probs)
    a, b = timeit(lambda: dw.step(ctx, True)), timeit(lambda: dw.step(ctx, False))
    print("k=%d dense  H/g %.1f us (%.1f Gpx-s/s)   cost-only %.1f us (%.1f Gpx-s/s)" % (k, a, ps / a / 1e3, b, ps / b / 1e3))
for k, N in ((4, 4), (2, 2)):
    probs = wl.pyramid_pair(480, 640, 4, S=8, k=k, N=N, mode="semidense", seed=1)
    dw = wl.DeviceWorkload(probs)
    a, b = timeit(lambda: dw.step(ctx, True)), timeit(lambda: dw.step(ctx, False))
    print("k=%d semi-dense  H/g %.1f us   cost-only %.1f us" % (k, a, b))
