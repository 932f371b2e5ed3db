"""configs[1] dense, every pyramid level evaluated ALONE (how the coarse-to-fine loop of blur_aware_direct_tracker.cpp:571-575 has
to run them): us per evaluation and the kernel taken, per level.  Usage: python tools/level_bench.py [env switches via the shell]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mba_vo_amd as M
from mba_vo_amd import workloads as wl
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
probs = wl.pyramid_pair(480, 640, 4, S=8, k=4, N=4, mode="dense", seed=1)
tot = 0.0
for l, p in enumerate(probs):
    dw = wl.DeviceWorkload([p])
    for _ in range(30): dw.step(ctx, True)
    torch.cuda.synchronize()
    n = 400
    t = time.perf_counter()
    for _ in range(n): dw.step(ctx, True)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / n * 1e6
    tot += us
    print("level %d  %6d px  %7.2f us  %s" % (l, p.K, us, ctx.lib.mbavo_last_kernel(ctx.handle).decode()))
print("sum %.2f us" % tot)
