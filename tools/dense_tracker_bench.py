"""Host-driven LM loop on a DENSE 640x480 4-level pair (mbavo_optimize_trajectory; every level takes the lane-per-pixel
kernel, three launches per evaluation): wall time per call.  Usage: python tools/dense_tracker_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mba_vo_amd as M
from oracle import binding as orc
import tracking
orc.build()
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
sc = tracking.make_tracking_scene(orc, H=480, W=640, levels=4, S=8, k=4, seed=5, mode="dense")
opts = dict(tracking.OPTS)
r = tracking.run_gpu_tracker(M, ctx, sc, opts)
ts = []
for _ in range(5):
    t = time.perf_counter(); r = tracking.run_gpu_tracker(M, ctx, sc, opts); ts.append(time.perf_counter() - t)
n = len(r["trace"])
print("dense tracker: %d LM records, %.3f ms per call (median of 5), %.1f us per record" % (n, 1e3 * sorted(ts)[2], 1e6 * sorted(ts)[2] / n))
