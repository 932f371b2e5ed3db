"""Seeded sweep of the host LM loop's shortened evaluation sequence (mbavo_optimize_trajectory on persistent kernels): for random
tracking scenes (size, levels, k, frames, exposure, perturbation) the loop with every short cut (candidates with H / g, ride-along,
re-summation of accepted steps) against the loop with none of them -- records with their costs, knots and final cost must be equal to
the last bit -- and against the oracle's loop (record sequence: level, iteration, kind, outlier count).
Usage (GPU box): python tools/tracker_fuzz.py [first_seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import mba_vo_amd as M
from oracle import binding as orc
import tracking

orc.build()
M.load()
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bit_bad = orc_bad = resums = accepted = waste_bad = 0
import ctypes as C
ride = (C.c_longlong * 3)()
ctx.lib.mbavo_ride_along_stats(ride)  # (reset)
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    H, W = [(120, 160), (240, 320), (480, 640), (96, 128)][rng.integers(4)]
    kw = dict(H=H, W=W, levels=int(rng.integers(2, 5 if H >= 240 else 4)), S=8, k=int(rng.choice([2, 4])), F=int(rng.choice([1, 1, 2])),
              seed=seed, exp=float(rng.choice([0.05, 0.1, 0.2])), perturb=float(rng.choice([1e-3, 4e-3, 8e-3])))
    sc = tracking.make_tracking_scene(orc, **kw)
    on = tracking.run_gpu_tracker(M, ctx, sc, dict(tracking.OPTS))
    off = tracking.run_gpu_tracker(M, ctx, sc, dict(tracking.OPTS, resum=-1, ride_along=-1, speculate=-1))
    same = on["trace"] == off["trace"] and np.array_equal(on["kt"], off["kt"]) and np.array_equal(on["kR"], off["kR"]) and on["cost"] == off["cost"]
    # every ride-along treated as taken at other knots (ride_along = 2): each level starts behind a wasted one that is waited out
    waste = tracking.run_gpu_tracker(M, ctx, sc, dict(tracking.OPTS, ride_along=2))
    waste_bad += 0 if (waste["trace"] == off["trace"] and np.array_equal(waste["kt"], off["kt"]) and np.array_equal(waste["kR"], off["kR"]) and waste["cost"] == off["cost"]) else 1
    want = tracking.run_oracle_tracker(orc, sc, dict(tracking.OPTS))
    seq = [t[:4] for t in on["trace"]] == [t[:4] for t in want["trace"]]
    grew = sum(1 for a, b in zip(on["trace"], on["trace"][1:]) if b[2] == 1 and a[0] == b[0] and b[3] > a[3])
    resums += grew
    accepted += sum(1 for t in on["trace"] if t[2] == 1)
    bit_bad += 0 if same else 1
    orc_bad += 0 if seq else 1
    if not same or not seq:
        print("seed %d %s: short cuts == plain loop: %s; record sequence == oracle's: %s (%d records)" % (seed, kw, same, seq, len(on["trace"])))
ctx.lib.mbavo_ride_along_stats(ride)
print("seeds %d..%d: %d scene(s) where the shortened loop differs from the plain one in any bit; %d where the record sequence differs from the "
      "oracle's; %d accepted steps, %d of them flagged new outliers (re-summations); %d scene(s) where the loop with every ride-along WASTED "
      "differs in any bit (ride-alongs posted %d, used %d, waited out %d)" % (first, first + count - 1, bit_bad, orc_bad, accepted, resums, waste_bad, ride[0], ride[1], ride[2]))
