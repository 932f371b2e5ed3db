"""Sample-parallel kernel vs lane-per-pixel kernel on batches of B semi-dense pairs (where should the switch be?).
Usage (GPU box): python tools/sp_threshold.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
import mba_vo_amd as M
from mba_vo_amd import workloads as wl
B = int(sys.argv[1])
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
probs = wl.pair_batch(B, mode="semidense", seed=1)
dw = wl.DeviceWorkload(probs)
for _ in range(30): dw.step(ctx, True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(300): dw.step(ctx, True)
torch.cuda.synchronize()
print("%%.2f %%d" %% ((time.perf_counter() - t) / 300 * 1e6, sum(p.pixel_samples // p.S for p in probs)))
''' % ROOT
for B in (1, 2, 4, 8, 12, 16, 24, 32, 48):
    out = []
    for sp in ("0", "1"):
        env = dict(os.environ, MBAVO_SP=sp)
        r = subprocess.run([sys.executable, "-c", CHILD, str(B)], env=env, capture_output=True, text=True)
        out.append(r.stdout.strip().split("\n")[-1] if r.returncode == 0 else "ERR " + r.stderr[-200:])
    print("B=%-3d lane-per-pixel %s   sample-parallel %s" % (B, out[0], out[1]))
