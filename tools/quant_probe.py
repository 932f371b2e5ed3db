import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import mba_vo_amd as M
from mba_vo_amd import workloads as wl
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
def run(H, W):
    probs = wl.pyramid_pair(H, W, 1, S=8, k=4, N=4, mode="dense", seed=1)
    dw = wl.DeviceWorkload(probs)
    for _ in range(20): dw.step(ctx)
    torch.cuda.synchronize()
    ctx.lib.mbavo_profile(ctx.handle, 4)
    t = time.perf_counter()
    for _ in range(200): dw.step(ctx)
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / 200 * 1e6
    ms, n = np.zeros(1), np.zeros(1, np.int32)
    ctx.lib.mbavo_profile_read(ctx.handle, M.capi.dp(ms), M.capi.ip(n)); ctx.lib.mbavo_profile(ctx.handle, 0)
    px = H * W
    print("%4dx%4d px %7d  px/256CU %.1f  chunks/wave %.3f  step %.1f us  fused %.1f us  -> %.0f px/us" % (H, W, px, px / 256, px / 256 / 768, el, ms[0] / n[0] * 1e3, px / (ms[0] / n[0] * 1e3)))
import os
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(480, 816), (480, 850), (480, 800), (480, 410), (480, 1228), (480, 1640), (480, 640)]
for H, W in sizes:
    run(H, W)
