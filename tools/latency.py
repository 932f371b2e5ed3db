"""Developer tool: fixed latency of the evaluation for tiny problems (fused-kernel floor)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mba_vo_amd as M
import scenes
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for name, kw in [("K1 P1 S1", dict(S=1, F=1, k=4, P=1, K=1)), ("K1 P1 S8", dict(S=8, F=1, k=4, P=1, K=1)),
                 ("K64 P1 S8", dict(S=8, F=1, k=4, P=1, K=64)), ("K768 P1 S8", dict(S=8, F=1, k=4, P=1, K=768)),
                 ("K374 P8 S8", dict(S=8, F=1, k=4, P=8, K=374)), ("K374 P8 S8 costonly", dict(S=8, F=1, k=4, P=8, K=374)),
                 ("K374 P8 S8 k2", dict(S=8, F=1, k=2, P=8, K=374))]:
    sc = scenes.Scene(**kw)
    d = scenes.DeviceScene(sc)
    arr = (M.capi.Problem * 1)(d.problem())
    fb = torch.zeros(sc.F * sc.E, dtype=torch.float64, device="cuda:0")
    wh = 0 if "costonly" in name else 1
    for _ in range(20):
        ctx.lib.mbavo_eval_batch(ctx.handle, 1, arr, sc.k, wh, fb.data_ptr(), None, None)
    torch.cuda.synchronize()
    ctx.lib.mbavo_profile(ctx.handle, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 200
    for _ in range(n):
        ctx.lib.mbavo_eval_batch(ctx.handle, 1, arr, sc.k, wh, fb.data_ptr(), None, None)
    e1.record(); torch.cuda.synchronize()
    ms, cnt = np.zeros(1), np.zeros(1, np.int32)
    ctx.lib.mbavo_profile_read(ctx.handle, M.capi.dp(ms), M.capi.ip(cnt)); ctx.lib.mbavo_profile(ctx.handle, 0)
    print("%-22s step %.1f us   fused %.1f us" % (name, e0.elapsed_time(e1) / n * 1e3, ms[0] / max(cnt[0], 1) * 1e3))
