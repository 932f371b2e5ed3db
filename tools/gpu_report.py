"""Developer report: GPU (HIP) vs oracle differences on a few scenes.  Run on the GPU box."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import binding as B
import mba_vo_amd as M
import scenes

ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
for name, kw in [("k4 S8 P8 K145 F2", dict(S=8, F=2, k=4, P=8, K=145)),
                 ("k2 S8 P8 K145 F1", dict(S=8, F=1, k=2, P=8, K=145)),
                 ("k4 S1 P1 dense96x128", dict(H=96, W=128, S=1, F=1, k=4, P=1, kp="dense", margin=0)),
                 ("k4 S8 P5 border outliers", dict(S=8, F=2, k=4, P=5, K=300, kp="border", outlier_frac=0.1)),
                 ("k4 S16 P8 K500 F3", dict(S=16, F=3, k=4, P=8, K=500)),
                 ("k2 S4 P8 huber0.1", dict(S=4, F=1, k=2, P=8, K=145, huber=0.1))]:
    sc = scenes.Scene(**kw)
    p, keep = sc.oracle_problem(B)
    t = time.time(); ro = B.evaluate(p); to = time.time() - t
    d = scenes.DeviceScene(sc)
    t = time.time(); rg = scenes.gpu_eval(ctx, d); tg = time.time() - t
    t = time.time(); rg = scenes.gpu_eval(ctx, d); tg2 = time.time() - t
    rc = scenes.gpu_eval(ctx, d, with_hessian=False)
    print("%-28s cost rel %.2e  H rel %.2e  g rel %.2e  costonly-diff %.2e | oracle %.3fs gpu %.4fs/%.4fs" % (
        name, abs(rg['cost'] - ro['cost']) / abs(ro['cost']), rel(rg['H'], ro['H']), rel(rg['g'], ro['g']),
        abs(rc['cost'] - rg['cost']), to, tg, tg2))
    fb, pc, valid = scenes.gpu_eval_batch(ctx, [d], sc.k)
    print("   frame blocks rel %.2e patch cost rel %.2e valid %s" % (rel(fb, ro['frame_blocks']),
          rel(pc, ro['patch_blocks'][:, :, 0].ravel()), valid))
