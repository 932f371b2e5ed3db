"""Where the workgroups of the dense kernel k_fused spend their time (library built with -DMBAVO_FUSED_STAMPS:
bash tools/ab_build.sh stamps -DMBAVO_FUSED_STAMPS; cp tools/_ab/libmbavo_stamps.so mba-vo_amd/libmbavo.so).
s_memrealtime stamps (100 MHz) of thread 0 of every workgroup, last launch of a run:
  0 kernel entry | 1 descriptors / table warm / accumulators ready | 2 sample-parallel remainder round done |
  3 wave 0's rounds done | 4 after the end-of-tile barrier (all waves done, accumulators parked) | 5 partial written
Usage: python tools/fused_stamps.py [workload ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mba_vo_amd as M
from mba_vo_amd import workloads as wl
import bench_core as bench
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
lib = ctx.lib
lib.mbavo_debug_fused_stamps.argtypes = [C.c_void_p, C.c_int]
names = sys.argv[1:] or ["c2_dense", "c3_batch64"]
for name in names:
    probs = bench.build_workload(name, 1, ctx=ctx)[0]
    dw = probs if hasattr(probs, "step") else wl.DeviceWorkload(probs)  # rendered pair batches are device workloads already
    for _ in range(30): dw.step(ctx, True)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        buf = np.zeros(2048 * 8, np.uint64)
        dw.step(ctx, True); torch.cuda.synchronize()
        assert lib.mbavo_debug_fused_stamps(buf.ctypes.data, buf.size) == 0
        res.append(buf.reshape(2048, 8).copy())
    st = res[-1]
    n = int((st[:, 0] > 0).sum())
    st = st[:n].astype(np.int64)
    t0 = st[:, 0].min()
    us = (st[:, :6] - t0) / 100.0
    xcc = (st[:, 7] >> 32) & 0xf
    print("== %s: %d workgroups, last entry %.2f us after the first, last exit %.2f us" % (name, n, us[:, 0].max(), us[:, 5].max()))
    lab = ["entry->ready", "remainder round", "wave-0 rounds", "wait + park", "gather + store"]
    for i in range(5):
        dseg = us[:, i + 1] - us[:, i]
        print("   %-16s mean %6.2f  min %6.2f  max %6.2f us" % (lab[i], dseg.mean(), dseg.min(), dseg.max()))
    print("   entry by XCC (mean us):", " ".join("%d:%.2f" % (x, us[xcc == x, 0].mean()) for x in sorted(set(xcc.tolist()))))
    print("   exit  percentiles 10/50/90/100: %.2f %.2f %.2f %.2f us" % tuple(np.percentile(us[:, 5], [10, 50, 90, 100])))
    wb = np.zeros(1024 * 16 * 4, np.uint64)
    lib.mbavo_debug_wave_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.mbavo_debug_wave_stamps(wb.ctypes.data, wb.size) == 0
    wb = wb.reshape(1024, 16, 4).astype(np.int64)
    for b in (0, 1, int(np.argmax(us[:, 5]))):
        ws = wb[b]
        nw = int((ws[:, 0] > 0).sum())
        print("   block %d waves (simd: loop start -> end us): " % b + "  ".join(
            "w%d/s%d:%.1f-%.1f" % (w, (ws[w, 2] >> 4) & 3, (ws[w, 0] - t0) / 100., (ws[w, 1] - t0) / 100.) for w in range(nw)))
    slow = np.argsort(-us[:, 5])[:4]
    for b in slow: print("   slowest block %4d (xcc %d): " % (b, xcc[b]) + " ".join("%.2f" % v for v in us[b]))
