"""Where the pose PROLOGUE of k_fused<.., POSE> spends its time (library built with -DMBAVO_FUSED_STAMPS -DMBAVO_POSE_STAMPS:
bash tools/ab_build.sh pstamps "-DMBAVO_FUSED_STAMPS -DMBAVO_POSE_STAMPS"; cp tools/_ab/libmbavo_pstamps.so mba-vo_amd/libmbavo.so).
s_memrealtime stamps of thread 0 of every workgroup, mean over the workgroups, us since the workgroup's entry."""
import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import mba_vo_amd as M
from mba_vo_amd import workloads as wl
import bench_core as bench
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
lib = ctx.lib
lib.mbavo_debug_fused_stamps.argtypes = [C.c_void_p, C.c_int]
for name in sys.argv[1:] or ["c2_dense"]:
    probs = bench.build_workload(name, 1)[0]
    dw = wl.DeviceWorkload(probs)
    for _ in range(30): dw.step(ctx, True)
    torch.cuda.synchronize()
    buf = np.zeros(2048 * 8, np.uint64)
    dw.step(ctx, True); torch.cuda.synchronize()
    assert lib.mbavo_debug_fused_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(2048, 8)
    n = int((st[:, 0] > 0).sum()); st = st[:n].astype(np.int64)
    us = (st[:, :6] - st[:, :1]) / 100.0
    print(name, n, "wgs; mean us since entry: descs %.2f | stage A %.2f | stage B %.2f | visible %.2f | ready %.2f" % tuple(us[:, 1:6].mean(0)), ctx.lib.mbavo_last_kernel(ctx.handle).decode())
