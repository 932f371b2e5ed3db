"""trackFrame throughput on a synthetic blurred sequence: GPU front end (mbavo_vo_track_frame) against the CPU oracle's
restatement (orc_vo_track_frame, one host core).  Reference-shaped configuration: 640x480, 4 levels, 30-px grid
keypoints x 8-pixel pattern, k = 2 (two control knots), S = 8 blur samples.
Usage: python tools/vo_bench.py [frames]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import frontend
import mba_vo_amd as mbavo
from oracle import binding as orc

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
orc.build()
seq = frontend.make_sequence(orc, H=480, W=640, M=M, trans_scale=0.15, rot_scale=0.02, blur_samples=8)
cfg = dict(frontend.DEFAULTS, levels=4, S=(8, 8, 8, 8), thr=3.0, cell=30, flow0=10.0, flow1=24.0)
ctx = mbavo.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
frontend.run_gpu_vo(mbavo, ctx, seq, cfg)  # warm-up (allocations, code objects)
if os.environ.get('MBAVO_TIMING'):
    print('-- warm-up run', file=sys.stderr); ctx.lib.mbavo_timing_report()
# per frame: the mbavo_vo_track_frame calls only (creating the tracker allocates ~40 device buffers: not a per-frame cost)
calls = []
t = time.perf_counter(); got = frontend.run_gpu_vo(mbavo, ctx, seq, cfg, calls); t_all = time.perf_counter() - t
t_gpu = sum(calls)
if os.environ.get('MBAVO_TIMING'):
    print('-- timed run', file=sys.stderr); ctx.lib.mbavo_timing_report()
t = time.perf_counter(); want = frontend.run_oracle_vo(orc, seq, cfg); t_cpu = time.perf_counter() - t
gt = frontend.gt_relative(orc, seq)
err = [frontend.reprojection_error(seq, o["T"], g)[0] for o, g in zip(got[1:], gt[1:])]
print(json.dumps({"frames": M + 1, "keypoints_level0": got[0]["K"][0], "keyframes": sum(o["is_keyframe"] for o in got),
                  "lm_trace_records": sum(o["num_trace"] for o in got),
                  "gpu_ms_per_frame": 1e3 * t_gpu / (M + 1), "gpu_ms_per_frame_with_create_destroy": 1e3 * t_all / (M + 1),
                  "oracle_ms_per_frame_1core": 1e3 * t_cpu / (M + 1),
                  "same_decisions": all(a["is_keyframe"] == b["is_keyframe"] and a["K"] == b["K"] for a, b in zip(got, want)),
                  "max_pose_diff": max(float(np.abs(a["T"] - b["T"]).max()) for a, b in zip(got, want)),
                  "mean_reprojection_error_px": float(np.mean(err))}))
