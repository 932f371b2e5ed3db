"""trackFrame on the GPU-rendered 640x480 sequence (mba_vo_amd/sequence.py, product code only): median / min ms per frame
over a few passes and a digest of the tracked poses (equal digests = bit-identical runs).  Used by tools/ab_vo.sh."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import mba_vo_amd as M
from mba_vo_amd import sequence

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = sequence.make_sequence(ctx, H=480, W=640, M=8)
sequence.track_sequence(ctx, seq)
if os.environ.get("MBAVO_TIMING"):  # the warm-up pass' phases (allocations, code objects) are reported and reset here
    print("-- warm-up pass", file=sys.stderr)
    ctx.lib.mbavo_timing_report()
    print("-- timed passes", file=sys.stderr)
runs = [sequence.track_sequence(ctx, seq) for _ in range(passes)]
pf = sorted(sum(f["seconds"] for f in r) / len(r) for r in runs)
dig = hashlib.sha256(b"".join(np.ascontiguousarray(f["T"]).tobytes() for f in runs[0])).hexdigest()[:10]
gt = sequence.gt_relative(ctx, seq)
print("%.4f %.4f %s %d %.6e" % (1e3 * pf[len(pf) // 2], 1e3 * pf[0], dig, sum(f["num_trace"] for f in runs[0]), sequence.ate(runs[0], gt)))
