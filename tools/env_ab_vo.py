"""Interleaved A/B of an environment switch on trackFrame (GPU-rendered sequence, mba_vo_amd/sequence.py): ms per frame
and bit-identity of poses / costs / LM record counts.  Usage: python tools/env_ab_vo.py MBAVO_OVERLAP_SOLVE [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mba_vo_amd as M
from mba_vo_amd import sequence
var = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
seq = sequence.make_sequence(ctx, H=480, W=640, M=8)
first = {}
for r in range(rounds):
    for val in ("0", "1"):
        os.environ[var] = val
        sequence.track_sequence(ctx, seq)
        runs = [sequence.track_sequence(ctx, seq) for _ in range(5)]
        pf = sorted(sum(f["seconds"] for f in x) / len(x) for x in runs)
        first.setdefault(val, runs[0])
        print("%s=%s  ms/frame median %.4f  min %.4f" % (var, val, 1e3 * pf[2], 1e3 * pf[0]))
a, b = first["0"], first["1"]
print("bit-identical poses:", all(np.array_equal(x["T"], y["T"]) for x, y in zip(a, b)), " costs:", all(x["cost"] == y["cost"] for x, y in zip(a, b)),
      " LM records:", all(x["num_trace"] == y["num_trace"] for x, y in zip(a, b)))
