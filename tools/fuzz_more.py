"""Extended run of the seeded sweeps in tests/test_gpu_fuzz.py (more seeds than the suite's fixed ones).
Usage (GPU box): python tools/fuzz_more.py [first_seed] [count]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401

import mba_vo_amd as M
from oracle import binding as orc
import test_gpu_fuzz as T

orc.build()
M.load()
ctx = M.capi.Context(0, stream=torch.cuda.current_stream().cuda_stream)
if os.environ.get("MBAVO_FUZZ_FLAT_TOL"):  # experiment: a flat 1e-9 instead of the small-problem allowance of _tol()
    T._tol = lambda sc: 1e-9
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    for fn in (T.test_random_problem_matches_oracle, T.test_random_batch_matches_oracle):
        try:
            fn(orc, M, ctx, seed)
        except AssertionError:
            bad += 1
            print("FAIL", fn.__name__, seed)
            traceback.print_exc(limit=1)
print("seeds %d..%d: %d failure(s)" % (first, first + count - 1, bad))
