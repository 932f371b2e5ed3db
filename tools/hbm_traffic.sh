#!/bin/bash
# HBM traffic of the fused kernel from the TCC counters, one counter per pass (MI355X_MICROARCH.md, HBM section),
# with a calibration run on a streaming copy of known size so the gfx950 FETCH_SIZE under-count can be corrected.
export TMPDIR=/tmp
OUT=${PROF_SCRATCH:-gpurun_out}/hbm
rm -rf "$OUT"; mkdir -p "$OUT"
cat > /tmp/calib.py <<'PY'
import torch
a = torch.empty(256 * 1024 * 1024 // 8, dtype=torch.float64, device="cuda:0").normal_()
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(5):
    b.copy_(a)
torch.cuda.synchronize()
PY
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/bench_$C" -o p -- python bench.py --steps 20 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-configs --details-out /dev/null $HBM_BENCH_ARGS > /dev/null 2> "$OUT/err_$C.txt"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/calib_$C" -o p -- python /tmp/calib.py > /dev/null 2>> "$OUT/err_$C.txt"
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for kind in ("bench", "calib"):
        f = glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, kind, C), recursive=True)[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C:
                acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if "k_fused" in k or "k_pose" in k or "k_finalize" in k or "copy" in k.lower() or "elementwise" in k:
                res["%s|%s|%s" % (kind, C, k)] = dict(mean=sum(v) / len(v), n=len(v))
sys.path.insert(0, ".")
import mba_vo_amd
res["_source_sha"] = mba_vo_amd.capi.kernel_source_sha()
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/hbm_counters.json", "w"), indent=1)
import os
if os.environ.get("HBM_OUT"):
    json.dump(res, open(os.environ["HBM_OUT"], "w"), indent=1)
PY
