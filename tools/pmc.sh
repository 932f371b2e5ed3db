#!/bin/bash
# rocprofv3 PMC pass(es) over the default bench (counters only: no kernel-trace/stats combos that gpurun refuses)
# usage: bash tools/pmc.sh TAG "CTR1 CTR2 ..." [bench args]
TAG=$1; CTRS=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -o pmc -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || tail -5 "$OUT/bench.err"
F=$(find "$OUT" -name "*counter_collection.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "fused" in k or "pose" in k or "finalize" in k:
        print(k, {c: round(v / n[(k, c)], 1) for c, v in d.items()})
PY
