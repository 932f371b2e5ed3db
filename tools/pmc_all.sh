#!/bin/bash
# SQ counter passes over the default bench (one rocprofv3 --pmc run per counter set, no trace flags), merged per kernel
# into profiles/<tag>_pmc_sq.json.  Usage (GPU box): bash tools/pmc_all.sh r01
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=${PROF_SCRATCH:-gpurun_out}/pmc_all
rm -rf "$OUT"; mkdir -p "$OUT" profiles
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/set$i" -o pmc -- python bench.py --steps 20 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-configs --details-out /dev/null > /dev/null 2> "$OUT/err$i.txt" || tail -3 "$OUT/err$i.txt"
done
python - "$OUT" "profiles/${TAG}_pmc_sq.json" <<'PY'
import csv, sys, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/set*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "mbavo::" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
sys.path.insert(0, ".")
import mba_vo_amd
sha = mba_vo_amd.capi.kernel_source_sha()
json.dump(dict(out, _source_sha=sha), open(sys.argv[2], "w"), indent=1)
for k, d in out.items():
    if "k_fused" in k and d.get("SQ_WAVE_CYCLES"):
        wc = d["SQ_WAVE_CYCLES"]
        print(k, "VALU busy/SIMD-cycle %.2f" % (d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (d.get("SQ_BUSY_CYCLES", 1) * 4)) if False else "",
              {c: round(v / wc, 3) for c, v in d.items() if c.startswith("SQ_") and c not in ("SQ_WAVES",)})
PY
