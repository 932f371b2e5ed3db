#!/bin/bash
# Executed FP64 work of the dominant kernel from the SQ instruction counters, one rocprofv3 --pmc pass per counter
# (no trace flags), per workload -> profiles/<tag>_pmc_fp64[_<workload>].json, which bench.py reads for
# roofline.frac_executed.  flops = 64 lanes x (2 FMA + ADD + MUL + TRANS) + 512 x MFMA_MOPS_F64 (MOPS unit = 512 flop).
# Usage (GPU box, repo root): bash tools/pmc_fp64.sh r02 [workload ...]
TAG=${1:-r02}; shift || true
WL=${@:-c2_dense}
export TMPDIR=/tmp
mkdir -p profiles gpurun_out
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_[A-Z_]*F64[A-Z0-9_]*" | sort -u > gpurun_out/sq_valu_counters.txt
# a workload token may carry the variant of the evaluation: c2_dense_cost_only, c2_dense_k2, c3_batch64_cost_only, ...
wl_args() { local w=$1 a=""; case $w in *_cost_only) a="$a --cost-only"; w=${w%_cost_only};; esac; case $w in *_k2) a="$a --spline-k 2"; w=${w%_k2};; esac; echo "--workload $w$a"; }
for W in $WL; do
  OUT=${PROF_SCRATCH:-gpurun_out}/pmc_fp64_$W
  rm -rf "$OUT"; mkdir -p "$OUT"
  for C in SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_INSTS_MFMA; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o p -- python bench.py --steps 10 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-configs --details-out /dev/null $(wl_args $W) > /dev/null 2> "$OUT/err_$C.txt" || echo "pass $C failed: $(tail -1 $OUT/err_$C.txt)"
  done
  SUF=""; [ "$W" != "c2_dense" ] && SUF="_$W"
  python - "$OUT" "profiles/${TAG}_pmc_fp64${SUF}.json" <<'PY'
import csv, sys, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "mbavo::" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "SQ_INSTS_VALU_FMA_F64" in m:
        m["flops_fp64_per_launch"] = 64.0 * (2 * m.get("SQ_INSTS_VALU_FMA_F64", 0) + m.get("SQ_INSTS_VALU_ADD_F64", 0)
                                             + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0)) \
                                     + 512.0 * m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)
    m["launches"] = max(len(v) for v in d.values())
    out[k] = m
sys.path.insert(0, ".")
import mba_vo_amd
out["_source_sha"] = mba_vo_amd.capi.kernel_source_sha()
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, m in out.items():
    if "k_fused" in k:
        print(k, {c: round(v) for c, v in m.items()})
PY
done
