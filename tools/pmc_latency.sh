#!/bin/bash
# Latency / occupancy-of-units counters of the dominant kernel (one rocprofv3 --pmc pass per set, no trace flags):
# average scalar-load, vector-memory and LDS latency = SQ_INST_LEVEL_x / SQ_INSTS_x, instruction-fetch level, unit busy
# cycles.  -> profiles/<tag>_pmc_latency[_<workload>].json.  Usage (GPU box): bash tools/pmc_latency.sh r02 [workload]
TAG=${1:-r02}; W=${2:-c2_dense}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_lat_$W
rm -rf "$OUT"; mkdir -p "$OUT" profiles
i=0
for SET in "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/set$i" -o pmc -- python bench.py --steps 20 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-configs --details-out /dev/null --workload $W > /dev/null 2> "$OUT/err$i.txt" || tail -3 "$OUT/err$i.txt"
done
SUF=""; [ "$W" != "c2_dense" ] && SUF="_$W"
python - "$OUT" "profiles/${TAG}_pmc_latency${SUF}.json" <<'PY'
import csv, sys, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/set*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "mbavo::k_fused" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    for unit, n in (("SMEM", "SQ_INSTS_SMEM"), ("LDS", "SQ_INSTS_LDS")):
        if d.get(n) and d.get("SQ_INST_LEVEL_" + unit): d["avg_latency_cycles_" + unit] = d["SQ_INST_LEVEL_" + unit] / d[n]
    if d.get("SQ_INST_LEVEL_VMEM") and d.get("SQ_INSTS_VMEM_RD"):
        d["avg_latency_cycles_VMEM"] = d["SQ_INST_LEVEL_VMEM"] / (d["SQ_INSTS_VMEM_RD"] + d.get("SQ_INSTS_VMEM_WR", 0))
    if d.get("SQ_IFETCH") and d.get("SQ_IFETCH_LEVEL"): d["avg_latency_cycles_IFETCH"] = d["SQ_IFETCH_LEVEL"] / d["SQ_IFETCH"]
    print(k, json.dumps({c: round(v, 1) for c, v in sorted(d.items())}))
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_pmc_latency${SUF}.json gpurun_out/profiles_$TAG/
