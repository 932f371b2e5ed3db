#!/bin/bash
# Batched-LM measurements of one GPU-box session: wall clock per round (tools/lm_bench.py, 64 and 512 pairs), the host
# phases of a call (MBAVO_LM_STAMPS=1) and the kernel timeline of the last call under rocprofv3 --kernel-trace.
# Usage (GPU box, repo root): bash tools/lm_profile.sh <tag>      -> gpurun_out/<tag>_lm_*.{jsonl,txt}
TAG=${1:-r04}
export TMPDIR=/tmp
mkdir -p gpurun_out
for B in 64 512; do
  python tools/lm_bench.py $B 10 0 > gpurun_out/${TAG}_lm_batch$B.jsonl 2> gpurun_out/${TAG}_lm_batch$B.err || tail -5 gpurun_out/${TAG}_lm_batch$B.err
  python - <<P
import json
d=json.loads(open("gpurun_out/${TAG}_lm_batch$B.jsonl").read().strip().splitlines()[-1])
for k in ("device_svd","device_ldlt","device_svd_packed_keyframes"):
    v=d[k]; print("B=$B %-28s %7.2f us/round  total %.4f ms  rounds %d  acc %d rej %d  cost %.12g"%(k,v["us_per_round"],v["ms_total"],v["rounds"],v["accepted"],v["rejected"],v["final_cost_sum"]))
P
done
for B in 64 512; do
  MBAVO_LM_STAMPS=1 python tools/lm_bench.py $B 10 0 2>&1 >/dev/null | grep "lm_batch:" | tail -24 > gpurun_out/${TAG}_lm_stamps$B.txt; cat gpurun_out/${TAG}_lm_stamps$B.txt
done
for B in 64 512; do
  OUT=/tmp/lmprof_$B; rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python tools/lm_bench.py $B 10 0 > /dev/null 2> gpurun_out/${TAG}_lm_prof$B.err || tail -5 gpurun_out/${TAG}_lm_prof$B.err
  python tools/lm_timeline.py $OUT > gpurun_out/${TAG}_lm_batch${B}_timeline.txt 2>&1
  cp $(find $OUT -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_lm_batch${B}_kernel_stats.csv
done
head -60 gpurun_out/${TAG}_lm_batch64_timeline.txt
