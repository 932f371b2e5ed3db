#!/bin/bash
# All committed profiler artefacts of a round, on the GPU box: bash tools/profile_round.sh r02
TAG=${1:-r02}
export TMPDIR=/tmp
# raw rocprofv3 output stays on the box (gpurun merges at most 64 MiB back): only the extracts travel, through gpurun_out/profiles_<tag>
export PROF_SCRATCH=${PROF_SCRATCH:-/tmp/mbavo_prof}; mkdir -p $PROF_SCRATCH
mkdir -p profiles gpurun_out
bash tools/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; tail -12 gpurun_out/${TAG}_profile.log | cut -c1-160
for W in c2_dense c3_batch64 c4_batch512; do
  SUF=""; [ "$W" != c2_dense ] && SUF="_$W"
  HBM_OUT=profiles/${TAG}_hbm_counters${SUF}.json HBM_BENCH_ARGS="--workload $W" bash tools/hbm_traffic.sh > gpurun_out/${TAG}_hbm_$W.log 2>&1
  python - profiles/${TAG}_hbm_counters${SUF}.json <<'PY'
import json, sys
h = json.load(open(sys.argv[1]))
for k, v in h.items():
    if "k_fused" in k or "calib" in k: print(k[:90], round(v["mean"], 1), v["n"])
PY
done
bash tools/pmc_fp64.sh $TAG c2_dense c3_batch64 c4_batch512 c5_1080p c2_semidense c1_dense c2_dense_cost_only c3_batch64_cost_only c2_semidense_cost_only c2_dense_k2 2>&1 | tail -16
bash tools/pmc_all.sh $TAG > gpurun_out/${TAG}_pmc_all.log 2>&1; tail -3 gpurun_out/${TAG}_pmc_all.log | cut -c1-300
# gpurun only merges gpurun_out/ back: hand the artefacts over through it (copy them into profiles/ at home)
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
