#!/bin/bash
# A/B of the pose entries as the fused kernel's prologue (MBAVO_FUSED_POSE=1) against the pose kernel (=0), one library
cd "$(dirname "$0")/.."
mkdir -p tools/_ab; cp mba-vo_amd/libmbavo.so tools/_ab/libmbavo_cur.so
for w in "$@"; do
  echo "== $w"
  BENCH_ARGS="--workload $w" bash tools/ab_run_env.sh 3 "kernel cur MBAVO_FUSED_POSE=0" "prologue cur MBAVO_FUSED_POSE=1"
done
