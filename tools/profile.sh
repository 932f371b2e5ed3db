#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summaries are copied to profiles/<tag>_*.
# Usage (on the GPU box, from the repo root): bash tools/profile.sh r01 [extra bench args]
set -e
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=${PROF_SCRATCH:-gpurun_out}/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" profiles
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python bench.py --no-cpu-baseline --no-configs --details-out /dev/null "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
STATS=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
cp "$STATS" profiles/${TAG}_kernel_stats.csv
tail -1 "$OUT/bench.json" > profiles/${TAG}_bench_under_rocprof.json
cat profiles/${TAG}_kernel_stats.csv | cut -c1-200
