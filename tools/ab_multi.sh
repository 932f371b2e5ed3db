#!/bin/bash
# tools/ab_multi.sh [rounds] workload...: tools/ab_run.sh over several workloads (libraries from tools/ab_build.sh)
cd "$(dirname "$0")/.."
R=${1:-3}; shift
for w in "$@"; do echo "== $w"; bash tools/ab_run.sh $R --workload $w --no-configs; done
