#!/bin/bash
# Timing experiments: rebuild engine.hip with -DMBAVO_EXP_* switches (results are WRONG by design) and time c2_dense.
# usage (GPU box): bash tools/exp_variants.sh "NO_PHASE2" "NO_TAPS" "NO_CHAIN" ...
cd "$(dirname "$0")/.."
cp mba-vo_amd/libmbavo.so /tmp/libmbavo_good.so
for v in "$@"; do
  defs=""; for d in $v; do case $d in FLAG:*) defs="$defs ${d#FLAG:}";; *=*) defs="$defs -DMBAVO_$d";; BASE) ;; *) defs="$defs -DMBAVO_EXP_$d";; esac; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $defs $EXP_FLAGS -c mba-vo_amd/csrc/engine.hip -o /tmp/engine_exp.o 2>/dev/null || { echo "compile failed: $v"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mba-vo_amd/libmbavo.so /tmp/engine_exp.o $(ls mba-vo_amd/build/*.o | grep -v engine) -ldl
  if [ -n "$EXP_CMD" ]; then echo "== $v"; eval "$EXP_CMD"; continue; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs --details-out /dev/null ${EXP_BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s fused %.4f ms  step %.4f ms' % ('$v', r['kernel_ms'], d['ms_per_step']))"
done
cp /tmp/libmbavo_good.so mba-vo_amd/libmbavo.so
