#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/rb_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/rb_pytest.log
for one in 0 1; do
  echo "== MBAVO_ONE=$one"
  for w in c2_semidense c3_batch64; do
    MBAVO_ONE=$one python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --workload $w 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-13s value %.0f Mpx-s/s  step %.4f ms  fused %.4f ms  fp64frac %.3f  %s' % (d['config']['name'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel']))"
  done
  MBAVO_ONE=$one python tools/latency.py 2>&1 | tail -8
  MBAVO_ONE=$one python tools/vo_bench.py 8 2>&1 | tail -1
done
