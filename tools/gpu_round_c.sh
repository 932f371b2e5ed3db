#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/rc_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/rc_pytest.log | head -2
for w in c2_dense c2_semidense c1_dense c3_batch64 c4_batch512 c5_1080p; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs --workload $w 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-13s value %.0f Mpx-s/s  step %.4f ms  fused %.4f ms  fp64frac %.3f  %s' % (d['config']['name'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel']))"
done
