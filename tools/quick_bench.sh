#!/bin/bash
# quick A/B on the GPU box: parity report + bench numbers (no CPU baseline)
python tools/gpu_report.py 2>&1 | grep -E "rel" | head -4
for w in c2_dense c1_dense c2_semidense c3_batch64; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-13s value %.0f Mpx-s/s  step %.4f ms  fused %.4f ms  fp64frac %.3f' % (d['config']['name'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))"
done
