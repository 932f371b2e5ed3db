#!/bin/bash
# interleaved A/B of an environment switch of the batched LM: bash tools/lm_env_ab.sh VAR [rounds] [batch sizes]
VAR=${1:-MBAVO_LM_RETILE}; R=${2:-3}; BS=${3:-"64 512"}
for r in $(seq $R); do for v in 1 0; do for B in $BS; do
  env $VAR=$v python tools/lm_bench.py $B 10 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v B=$B', ' '.join('%s %.2f' % (k.replace('device_',''), d[k]['us_per_round']) for k in ('device_svd','device_ldlt','device_svd_packed_keyframes')), d['device_svd']['accepted'], d['device_svd']['rejected'], d['device_svd']['final_cost_sum'])"
done; done; done | sort
