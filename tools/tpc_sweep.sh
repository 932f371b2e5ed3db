for t in 1 2 4 8; do echo "== TILES_PER_CU=$t"; MBAVO_TILES_PER_CU=$t python tools/lm_bench.py 512 10 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('device_svd','device_svd_packed_keyframes'): print(k, d[k]['us_per_round'], d[k]['accepted'], d[k]['final_cost_sum'])"
MBAVO_TILES_PER_CU=$t python bench.py --workload c4_batch512 --steps 100 --warmup 10 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c4_batch512 step %.2f kernel %.2f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
MBAVO_TILES_PER_CU=$t python bench.py --workload c4_batch512 --packed-keyframes --steps 100 --warmup 10 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c4_batch512 packed step %.2f kernel %.2f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done
