#!/bin/bash
# Interleaved A/B of (library variant, environment) pairs: tools/ab_run_env.sh rounds "name lib ENV=.." ... ; bench args in $BENCH_ARGS
cd "$(dirname "$0")/.."
R=$1; shift
specs=("$@")
cp mba-vo_amd/libmbavo.so /tmp/libmbavo_good.so
for r in $(seq $R); do
  for sp in "${specs[@]}"; do
    set -- $sp; name=$1; lib=$2; shift 2
    cp tools/_ab/libmbavo_$lib.so mba-vo_amd/libmbavo.so
    env "$@" python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-configs $BENCH_ARGS 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3)"
  done
done | python -c "
import sys, collections, statistics as st
k=collections.OrderedDict()
for l in sys.stdin:
    n,a,b=l.split(); k.setdefault(n,[]).append((float(a),float(b)))
for n,v in k.items():
    a=[x[0] for x in v]; b=[x[1] for x in v]
    print('%-24s fused %.2f +- %.2f us   step %.2f +- %.2f us   (n=%d)' % (n, st.mean(a), st.pstdev(a), st.mean(b), st.pstdev(b), len(v)))"
cp /tmp/libmbavo_good.so mba-vo_amd/libmbavo.so
