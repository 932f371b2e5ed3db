#!/bin/bash
# Interleaved A/B timing of the libraries built by tools/ab_build.sh: R rounds over all variants, mean +- std of the
# fused kernel's duration and of the step.
cd "$(dirname "$0")/.."
R=${1:-5}; shift || true
cp mba-vo_amd/libmbavo.so /tmp/libmbavo_good.so
for r in $(seq $R); do
  for f in tools/_ab/libmbavo_*.so; do
    v=$(basename $f .so); v=${v#libmbavo_}
    cp $f mba-vo_amd/libmbavo.so
    python bench.py --steps 300 --warmup 40 --no-cpu-baseline --details-out /dev/null "$@" 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['roofline']['kernel_ms']*1e3, d['ms_per_step']*1e3)"
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
