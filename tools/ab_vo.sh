#!/bin/bash
# Interleaved A/B of trackFrame over (library variant, environment) pairs:
#   tools/ab_vo.sh rounds "name lib ENV=.." "name2 lib2" ...     (lib = a tools/_ab/libmbavo_<lib>.so built by ab_build.sh, or "cur")
cd "$(dirname "$0")/.."
R=$1; shift
specs=("$@")
cp mba-vo_amd/libmbavo.so /tmp/libmbavo_good.so
for r in $(seq $R); do
  for sp in "${specs[@]}"; do
    set -- $sp; name=$1; lib=$2; shift 2
    if [ "$lib" = cur ]; then cp /tmp/libmbavo_good.so mba-vo_amd/libmbavo.so; else cp tools/_ab/libmbavo_$lib.so mba-vo_amd/libmbavo.so; fi
    echo "$name $(env "$@" python tools/vo_quick.py 2>/dev/null | tail -1)"
  done
done | python -c "
import sys, collections, statistics as st
k=collections.OrderedDict()
for l in sys.stdin:
    f=l.split()
    if len(f) < 6: print('bad line:', l.strip()); continue
    k.setdefault(f[0],[]).append((float(f[1]),float(f[2]),f[3],f[4],f[5]))
for n,v in k.items():
    print('%-22s ms/frame median %.4f +- %.4f  min %.4f   digest %s  records %s  ATE %s  (n=%d)' % (n, st.mean(x[0] for x in v), st.pstdev([x[0] for x in v]), min(x[1] for x in v), ','.join(sorted(set(x[2] for x in v))), v[0][3], v[0][4], len(v)))"
cp /tmp/libmbavo_good.so mba-vo_amd/libmbavo.so
