#!/bin/bash
# ablations of round 2 (profiles/r02_ablations.txt): no-taps and contraction-free warp, interleaved timing on configs[1]
# and configs[4]; fuzz sweep with a flat 1e-9 tolerance on both numeric variants
export TMPDIR=/tmp
for w in c2_dense c5_1080p; do echo "== $w"; bash tools/ab_run.sh 3 --workload $w --no-configs 2>&1 | tail -3; done
cp mba-vo_amd/libmbavo.so /tmp/libmbavo_good.so
for v in base uvnc; do
  cp tools/_ab/libmbavo_$v.so mba-vo_amd/libmbavo.so
  echo "== fuzz, flat 1e-9 tolerance, variant $v"; MBAVO_FUZZ_FLAT_TOL=1 python tools/fuzz_more.py 2000 400 2>&1 | grep -c "^FAIL"; 
  echo "== fuzz, suite tolerance, variant $v";  python tools/fuzz_more.py 2000 400 2>&1 | tail -1
done
cp /tmp/libmbavo_good.so mba-vo_amd/libmbavo.so
