#!/bin/bash
# interleaved A/B of library variants on trackFrame (tools/vo_bench.py) and the single-problem latency (tools/latency.py)
cd "$(dirname "$0")/.."
cp mba-vo_amd/libmbavo.so /tmp/libmbavo_good.so
for r in 1 2 3; do
  for f in tools/_ab/libmbavo_*.so; do
    v=$(basename $f .so); v=${v#libmbavo_}
    cp $f mba-vo_amd/libmbavo.so
    echo "$v vo $(python tools/vo_bench.py 8 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['gpu_ms_per_frame'],4))")"
    [ $r = 1 ] && env $LAT_ENV python tools/latency.py 2>&1 | tail -4 | sed "s/^/$v /"
  done
done
cp /tmp/libmbavo_good.so mba-vo_amd/libmbavo.so
