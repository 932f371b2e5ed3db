#!/bin/bash
# variant builds where the switch lives in a header several objects include: full library builds with extra flags
cd /root/repo
mkdir -p tools/_ab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  d=/tmp/abobj_$name; rm -rf $d; mkdir -p $d
  for f in engine.hip ba_tracker.hip image_ops.hip keyframe_ops.hip lm_batch.hip multi_gpu.hip; do
    extra=""; [ "$f" = engine.hip ] && extra="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra $flags -c mba-vo_amd/csrc/$f -o $d/$f.o 2>/dev/null &
  done
  for f in host_math.cpp tracker.cpp vo_frontend.cpp c_api.cpp; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -x hip -c mba-vo_amd/csrc/$f -o $d/$f.o 2>/dev/null &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/libmbavo_$name.so $d/*.o -ldl && echo "built $name ($flags)"
done
