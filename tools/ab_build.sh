#!/bin/bash
# Build variant libraries of the engine for an A/B timing on the GPU box (which differs run to run by +-3 %, and
# drifts within a run: variants are only comparable interleaved inside one call).
# usage (here, no GPU needed): bash tools/ab_build.sh name1 "flags1" name2 "flags2" ...   -> tools/_ab/libmbavo_<name>.so
# then on the GPU box:          bash tools/ab_run.sh [rounds] [bench args]
cd "$(dirname "$0")/.."
mkdir -p tools/_ab; rm -f tools/_ab/*.so
bash mba-vo_amd/build.sh > /dev/null
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize ${OPT:--O3} $flags -c mba-vo_amd/csrc/engine.hip -o tools/_ab/engine_$name.o 2>/dev/null \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/libmbavo_$name.so tools/_ab/engine_$name.o $(ls mba-vo_amd/build/*.o | grep -v engine) -ldl \
    && rm tools/_ab/engine_$name.o && echo "built $name ($flags)" || echo "FAILED $name" ) &
done
wait
