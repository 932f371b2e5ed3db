#!/bin/bash
# builds tests/harness/harness_bin against the in-tree libmbavo.so (gfx950)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -I ../../mba-vo_amd/csrc harness.cpp \
    -L ../../mba-vo_amd -lmbavo -Wl,-rpath,'$ORIGIN/../../mba-vo_amd' -o harness_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../mba-vo_amd/csrc -I ../../include -x hip solver_check.hip \
    ../../mba-vo_amd/csrc/host_math.cpp -o solver_check_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../mba-vo_amd/csrc -I ../../include div_check.hip -o div_check_bin
