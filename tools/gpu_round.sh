#!/bin/bash
# One GPU-box session of a round: the -m gpu suite, the default bench line, the N > 1 code path on one GPU, the
# kernel-trace profile and the counter passes.  Everything lands under gpurun_out/ (copy what is to be judged into profiles/).
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/${TAG}_bench.json
MBAVO_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 50 > gpurun_out/${TAG}_bench_dist1.json 2> gpurun_out/${TAG}_bench_dist1.err; echo "bench dist1 rc=$?"; tail -c 1500 gpurun_out/${TAG}_bench_dist1.json
MBAVO_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 50 --workload c4_batch512 > gpurun_out/${TAG}_bench_dist1_c4.json 2> gpurun_out/${TAG}_bench_dist1_c4.err; echo "bench dist1 c4 rc=$?"; tail -c 800 gpurun_out/${TAG}_bench_dist1_c4.json
