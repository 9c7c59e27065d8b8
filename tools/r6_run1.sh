#!/bin/bash
# round 6, first contact: placement experiment FIRST (fresh box, nothing else has touched the device), then the GPU suite,
# then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/sweep_placement.py --big > gpurun_out/r6_placement_$1.txt 2>&1
echo "placement rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest_$1.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r6_pytest_$1.txt
timeout 900 python bench.py > gpurun_out/r6_bench_$1.json 2> gpurun_out/r6_bench_$1.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r6_bench_$1.err
