#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/sweep_random_placement.py 40 > gpurun_out/r6_randplace_$1.txt 2>&1
echo "randplace rc=$?"; tail -4 gpurun_out/r6_randplace_$1.txt
timeout 1200 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_switches.py tests/test_gpu_bench_contract.py -x -q > gpurun_out/r6_pytest3_$1.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r6_pytest3_$1.txt
