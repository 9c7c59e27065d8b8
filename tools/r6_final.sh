#!/bin/bash
# the driver's round-end sequence on the final tree: GPU suite, smoke, default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6_final_pytest_$1.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r6_final_pytest_$1.txt | tail -2
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py > gpurun_out/r6_final_bench_$1.json 2> gpurun_out/r6_final_bench_$1.err; echo "bench rc=$?"
