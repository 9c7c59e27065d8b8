#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6_final_pytest_$1.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r6_final_pytest_$1.txt | tail -2
python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 300 python tools/fuzz_train.py 200 10$1 > gpurun_out/r6_fuzz3_train_$1.txt 2>&1; tail -1 gpurun_out/r6_fuzz3_train_$1.txt
timeout 200 python tools/fuzz_graph.py 120 11$1 > gpurun_out/r6_fuzz3_graph_$1.txt 2>&1; tail -1 gpurun_out/r6_fuzz3_graph_$1.txt
timeout 200 python tools/fuzz_adam.py 100 12$1 > gpurun_out/r6_fuzz3_adam_$1.txt 2>&1; tail -1 gpurun_out/r6_fuzz3_adam_$1.txt
python bench.py > gpurun_out/r6_final_bench_$1.json 2> gpurun_out/r6_final_bench_$1.err; echo "bench rc=$?"
