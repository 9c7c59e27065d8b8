#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python tools/fuzz_train.py 300 101 > gpurun_out/r6_fuzz3_train.txt 2>&1; tail -1 gpurun_out/r6_fuzz3_train.txt
timeout 300 python tools/fuzz_graph.py 200 111 > gpurun_out/r6_fuzz3_graph.txt 2>&1; tail -1 gpurun_out/r6_fuzz3_graph.txt
timeout 200 python tools/fuzz_adam.py 120 121 > gpurun_out/r6_fuzz3_adam.txt 2>&1; tail -1 gpurun_out/r6_fuzz3_adam.txt
