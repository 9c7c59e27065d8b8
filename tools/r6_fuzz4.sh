#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1000 python tools/fuzz_sharded.py 900 201 > gpurun_out/r6_fuzz4_sharded.txt 2>&1; tail -1 gpurun_out/r6_fuzz4_sharded.txt
timeout 700 python tools/fuzz_sharded_mips.py 600 202 > gpurun_out/r6_fuzz4_sharded_mips.txt 2>&1; tail -1 gpurun_out/r6_fuzz4_sharded_mips.txt
timeout 400 python tools/fuzz_train.py 300 203 > gpurun_out/r6_fuzz4_train.txt 2>&1; tail -1 gpurun_out/r6_fuzz4_train.txt
timeout 300 python tools/fuzz_ce16.py 200 204 > gpurun_out/r6_fuzz4_ce16.txt 2>&1; tail -1 gpurun_out/r6_fuzz4_ce16.txt
