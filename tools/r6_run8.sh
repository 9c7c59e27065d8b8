#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_emulated_world.py 8 P > gpurun_out/r6_emu_wgrad_aside_$1.txt 2>&1; tail -12 gpurun_out/r6_emu_wgrad_aside_$1.txt
timeout 1500 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_bench_contract.py tests/test_gpu_switches.py tests/test_gpu_models.py -x -q > gpurun_out/r6_pytest8_$1.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r6_pytest8_$1.txt
python bench.py > gpurun_out/r6_bench_default_$1.json 2> gpurun_out/r6_bench_default_$1.err; echo "bench rc=$?"
