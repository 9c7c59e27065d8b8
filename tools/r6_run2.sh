#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o /tmp/copy_probe && timeout 300 /tmp/copy_probe > gpurun_out/r6_copy_probe_$1.txt 2>&1
echo "copy probe rc=$?"
timeout 600 python tools/sweep_offsets2.py > gpurun_out/r6_offsets2_$1.txt 2>&1
echo "offsets rc=$?"
timeout 900 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_switches.py tests/test_gpu_bench_contract.py -x -q > gpurun_out/r6_pytest2_$1.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r6_pytest2_$1.txt
