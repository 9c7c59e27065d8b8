#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r6_pytest22.txt 2>&1; tail -3 gpurun_out/r6_pytest22.txt
python tools/bench_emulated_world.py 8 P > gpurun_out/r6_emu_slab4.txt 2>&1; tail -7 gpurun_out/r6_emu_slab4.txt
timeout 900 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_models.py -x -q > gpurun_out/r6_pytest22b.txt 2>&1; tail -3 gpurun_out/r6_pytest22b.txt
