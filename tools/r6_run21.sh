#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/bench_emulated_world.py 8 P > gpurun_out/r6_emu_phases.txt 2>&1; tail -9 gpurun_out/r6_emu_phases.txt
