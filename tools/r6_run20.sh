#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/emu_front_probe.py 8 > gpurun_out/r6_emu_front.txt 2>&1; cat gpurun_out/r6_emu_front.txt | tail -20
