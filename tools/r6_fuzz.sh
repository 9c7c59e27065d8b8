#!/bin/bash
# round 6: smoke() + the differential fuzzers over everything this round touched (routing jobs, arena, sharded MIPS with first-try k')
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python __graft_entry__.py --smoke > gpurun_out/r6_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r6_smoke.txt
timeout 400 python tools/fuzz_sharded_mips.py 300 61 > gpurun_out/r6_fuzz_sharded_mips.txt 2>&1; echo "fuzz_sharded_mips rc=$?"; tail -2 gpurun_out/r6_fuzz_sharded_mips.txt
timeout 500 python tools/fuzz_sharded.py 360 62 > gpurun_out/r6_fuzz_sharded.txt 2>&1; echo "fuzz_sharded rc=$?"; tail -2 gpurun_out/r6_fuzz_sharded.txt
timeout 300 python tools/fuzz_train.py 200 63 > gpurun_out/r6_fuzz_train.txt 2>&1; echo "fuzz_train rc=$?"; tail -2 gpurun_out/r6_fuzz_train.txt
timeout 200 python tools/fuzz_adam.py 120 64 > gpurun_out/r6_fuzz_adam.txt 2>&1; echo "fuzz_adam rc=$?"; tail -2 gpurun_out/r6_fuzz_adam.txt
timeout 200 python tools/fuzz_graph.py 90 65 > gpurun_out/r6_fuzz_graph.txt 2>&1; echo "fuzz_graph rc=$?"; tail -2 gpurun_out/r6_fuzz_graph.txt
timeout 200 python tools/fuzz_mips.py 90 66 > gpurun_out/r6_fuzz_mips.txt 2>&1; echo "fuzz_mips rc=$?"; tail -2 gpurun_out/r6_fuzz_mips.txt
