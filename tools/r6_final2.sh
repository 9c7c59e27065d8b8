#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py > gpurun_out/r6_final_bench_$1.json 2> gpurun_out/r6_final_bench_$1.err; echo "bench rc=$?"
timeout 700 python tools/fuzz_sharded_mips.py 600 7$1 > gpurun_out/r6_fuzz2_sharded_mips_$1.txt 2>&1; tail -1 gpurun_out/r6_fuzz2_sharded_mips_$1.txt
timeout 700 python tools/fuzz_sharded.py 600 8$1 > gpurun_out/r6_fuzz2_sharded_$1.txt 2>&1; tail -1 gpurun_out/r6_fuzz2_sharded_$1.txt
timeout 400 python tools/fuzz_train.py 300 9$1 > gpurun_out/r6_fuzz2_train_$1.txt 2>&1; tail -1 gpurun_out/r6_fuzz2_train_$1.txt
timeout 300 python tools/fuzz_encoder.py 200 5$1 > gpurun_out/r6_fuzz2_encoder_$1.txt 2>&1; tail -1 gpurun_out/r6_fuzz2_encoder_$1.txt
