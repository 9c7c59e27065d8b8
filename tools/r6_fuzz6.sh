#!/bin/bash
# Fifth campaign: sharded training with the marked sweep in half of the eligible cases; two more default bench lines of the final tree.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_fuzz6
mkdir -p $O
timeout 500 python tools/fuzz_sharded.py 400 616 > $O/sharded.txt 2>&1; tail -1 $O/sharded.txt; grep -c mark_from $O/sharded.txt
timeout 300 python tools/fuzz_adam.py 200 617 > $O/adam.txt 2>&1; tail -1 $O/adam.txt
python bench.py > $O/bench_default_a.json 2> $O/bench_default_a.err; cut -c1-200 $O/bench_default_a.json
