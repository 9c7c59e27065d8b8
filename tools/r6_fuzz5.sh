#!/bin/bash
# Fourth campaign of round 6: after the marked sweep (fuzz_adam carries the "marked" schedule; fuzz_sharded / fuzz_train at their
# own sizes keep the parked one), then the default bench line of the tree.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_fuzz5
mkdir -p $O
timeout 400 python tools/fuzz_adam.py 300 612 > $O/adam.txt 2>&1; tail -1 $O/adam.txt
timeout 400 python tools/fuzz_train.py 240 613 > $O/train.txt 2>&1; tail -1 $O/train.txt
timeout 500 python tools/fuzz_sharded.py 400 614 > $O/sharded.txt 2>&1; tail -1 $O/sharded.txt
timeout 300 python tools/fuzz_encoder.py 120 615 > $O/encoder.txt 2>&1; tail -1 $O/encoder.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
