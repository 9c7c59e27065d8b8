#!/bin/bash
# Final-tree evidence after the marked sweep: the whole GPU suite, smoke, the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_final4
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-260 $O/bench_default.json
