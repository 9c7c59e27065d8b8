#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "tower" > gpurun_out/r6_pytest15a.txt 2>&1; tail -3 gpurun_out/r6_pytest15a.txt
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_reference_suite.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r6_pytest15b.txt 2>&1; tail -3 gpurun_out/r6_pytest15b.txt
B="python bench.py --no-cpu-baseline --no-secondary --spinup 0"
$B --adam lazy --graph --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r6_lazyg_pair.json
TT_TOWERS_SERIAL=1 $B --adam lazy --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r6_lazy_serial_pair.json
$B --adam lazy --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r6_lazy_fork.json
$B --workload tiny --steps 400 --warmup 100 2>/dev/null | tail -1 > gpurun_out/r6_tiny_pair.json
$B --workload tiny --graph --steps 400 --warmup 20 2>/dev/null | tail -1 > gpurun_out/r6_tiny_graph_pair.json
python - <<'PY'
import json,glob
for f in ("r6_lazyg_pair","r6_lazy_serial_pair","r6_lazy_fork","r6_tiny_pair","r6_tiny_graph_pair"):
    try:
        p=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, round(p["ms_per_step"],4), round(p["value"]))
    except Exception as e: print(f,"ERR",e)
PY
