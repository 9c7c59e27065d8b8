#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "plan or adam or tower" > gpurun_out/r6_pytest19a.txt 2>&1; tail -3 gpurun_out/r6_pytest19a.txt
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_parallel.py -x -q > gpurun_out/r6_pytest19b.txt 2>&1; tail -3 gpurun_out/r6_pytest19b.txt
B="python bench.py --no-cpu-baseline --no-secondary --spinup 0"
for rep in 1 2; do
$B --adam lazy --graph --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r6_lazyg_plans_$rep.json
$B --adam lazy --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r6_lazy_plans_$rep.json
$B --workload C2 --steps 100 --warmup 90 2>/dev/null | tail -1 > gpurun_out/r6_C2_plans_$rep.json
done
python tools/bench_emulated_world.py 8 P 2>&1 | grep -E "emulated W=8|main-stream" > gpurun_out/r6_emu_plans.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_*_plans_*.json")):
    try:
        p=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(p["ms_per_step"],4), round(p["value"]))
    except Exception as e: print(f,"ERR",e)
PY
cat gpurun_out/r6_emu_plans.txt
