#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-secondary --spinup 0"
for rep in 1 2 3; do
  $B --workload C3 --steps 60 --warmup 90 2>/dev/null | tail -1 > gpurun_out/r6_aux_C3_norm_$rep.json
  TT_AUX_HIGH=1 $B --workload C3 --steps 60 --warmup 90 2>/dev/null | tail -1 > gpurun_out/r6_aux_C3_high_$rep.json
done
for rep in 1 2; do
  $B --workload C2 --steps 100 --warmup 90 2>/dev/null | tail -1 > gpurun_out/r6_aux_C2_norm_$rep.json
  TT_AUX_HIGH=1 $B --workload C2 --steps 100 --warmup 90 2>/dev/null | tail -1 > gpurun_out/r6_aux_C2_high_$rep.json
done
python tools/bench_emulated_world.py 8 P 2>&1 | grep "emulated W=8" > gpurun_out/r6_aux_emu_norm.txt
TT_AUX_HIGH=1 python tools/bench_emulated_world.py 8 P 2>&1 | grep "emulated W=8" > gpurun_out/r6_aux_emu_high.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_aux_C*.json")):
    try:
        p=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(p["ms_per_step"],4))
    except Exception as e: print(f,"ERR",e)
PY
cat gpurun_out/r6_aux_emu_*.txt
timeout 900 python -m pytest tests/test_gpu_parallel.py -x -q -k "hist50_dup or overlap or host_timing or issued" > gpurun_out/r6_pytest10.txt 2>&1; tail -3 gpurun_out/r6_pytest10.txt
