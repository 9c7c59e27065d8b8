#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-secondary"
$B --spinup 0 --adam lazy --graph --steps 40 --warmup 10 2> gpurun_out/r6_lazyg_pair.err | tail -1 > gpurun_out/r6_lazyg_pair.json; tail -3 gpurun_out/r6_lazyg_pair.err
for rep in 1 2 3; do
  $B --spinup 0 --workload C2 --steps 100 --warmup 90 2>/dev/null | tail -1 > gpurun_out/r6_pairAB_C2_fork_$rep.json
  TT_TOWERS_SERIAL=1 $B --spinup 0 --workload C2 --steps 100 --warmup 90 2>/dev/null | tail -1 > gpurun_out/r6_pairAB_C2_pair_$rep.json
done
for rep in 1 2; do
  $B 2>/dev/null | tail -1 > gpurun_out/r6_pairAB_P_fork_$rep.json
  TT_TOWERS_SERIAL=1 $B 2>/dev/null | tail -1 > gpurun_out/r6_pairAB_P_pair_$rep.json
  $B --spinup 0 --adam lazy --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r6_pairAB_lazy_fork_$rep.json
  TT_TOWERS_SERIAL=1 $B --spinup 0 --adam lazy --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r6_pairAB_lazy_pair_$rep.json
done
python - <<'PY'
import json,glob
for f in ["gpurun_out/r6_lazyg_pair.json"]+sorted(glob.glob("gpurun_out/r6_pairAB_*.json")):
    try:
        p=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(p["ms_per_step"],4), round(p["value"]))
    except Exception as e: print(f,"ERR",e)
PY
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "graph" > gpurun_out/r6_pytest16.txt 2>&1; tail -3 gpurun_out/r6_pytest16.txt
