#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2 3 4; do
  timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r6_tour_$1_on_$rep.json 2>/dev/null
  TT_ADAM_ARENA_TRIES=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r6_tour_$1_off_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_tour_%s_*.json" % "$1")):
    try:
        p=json.loads(open(f).read().strip().splitlines()[-1]); r=p["roofline"]
        print(f, round(p["ms_per_step"],3), r["frac"], r["avg_launch_ms"], r.get("hbm_copy_GBps"), r.get("frac_of_copy"), r.get("table_arena"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_models.py -x -q > gpurun_out/r6_pytest6_$1.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r6_pytest6_$1.txt
