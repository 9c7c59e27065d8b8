#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/sweep_random_placement.py 12 > gpurun_out/r6_randplace_$1.txt 2>&1
echo "randplace rc=$?"; tail -3 gpurun_out/r6_randplace_$1.txt
for i in 1 2 3; do
  timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r6_ab_arena_$1_$i.json 2>/dev/null
  TT_ADAM_NO_ARENA=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r6_ab_noarena_$1_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_ab_*_%s_*.json" % "$1")):
    try:
        p=json.loads(open(f).read().strip().splitlines()[-1]); r=p["roofline"]
        print(f, round(p["ms_per_step"],3), r["frac"], r.get("hbm_copy_GBps"), r.get("frac_of_copy"), p.get("spinup_ms_per_step_by_block"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_pytest4_$1.txt 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r6_pytest4_$1.txt
