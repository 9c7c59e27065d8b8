#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2 3; do
for w in 768 512 640; do
  TT_SWEEP_WGS=$w timeout 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_width2_${w}_$rep.json
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_width2_*.json")):
    try:
        p=json.loads(open(f).read().strip().splitlines()[-1]); r=p["roofline"]
        print(f, round(p["ms_per_step"],3), r["frac"], r["avg_launch_ms"], r.get("hbm_copy_GBps"), r.get("frac_of_copy"), r["table_arena"]["candidates_GBps"], r["table_arena"]["kept"])
    except Exception as e: print(f, "ERR", e)
PY
