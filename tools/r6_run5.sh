#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for w in 768 512 384 640; do
  TT_SWEEP_WGS=$w timeout 300 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r6_width_$1_${w}_$rep.json 2>/dev/null
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_width_%s_*.json" % "$1")):
    try:
        p=json.loads(open(f).read().strip().splitlines()[-1]); r=p["roofline"]
        print(f, round(p["ms_per_step"],3), r["frac"], r["avg_launch_ms"], r.get("hbm_copy_GBps"), r.get("frac_of_copy"))
    except Exception as e: print(f, "ERR", e)
PY
