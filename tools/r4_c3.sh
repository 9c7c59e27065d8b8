#!/bin/bash
# Round-4 C3 evidence (one gpurun call): kernel-trace stats + timeline of the C3 step, the step without the optimiser,
# and optionally the PMC traffic passes (PMC=1).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4c3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary --workload C3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C3 -- $B --steps 30 --warmup 70 > $OUT/stats_C3.log 2>&1
T=$(find $OUT/stats_C3 -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline.py $T 0 > $OUT/timeline_C3.txt 2>&1
cp $(find $OUT/stats_C3 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_C3.csv
if [ -n "$PMC" ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c3_fetch -- $B --steps 4 --warmup 2 > $OUT/pmc_c3_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c3_write -- $B --steps 4 --warmup 2 > $OUT/pmc_c3_write.log 2>&1
  python $R/tools/pmc_step_traffic.py $OUT/pmc_c3_fetch $OUT/pmc_c3_write $OUT/pmc_c3_step_traffic.json 6.4 > $OUT/pmc_c3_step_traffic.txt 2>&1
fi
cd $R
python bench.py --no-cpu-baseline --no-secondary --workload C3 --steps 40 --warmup 80 2>/dev/null | tail -1 > $OUT/bench_C3.json
python bench.py --no-cpu-baseline --no-secondary --workload C3 --phase fwdbwd --steps 40 --warmup 20 2>/dev/null | tail -1 > $OUT/bench_C3_fwdbwd.json
python bench.py --no-cpu-baseline --no-secondary --workload C3 --phase fwd --steps 40 --warmup 20 2>/dev/null | tail -1 > $OUT/bench_C3_fwd.json
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
find $OUT -name "*.db" -delete
cat $OUT/timeline_C3.txt
python - <<PY
import json
for n in ("bench_C3","bench_C3_fwdbwd","bench_C3_fwd"):
    try:
        d=json.load(open("$OUT/%s.json"%n)); print(n, d["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
