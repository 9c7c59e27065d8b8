#!/bin/bash
# Round-2 evidence run (one gpurun call): kernel-trace stats of the default bench command, the two PMC passes for the
# sweep's HBM traffic (separate runs, as MI355X_MICROARCH.md prescribes), and the default bench line itself.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_P -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/stats_P.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_write.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_adam_sweep.csv $OUT/pmc_traffic_new.json > $OUT/pmc_summary.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mips -- python $R/tools/bench_mips.py > $OUT/stats_mips.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C3 -- python $R/bench.py --workload C3 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats_C3.log 2>&1
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
ls -laR $OUT | head -60
