#!/bin/bash
# Round-5 evidence run (one gpurun call): kernel-trace stats + timelines of the default bench command (P), C2, C3, the
# graphed deferred-Adam step, the emulated W = 8 step (module path) and MIPS; PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes: --pmc with --kernel-trace only) for the sweep's HBM traffic (P) and the whole C3 step;
# the default bench line itself.  Summaries land in gpurun_out/r5prof; the ones quoted are copied to profiles/r05_*.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_P -- $B --steps 10 --warmup 3 > $OUT/stats_P.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B --spinup 0 --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B --spinup 0 --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_adam_sweep.csv $OUT/pmc_traffic_new.json > $OUT/pmc_summary.log 2>&1
for wl in C2 C3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- $B --workload $wl --steps 30 --warmup 70 > $OUT/stats_$wl.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lazyg -- $B --adam lazy --graph --steps 30 --warmup 10 > $OUT/stats_lazyg.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c3_fetch -- $B --spinup 0 --workload C3 --steps 4 --warmup 2 > $OUT/pmc_c3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_c3_write -- $B --spinup 0 --workload C3 --steps 4 --warmup 2 > $OUT/pmc_c3_write.log 2>&1
python $R/tools/pmc_step_traffic.py $OUT/pmc_c3_fetch $OUT/pmc_c3_write $OUT/pmc_c3_step_traffic.json 6.4 > $OUT/pmc_c3_step_traffic.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_emu -- python $R/tools/bench_emulated_world.py 8 P > $OUT/stats_emu.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mips -- python $R/tools/bench_mips.py > $OUT/stats_mips.log 2>&1
for n in P C2 C3 lazyg emu mips; do
  T=$(find $OUT/stats_$n -name "*kernel_trace.csv" | head -1)
  [ "$n" != mips ] && python $R/tools/timeline.py $T 0 > $OUT/timeline_$n.txt 2>&1
  S=$(find $OUT/stats_$n -name "*kernel_stats.csv" | head -1)
  cp $S $OUT/kernel_stats_$n.csv
  grep '^{"metric"' $OUT/stats_$n.log | tail -1 > $OUT/bench_profiled_$n.json
done
cd $R
python tools/bench_emulated_world.py 8 P > $OUT/emulated_W8.txt 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
find $OUT -name "*.db" -delete
ls -la $OUT | head -60
