#!/bin/bash
# Round-6 evidence run (one gpurun call): the GPU suite, then kernel-trace stats + timelines of the default bench command
# (P), C3, the emulated W = 8 step (train, module path), the emulated W = 8 config-5 search, the single-GPU MIPS (20 warm
# calls per dtype); PMC passes (separate runs, --pmc with --kernel-trace only) for the sweep's HBM traffic; `bench.py
# --workload C5`; the default bench line itself.  Summaries land in gpurun_out/r6prof; the quoted ones are copied to profiles/r06_*.
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6prof
mkdir -p $OUT
cd $R && timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_P -- $B --steps 10 --warmup 3 > $OUT/stats_P.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B --spinup 0 --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B --spinup 0 --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_adam_sweep.csv $OUT/pmc_traffic_new.json > $OUT/pmc_summary.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C3 -- $B --workload C3 --steps 30 --warmup 70 > $OUT/stats_C3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_emu -- python $R/tools/bench_emulated_world.py 8 P > $OUT/stats_emu.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_emuC5 -- python $R/tools/bench_emulated_world.py 8 C5 > $OUT/stats_emuC5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mips -- python $R/tools/bench_mips.py > $OUT/stats_mips.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C5 -- python $R/bench.py --workload C5 --steps 10 --warmup 3 > $OUT/stats_C5.log 2>&1
for n in P C3 emu emuC5 mips C5; do
  T=$(find $OUT/stats_$n -name "*kernel_trace.csv" | head -1)
  case $n in P|C3|emu) python $R/tools/timeline.py $T 0 > $OUT/timeline_$n.txt 2>&1;; esac
  S=$(find $OUT/stats_$n -name "*kernel_stats.csv" | head -1)
  cp $S $OUT/kernel_stats_$n.csv
  grep '^{"metric"' $OUT/stats_$n.log | tail -1 > $OUT/bench_profiled_$n.json
done
cd $R
python tools/bench_emulated_world.py 8 P > $OUT/emulated_W8.txt 2>&1
python tools/bench_emulated_world.py 8 C5 > $OUT/emulated_W8_C5.txt 2>&1
python bench.py --workload C5 --steps 10 --warmup 3 > $OUT/bench_C5_1gpu.json 2> $OUT/bench_C5_1gpu.err
python bench.py --workload C5 --gpus 2 --steps 5 --warmup 2 > $OUT/bench_C5_2ranks_gloo.json 2> $OUT/bench_C5_2ranks_gloo.err
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
find $OUT -name "*.db" -delete
ls -la $OUT | head -80
