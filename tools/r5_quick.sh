#!/bin/bash
# quick look: timelines of the C3 step and of the graphed deferred-Adam P step
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C3 -- $B --workload C3 --steps 30 --warmup 70 > $OUT/stats_C3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lazyg -- $B --adam lazy --graph --steps 30 --warmup 10 > $OUT/stats_lazyg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lazy -- $B --adam lazy --steps 30 --warmup 10 > $OUT/stats_lazy.log 2>&1
for n in C3 lazyg lazy; do
  T=$(find $OUT/stats_$n -name "*kernel_trace.csv" | head -1)
  python $R/tools/timeline.py $T 0 > $OUT/timeline_$n.txt 2>&1
  S=$(find $OUT/stats_$n -name "*kernel_stats.csv" | head -1)
  cp $S $OUT/kernel_stats_$n.csv
  tail -1 $OUT/stats_$n.log | cut -c1-300 > $OUT/line_$n.txt
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
