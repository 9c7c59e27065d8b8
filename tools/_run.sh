cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tower" 2>&1 | tail -2
python tools/bench_emulated_world.py 8 P 2>&1 | grep "ms/step per\|phases" | cut -c1-200
timeout 600 python bench.py --workload C2 --steps 200 --warmup 80 --no-cpu-baseline --no-secondary 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2', d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/p.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/timeline.py $T 0 | grep "tower\|step span\|plan_small\|sweep"
grep '^{' $OUT/p.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('P', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete
