cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_sharded.py tests/test_gpu_reference_suite.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 30 --warmup 70 --no-cpu-baseline > $OUT/c3.log 2>&1
T=$(find $OUT/prof_c3 -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/timeline.py $T 0 > $OUT/c3_timeline.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete
grep '^{' $OUT/c3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'])"
