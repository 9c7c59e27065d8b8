cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -2
for g in 0 1; do
TT_SWEEP_GATE=$g timeout 600 python bench.py --workload C3 --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 gate=$g', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_c3g -- python $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
f=$(find gpurun_out/r3c/prof_c3g -name "*kernel_trace.csv" | sort | tail -1)
python tools/timeline.py $f 25 > gpurun_out/r3c/c3_timeline_gate.txt
find gpurun_out/r3c/prof_c3g -name "*kernel_trace.csv" -size +3M -delete; find gpurun_out/r3c -name "*.db" -delete
cat gpurun_out/r3c/c3_timeline_gate.txt
