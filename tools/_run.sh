cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_tn" 2>&1 | tail -2
TT_GEMM_NO_TN_STREAM=1 timeout 600 python bench.py --workload C3 --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('generic', d['ms_per_step'])"
timeout 600 python bench.py --workload C3 --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream ', d['ms_per_step'])"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3a/prof_c3 -- python $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
f=$(find gpurun_out/r3a/prof_c3 -name "*kernel_trace.csv" | sort | tail -1)
python tools/timeline.py $f 0 > gpurun_out/r3a/c3_timeline.txt
find gpurun_out/r3a/prof_c3 -name "*kernel_trace.csv" -size +3M -delete; find gpurun_out/r3a -name "*.db" -delete
grep "tn_stream" gpurun_out/r3a/c3_timeline.txt
