cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
python tools/bench_gemm.py 2>&1 | grep -v amdgpu | head -4
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "encoder or history or known_answer" 2>&1 | tail -2
timeout 600 python bench.py --workload C3 --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'])"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3e/prof_c3 -- python $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
f=$(find gpurun_out/r3e/prof_c3 -name "*kernel_trace.csv" | sort | tail -1)
python tools/timeline.py $f 25 > gpurun_out/r3e/c3_timeline.txt
find gpurun_out/r3e/prof_c3 -name "*kernel_trace.csv" -size +3M -delete; find gpurun_out/r3e -name "*.db" -delete
cat gpurun_out/r3e/c3_timeline.txt
