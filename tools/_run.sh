cd $GRAFT_REPO_ROOT
python tools/bench_attn.py 2>&1 | grep attention
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "encoder or history or known_answer or wide" 2>&1 | tail -2
timeout 600 python bench.py --workload C3 --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'])"
