cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
timeout 600 python bench.py --workload C2 --steps 200 --warmup 80 --no-cpu-baseline --no-secondary 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2 fused', d['ms_per_step'])"
TT_CE_NO_FUSED_LOSS=1 timeout 600 python bench.py --workload C2 --steps 200 --warmup 80 --no-cpu-baseline --no-secondary 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2 two-op', d['ms_per_step'])"
done
