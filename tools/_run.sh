cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "trajectory" 2>&1 | tail -4
