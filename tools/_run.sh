cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "checkpoint_adaptor" 2>&1 | tail -15
