cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_loss_head or weighted_mean" 2>&1 | tail -12
