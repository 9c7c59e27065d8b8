cd $GRAFT_REPO_ROOT
TT_TUNE_DEBUG=1 python tools/bench_emulated_world.py 8 P 2>&1 | grep "\[tt\]\|ms/step per\|ce_fwd_du_kernel:\|ce_bwd_kept_kernel:" | cut -c1-170
TT_TUNE_DEBUG=1 python tools/bench_emulated_world.py 4 P 2>&1 | grep "\[tt\]\|ms/step per" | cut -c1-170
TT_TUNE_DEBUG=1 python tools/bench_emulated_world.py 8 C3 2>&1 | grep "\[tt\]\|ms/step per" | cut -c1-170
timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
