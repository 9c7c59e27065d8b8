cd $GRAFT_REPO_ROOT
timeout 600 python tools/bench_emulated_world.py 8 P 2>&1 | tail -12
