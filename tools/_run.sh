cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python tools/bench_emulated_world.py 8 P 2>&1 | grep "ms/step per\|phases" | cut -c1-200
python tools/bench_emulated_world.py 8 C3 2>&1 | grep "ms/step per" | cut -c1-200
timeout 200 python tools/fuzz_sharded.py 100 35 2>&1 | grep "FINDING\|cases,"
