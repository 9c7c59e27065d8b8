cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "wide_embeddings or trajectory" 2>&1 | tail -15 > gpurun_out/r3b/t_models.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3_whole or p_shape or c2_whole" 2>&1 | tail -15 > gpurun_out/r3b/t_full.log
(time timeout 900 python bench.py) > gpurun_out/r3b/bench_default.log 2>&1
cat gpurun_out/r3b/t_models.log gpurun_out/r3b/t_full.log; tail -5 gpurun_out/r3b/bench_default.log | cut -c1-3000
