cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_retain_gpu.py tests/test_batcher_gpu.py tests/test_range.py -x -q -m gpu -k "not full_size" > gpurun_out/t_dist.log 2>&1; tail -5 gpurun_out/t_dist.log
