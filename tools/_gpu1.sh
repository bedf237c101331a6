cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_batcher_gpu.py -x -q -m gpu > gpurun_out/t_b.log 2>&1; tail -5 gpurun_out/t_b.log
