cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "submit_wait or concurrent" > gpurun_out/t_dist.log 2>&1; tail -5 gpurun_out/t_dist.log
