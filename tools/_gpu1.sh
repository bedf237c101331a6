cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "caps or kat" > gpurun_out/t_caps.log 2>&1; tail -15 gpurun_out/t_caps.log
