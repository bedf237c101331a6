cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_zz_churn_gpu.py tests/test_batcher_gpu.py -x -q -m gpu > gpurun_out/t_dist.log 2>&1; tail -5 gpurun_out/t_dist.log
BMQ_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --steps 5 2>&1 | grep -E "bmq index|metric" | cut -c1-300 > gpurun_out/bench_c3.log; cat gpurun_out/bench_c3.log
BMQ_TIMING=1 timeout 600 python bench.py --churn 100000 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_churn.log 2>&1; grep -E "apply:" gpurun_out/bench_churn.log | tail -8; tail -1 gpurun_out/bench_churn.log | cut -c1-300; tail -1 gpurun_out/bench_churn.log | grep -o '"churn".*"kernel_ms"' | cut -c1-200
timeout 600 python bench.py --churn 100000 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
