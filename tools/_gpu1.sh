cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_dist_gpu.py tests/test_retain_gpu.py -x -q -m gpu -k "not full_size" > gpurun_out/t_dist.log 2>&1; tail -3 gpurun_out/t_dist.log
for lib in libbmq.so libbmq_short4.so libbmq_short16.so; do for wl in c3 c2 c4; do BMQ_LIB=$PWD/bifromq_amd/$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-host-path --steps 20 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $wl', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})"; done; done
