mkdir -p gpurun_out/x1
python -m pytest tests -m gpu -x -q > gpurun_out/x1/pytest_ilp1.log 2>&1; tail -2 gpurun_out/x1/pytest_ilp1.log
BMQ_WALK_ILP=2 python -m pytest tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/x1/pytest_ilp2.log 2>&1; tail -2 gpurun_out/x1/pytest_ilp2.log
for v in 1 2; do BMQ_WALK_ILP=$v python bench.py --no-cpu-baseline > gpurun_out/x1/bench_ilp$v.json 2> gpurun_out/x1/bench_ilp$v.err; python - <<PY
import json
d=json.loads(open('gpurun_out/x1/bench_ilp$v.json').read().strip().splitlines()[-1])
print('ILP$v', d['value'], d['ms_per_step'], d.get('kernel_ms'))
PY
done
BMQ_DEBUG=1 python bench.py --no-cpu-baseline > gpurun_out/x1/bench_dbg1.json 2> gpurun_out/x1/bench_dbg1.err; tail -c 600 gpurun_out/x1/bench_dbg1.json
BMQ_WALK_ILP=2 BMQ_QCAP=192 BMQ_PCAP=256 python bench.py --no-cpu-baseline > gpurun_out/x1/bench_ilp2_q192.json 2>&1; tail -c 400 gpurun_out/x1/bench_ilp2_q192.json
