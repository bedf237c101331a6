mkdir -p gpurun_out/x4
python -m pytest tests -m gpu -x -q > gpurun_out/x4/pytest.log 2>&1; tail -3 gpurun_out/x4/pytest.log
run() { # name, workload args..., -- env
  name=$1; shift; wl=$1; shift
  env "$@" python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/x4/$name.json 2> gpurun_out/x4/$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/x4/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value']/1e9,4), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'slow', d.get('slow_path_topics_per_batch'))
PY
  grep "k_walk waves" gpurun_out/x4/$name.err | tail -1
}
run c3 c3 BMQ_X=0
run c3_dbg c3 BMQ_DEBUG=2
run c3_q128 c3 BMQ_QCAP=128 BMQ_PCAP=128 BMQ_DEBUG=2
run c2 c2 BMQ_DEBUG=2
run c4 c4 BMQ_X=0
