mkdir -p gpurun_out/x3
python -m pytest tests -m gpu -x -q > gpurun_out/x3/pytest.log 2>&1; tail -2 gpurun_out/x3/pytest.log
run() { # name, workload args..., -- env
  name=$1; shift; wl=$1; shift
  env "$@" python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/x3/$name.json 2> gpurun_out/x3/$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/x3/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value']/1e9,4), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'slow', d.get('slow_path_topics_per_batch'))
PY
  grep "k_walk waves" gpurun_out/x3/$name.err | tail -1
}
run c3 c3 BMQ_DEBUG=2
run c3_q192 c3 BMQ_QCAP=192 BMQ_PCAP=160
run c3_q192_p128 c3 BMQ_QCAP=192 BMQ_PCAP=128
run c3_xcd8 c3 BMQ_XCD_CHUNK=8
run c3_xcd32 c3 BMQ_XCD_CHUNK=32
run c3_q192_xcd16 c3 BMQ_QCAP=192 BMQ_PCAP=160 BMQ_XCD_CHUNK=16
run c2 c2 BMQ_X=0
run c2_q192 c2 BMQ_QCAP=192 BMQ_PCAP=160
run c4 c4 BMQ_X=0
