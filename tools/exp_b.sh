mkdir -p gpurun_out/x2
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/x2/$name.json 2> gpurun_out/x2/$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/x2/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value']/1e9,3), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'slow', d['slow_path_topics_per_batch'])
PY
  grep "k_walk waves" gpurun_out/x2/$name.err | tail -1
}
run ilp1 BMQ_WALK_ILP=1 BMQ_DEBUG=2
run ilp2_q512 BMQ_WALK_ILP=2 BMQ_QCAP=512 BMQ_DEBUG=2
run ilp2_q384 BMQ_WALK_ILP=2 BMQ_QCAP=384 BMQ_DEBUG=2
run ilp1_q192_p160 BMQ_WALK_ILP=1 BMQ_QCAP=192 BMQ_PCAP=160 BMQ_DEBUG=2
run ilp1_q128_p128 BMQ_WALK_ILP=1 BMQ_QCAP=128 BMQ_PCAP=128 BMQ_DEBUG=2
run ilp1_plain BMQ_WALK_ILP=1
