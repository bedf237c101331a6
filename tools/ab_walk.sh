#!/bin/bash
# A/B of k_walk build variants on the GPU box: bench.py (C3, kernels only) once per variant library (BMQ_LIB) -> gpurun_out/ab_walk.txt
out=gpurun_out/ab_walk.txt
: > $out
run() { # name, env...
  name=$1; shift
  r=$(env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-extras 2>gpurun_out/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms'], 'step', round(d['ms_per_step'],4), 'novt', round(d['ms_per_step_without_kernel_timing'],4), 'fanout', (d.get('fanout_group') or {}).get('ms'))")
  echo "$name: $r" >> $out
  grep "k_walk" gpurun_out/ab_$name.err | tail -2 >> $out
}
for v in "$@"; do run $v BMQ_LIB=$PWD/build/variants/libbmq_$v.so; run ${v}_dbg BMQ_LIB=$PWD/build/variants/libbmq_$v.so BMQ_DEBUG=2; done
run default X=1
run default_dbg BMQ_DEBUG=2
cat $out
