#!/bin/bash
# quick A/B of env settings on the C3 bench: tools/ab_quick.sh name1 "ENV=.. ENV=.." name2 "..." -> gpurun_out/r04/abq.txt
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
out=$O/abq.txt; : > $out
while [ $# -gt 1 ]; do name=$1; envs=$2; shift 2
  r=$(env $envs python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-extras --batcher-threads 0 2>$O/abq_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms'], 'step', round(d['ms_per_step'],4))")
  echo "$name: $r" >> $out; grep "k_walk\|census" $O/abq_$name.err | tail -2 >> $out
done
cat $out
