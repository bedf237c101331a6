#!/bin/bash
# A/B of retain-walk settings on the C4 bench: tools/r5_c4_ab.sh name "ENV=.. ENV=.." [name2 "..."] -> gpurun_out/r05/c4_ab.txt
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
out=$O/c4_ab.txt
while [ $# -gt 1 ]; do name=$1; envs=$2; shift 2
  r=$(env $envs python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-extras --no-churn 2>$O/c4_ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms'], 'step', round(d['ms_per_step'],4), round(d['value']/1e6,1), 'M filters/s')")
  echo "$name [$envs]: $r" >> $out
done
cat $out
