#!/bin/bash
# round 4, k_expand rewrite, second A/B: the pipelined variants against variant a (kernel times on C3 / C2 / C4) + a short parity run
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
P="--no-cpu-baseline --no-host-path --no-extras --batcher-threads 0"
: > $O/ab2.txt
for v in e e5 g5; do
  lib=build/variants/$v/libbmq.so
  line="$v:"
  for w in c3 c2 c4; do
    steps=20; [ $w != c3 ] && steps=5
    r=$(BMQ_LIB=$lib timeout 120 python bench.py --workload $w --steps $steps --warmup 2 $P 2>$O/ab2_${v}_$w.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print(' '.join('%s=%.4f'%(n.replace('k_',''),x) for n,x in k.items() if x is not None), 'step=%.4f'%d['ms_per_step'])" 2>&1 | tail -1)
    line="$line [$w $r]"
  done
  echo "$line" >> $O/ab2.txt
done
cat $O/ab2.txt
for v in e g5; do
BMQ_LIB=build/variants/$v/libbmq.so timeout 100 python -m pytest tests/test_dist_gpu.py tests/test_retain_gpu.py -x -q -k "random_parity or edge or deep" > $O/pytest_quick_$v.log 2>&1; grep -E "passed|failed|error" $O/pytest_quick_$v.log | tail -1
done
