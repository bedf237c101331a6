#!/bin/bash
# round 4, k_expand rewrite: quick parity with the in-tree library, then the four prebuilt variants on C3 / C2 / C4 (kernel times)
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
timeout 240 python -m pytest tests/test_dist_gpu.py tests/test_retain_gpu.py tests/test_formats_gpu.py -x -q -k "not full_size" > $O/pytest_quick.log 2>&1
tail -3 $O/pytest_quick.log
P="--no-cpu-baseline --no-host-path --no-extras --batcher-threads 0"
: > $O/ab.txt
for v in a b c; do
  lib=build/variants/$v/libbmq.so; [ $v = base ] && lib=build/variants/libbmq_base.so
  line="$v:"
  for w in c3 c2 c4; do
    steps=20; [ $w != c3 ] && steps=5
    r=$(BMQ_LIB=$lib timeout 120 python bench.py --workload $w --steps $steps --warmup 2 $P 2>$O/ab_${v}_$w.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print(' '.join('%s=%.4f'%(n.replace('k_',''),x) for n,x in k.items() if x is not None), 'step=%.4f'%d['ms_per_step'])" 2>&1 | tail -1)
    line="$line [$w $r]"
  done
  echo "$line" >> $O/ab.txt
done
cat $O/ab.txt
BMQ_DEBUG=4 BMQ_LIB=build/variants/a/libbmq.so timeout 60 python bench.py --steps 3 --warmup 1 $P 2>&1 | grep "k_expand waves" | tail -1 | tee $O/expand_clocks_a.txt
