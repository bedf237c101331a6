#!/bin/bash
# A/B of library variants on the ordered-batch leg: tools/ab_ordered.sh name1 "ENV=.." name2 "ENV=.." ... -> gpurun_out/r06/ab_ordered.txt
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
out=$O/ab_ordered.txt; : > $out
while [ $# -gt 1 ]; do name=$1; envs=$2; shift 2
  env $envs timeout 600 python bench.py --ordered-only --no-cpu-baseline 2>$O/abo_$name.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d.get('ordered_batch') or d.get('extra',{}).get('ordered_batch') or d
print('$name:', '  '.join('%s walk %.4f expand %.4f step %.4f' % (k[:18], o[k]['kernel_ms']['k_walk'], o[k]['kernel_ms']['k_expand (+ k_fill_adj)'], o[k]['ms_per_step']) for k in ('as_generated','ordered_with_repeats','ordered_dedup_sorted','ordered_distinct')), o.get('ordered_dedup_sorted',{}).get('rows_equal_undeduplicated_engine'), o.get('ordered_distinct',{}).get('rows_equal_ordered_heads'))" >> $out
done
cat $out
