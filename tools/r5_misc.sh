#!/bin/bash
# Round-5 side measurements (through gpurun from the repo root): the hit / miss mix microbenchmark, k_expand's -DBMQ_EXP_K=320
# -DBMQ_EXP_MIN_WAVES=5 build on C2 / C4, the compaction leg at two chunk sizes.  Results under gpurun_out/r05/misc.
O=gpurun_out/r05/misc
mkdir -p $O
timeout 120 build/ubench_mix > $O/ubench_mix.txt 2>&1; cat $O/ubench_mix.txt
P="--no-cpu-baseline --no-host-path --no-extras --steps 10 --warmup 3 --batcher-threads 0"
for w in c2 c4; do
  for v in default k320; do
    if [ $v = default ]; then unset BMQ_LIB; else export BMQ_LIB=$PWD/build/variants/libbmq_$v.so; fi
    timeout 200 python bench.py --workload $w $P --no-churn > $O/${w}_$v.json 2> $O/${w}_$v.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/${w}_$v.json").read().strip().splitlines()[-1])
    print("$w $v", round(d["value"] / 1e6, 2), d["unit"], {k: round(x, 4) for k, x in d["kernel_ms"].items()})
except Exception as ex:
    print("$w $v unreadable", ex)
PY
  done
done
unset BMQ_LIB
for ch in 65536 524288; do
  timeout 300 python bench.py --no-cpu-baseline --no-host-path --steps 10 --warmup 3 --batcher-threads 0 --compact-chunk $ch > $O/compact_$ch.json 2> $O/compact_$ch.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/compact_$ch.json").read().strip().splitlines()[-1])
    c = d["extra"]["compaction"]
    print("chunk $ch", {k: c[k] for k in ("batch_ms_idle", "batch_ms_while_compacting", "p99_ratio", "poll_ms", "polls", "build_s", "swap_ms") if k in c} if "error" not in c else c)
except Exception as ex:
    print("chunk $ch unreadable", ex)
PY
done
