#!/bin/bash
# Secondary measurements quoted in DESIGN.md section 5 (run through gpurun from the repo root; results under gpurun_out/<round>/extras).
R=${1:-r05}
O=gpurun_out/$R/extras
mkdir -p $O
for n in 10000 100000 4000000; do
  python bench.py --topics $n --steps 30 --warmup 5 --no-cpu-baseline --no-host-path > $O/sweep_$n.json 2> $O/sweep_$n.err
done
python bench.py --ungrouped --no-cpu-baseline --no-host-path > $O/ungrouped.json 2> $O/ungrouped.err
python bench.py --exchange-selftest --no-cpu-baseline --no-host-path > $O/exchange_selftest.json 2> $O/exchange_selftest.err
python bench.py --exchange-selftest --exchange-impl lib --no-cpu-baseline --no-host-path > $O/exchange_selftest_lib.json 2> $O/exchange_selftest_lib.err
# per-wave phase clocks of k_walk (BMQ_DEBUG=2) and k_expand (BMQ_DEBUG=4), C3: the kernels' experiments exist in -DBMQ_EXPERIMENTS=1 builds
# only (tools/build_variant.sh exp -DBMQ_EXPERIMENTS=1 -DBMQ_EXP_CLOCKS=1 -> build/variants/libbmq_exp.so, built before the GPU call)
export BMQ_LIB=$PWD/build/variants/libbmq_exp.so
(BMQ_DEBUG=2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>&1 >/dev/null | grep 'k_walk waves' | tail -1
 BMQ_DEBUG=4 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>&1 >/dev/null | grep 'k_expand waves' | tail -1) > $O/wave_clocks.txt
unset BMQ_LIB
# per-wave phase clocks of k_retain_walk, C4 (tools/build_variant.sh rwclk -DBMQ_RW_CLOCKS=1 -> build/variants/libbmq_rwclk.so)
BMQ_LIB=$PWD/build/variants/libbmq_rwclk.so python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-churn 2>&1 >/dev/null | grep 'k_retain_walk:' | tail -1 > $O/rwalk_clocks.txt
BMQ_TIMING=1 python bench.py --churn 100000 --steps 10 --warmup 2 --no-cpu-baseline --no-host-path > $O/churn100k.json 2> $O/churn100k.err
grep 'bmq index' $O/churn100k.err | tail -30 > $O/churn100k_phases.txt
BMQ_TIMING=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>&1 | grep 'rebuild:' > $O/rebuild_phases.txt
for th in 16 64 256; do  # the box grants 16 CPUs (cgroup quota) whatever nproc says: 16 threads = one per CPU
  python bench.py --no-cpu-baseline --no-host-path --steps 5 --warmup 2 --batcher-threads $th > $O/batcher$th.json 2> $O/batcher$th.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:
        print(os.path.basename(f), "unreadable", ex); continue
    print(os.path.basename(f), round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"], 3), "ms p50", round(d.get("p50_batch_ms", 0), 3), "p99",
          round(d.get("p99_batch_ms", 0), 3), {k: round(v, 3) for k, v in d["kernel_ms"].items()}, d.get("batching_front", ""), d["churn"]["apply_ms_mean"])
PY
cat $O/rebuild_phases.txt $O/wave_clocks.txt $O/rwalk_clocks.txt
