#!/bin/bash
# round 4, after the k_expand rewrite: the evidence of profiles/r04 again, most important first, every step only while time is left
#   BUDGET=400 bash tools/r4_final.sh         (then, at home: python tools/collect_profiles.py r04)
export TMPDIR=/tmp
R=r04; O=gpurun_out/$R; mkdir -p $O
T0=$(date +%s)
left() { echo $(( ${BUDGET:-400} - ($(date +%s) - T0) )); }
step() { need=$1; shift; if [ $(left) -lt $need ]; then echo "skipped (time left $(left) s < $need): $*" | tee -a $O/skipped.txt; return; fi; "$@"; }
P="--no-cpu-baseline --no-extras --batcher-threads 0"
pmc() { for c in FETCH_SIZE WRITE_SIZE; do timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$1_$c -o $1 -- python bench.py --workload $1 --steps 4 --warmup 1 $P > $O/pmc_$1_$c.log 2>&1; done; }
kt() { timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1 -o $1 -- python bench.py --workload $1 --steps 20 --warmup 5 $P --no-host-path > $O/kt_$1.log 2>&1; }
b2() { timeout 150 python bench.py --workload c2 --cpu-sample-topics 20000 --no-extras > $O/bench_c2.json 2> $O/bench_c2.err; }
b4() { timeout 150 python bench.py --workload c4 --no-extras > $O/bench_c4.json 2> $O/bench_c4.err; }
sq() { timeout 120 python tools/pmc_sq.py r04x c3 --groups=0,1 > $O/pmc_sq_expand.log 2>&1; cp gpurun_out/r04x/c3_pmc_sq.csv $O/c3_pmc_sq_new_k_expand.csv 2>/dev/null; }
clk() { BMQ_DEBUG=4 BMQ_LIB=build/variants/libbmq_eclk.so timeout 60 python bench.py --steps 3 --warmup 1 $P --no-host-path 2>&1 | grep "k_expand waves" | tail -1 > $O/k_expand_clocks.txt; cat $O/k_expand_clocks.txt; }
timeout 420 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
echo "after pytest: $(left) s left"
step 60 bash -c "timeout 200 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 200 $O/bench_c3.json"
echo "after bench: $(left) s left"
step 35 pmc c3
step 25 kt c3
step 30 b2
step 30 b4
step 35 pmc c2
step 35 pmc c4
step 20 kt c2
step 20 kt c4
step 20 clk
step 40 sq
echo "done: $(left) s left"
