#!/bin/bash
# The round's LAST pass on a GPU box, sized for a small remaining budget (run through gpurun from the repo root):
#   bash tools/final_round.sh r05b
# the GPU suite on the final code (all of it, no -x: one failing test must not hide the others), the driver's command, and the two
# rocprofv3 passes of the ordered-batch leg (kernel trace; L2 counters of k_walk) that tools/ordered_collect.py turns into
# profiles/<round>/ordered_kernels.txt.  Most valuable first: whatever the budget cuts off is at the end.
R=${1:-r05b}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
T0=$SECONDS
timeout 480 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
echo "[final_round] pytest done at $((SECONDS - T0)) s"
timeout 360 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
tail -c 600 $O/bench_c3.json
echo "[final_round] bench done at $((SECONDS - T0)) s"
P="--ordered-only --steps 5 --warmup 2"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ord_kt -o ord -- python bench.py $P > $O/ord_kt.json 2> $O/ord_kt.err
echo "[final_round] kernel trace done at $((SECONDS - T0)) s"
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/ord_pmc -o ord -- python bench.py $P > $O/ord_pmc.json 2> $O/ord_pmc.err
echo "[final_round] counter pass done at $((SECONDS - T0)) s"
find $O -name "*.csv" | head -20
du -sh $O
