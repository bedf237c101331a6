#!/bin/bash
# Regenerates the evidence kept under profiles/<round>/ on a GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r03
# Writes under gpurun_out/<round>/ (scratch); tools/collect_profiles.py then copies the summaries into profiles/<round>/.
R=${1:-r04}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$SKIP_PYTEST" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
fi
# the driver's command (N = 1): default workload C3 + the compact C2 / C4 / C5 / batching-front legs under "extra"
timeout 300 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.json
timeout 300 python bench.py --workload c2 --cpu-sample-topics 20000 --no-extras > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --workload c4 --no-extras > $O/bench_c4.json 2> $O/bench_c4.err
P="--no-cpu-baseline --no-extras --batcher-threads 0"
for w in c3 c2 c4; do
  # (--no-host-path: only the timed loop launches the match kernels, so that the trace's AVERAGE is the average of the same launches the
  # bench line times -- the host-visible legs run the kernels next to PCIe copies and other result formats)
  # (--no-churn: C4's add / remove leg launches the same kernels on another index)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o $w -- python bench.py --workload $w --steps 20 --warmup 5 $P --no-host-path --no-churn > $O/kt_$w.log 2>&1
done
# HBM traffic: separate counter passes, nothing else enabled (MI355X_MICROARCH.md, HBM / rocprofv3 section)
for w in c3 c2 c4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -o $w -- python bench.py --workload $w --steps 4 --warmup 1 $P --no-host-path --no-churn > $O/pmc_${w}_$c.log 2>&1
  done
done
# what WRITE_SIZE reports for stores of known size (tools/ubench_stores.hip) -> $O/write_calibration.json, applied by tools/collect_profiles.py
timeout 300 python tools/calibrate_writes.py $R > $O/calibrate_writes.log 2>&1; tail -3 $O/calibrate_writes.log
# SQ / TCC / TCP counters of the C3 and C4 kernels (one pass per counter group; tools/pmc_sq.py writes $O/<workload>_pmc_sq.csv)
timeout 900 python tools/pmc_sq.py $R c3 > $O/pmc_sq.log 2>&1; tail -2 $O/pmc_sq.log
timeout 600 python tools/pmc_sq.py $R c4 --groups=0,1,2,4 --no-churn > $O/pmc_sq_c4.log 2>&1; tail -2 $O/pmc_sq_c4.log
find $O -name "*.csv" | head -60
