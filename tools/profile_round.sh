#!/bin/bash
# Regenerates the evidence kept under profiles/<round>/ on a GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02
# Writes under gpurun_out/<round>/ (scratch); tools/collect_profiles.py then copies the summaries into profiles/<round>/.
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.json
python bench.py --workload c2 --cpu-sample-topics 20000 > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --workload c4 > $O/bench_c4.json 2> $O/bench_c4.err
for w in c3 c2 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o $w -- python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_$w.log 2>&1
done
# HBM traffic: separate counter passes, nothing else enabled (MI355X_MICROARCH.md, HBM / rocprofv3 section)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o c3 -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
find $O -name "*.csv" | head -40
