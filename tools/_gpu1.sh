cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/kt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt/c3 -o c3 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/kt/c3.log 2>&1
f=$(find gpurun_out/kt/c3 -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt/s -o s -- python bench.py --steps 50 --warmup 5 --topics 10000 --no-cpu-baseline > gpurun_out/kt/s.log 2>&1
f=$(find gpurun_out/kt/s -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200
