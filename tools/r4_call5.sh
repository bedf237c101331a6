#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_zz_churn_gpu.py tests/test_batcher_gpu.py tests/test_formats_gpu.py -x -q -k "not full_size" > $O/pytest_dist.log 2>&1; tail -3 $O/pytest_dist.log | head -2
out=$O/ab_geom.txt; : > $out
run() { name=$1; shift
  r=$(env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-extras --batcher-threads 0 2>$O/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms'], 'step', round(d['ms_per_step'],4))")
  echo "$name: $r" >> $out; grep "k_walk" $O/ab_$name.err | tail -2 >> $out; }
run g0 BMQ_WALK_GEOM=0
run g0_mixed BMQ_WALK_GEOM=0 BMQ_WALK_MIXED=1
run g2 BMQ_WALK_GEOM=2
run g0_dbg BMQ_WALK_GEOM=0 BMQ_DEBUG=2
BMQ_DEBUG=8 BMQ_CENSUS_FILE=$O/census_g0.bin python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-extras --batcher-threads 0 > /dev/null 2> $O/census_g0.err; python tools/census.py $O/census_g0.bin | grep "wave_id" >> $out
cat $out
