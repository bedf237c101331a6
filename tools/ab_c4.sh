out=gpurun_out/ab_c4.txt; : > $out
for v in "$@"; do
  for r in 1 2; do
  L=""; [ "$v" != default ] && L=$PWD/build/variants/libbmq_$v.so
  res=$(env ${L:+BMQ_LIB=$L} python bench.py --workload c4 --no-extras --no-churn --no-cpu-baseline --steps 8 --warmup 2 2>gpurun_out/ab_c4_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms'], 'step', round(d['ms_per_step'],4), round(d['value']/1e6,1))")
  echo "$v: $res" >> $out
  done
done
cat $out
