export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for v in default 3 2 1; do
  if [ $v = default ]; then e="X=1"; else e="BMQ_TPW_SHIFT=$v"; fi
  r=$(env $e python bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline --no-host-path --no-extras 2>$O/c4_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,1),'M filters/s', d['kernel_ms'], d['roofline']['frac'], (d.get('churn') or {}).get('match_ms_after'), (d.get('churn') or {}).get('kernel_ms_after'))")
  echo "tpw_shift $v: $r"
done
