for v in default "HSA_ENABLE_SDMA=1" "HSA_ENABLE_SDMA=0" "HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_SDMA=1 ROC_AQL_QUEUE_SIZE=16384"; do
  echo "== $v"
  if [ "$v" = default ]; then E=""; else E="$v"; fi
  env $E timeout 100 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --batcher-threads 0 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());h=d['host_visible'];print(round(h['value_host_visible']/1e6), {k:(round(v['value_host_visible']/1e6), round(v['p50_host_visible_ms'],3)) for k,v in h['formats'].items()}, round(h['p50_host_visible_ms'],3), round(d['value']/1e6))"
done
