#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_zz_churn_gpu.py tests/test_retain_gpu.py -x -q -m gpu > gpurun_out/pt.log 2>&1
grep -E "passed|failed" gpurun_out/pt.log | tail -2
for w in c3 c2 c4; do
python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(d['value']/1e6,1), round(d['ms_per_step'],4), d.get('ms_per_step_without_kernel_timing'), d['kernel_ms'])"
BMQ_DEBUG=4 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep k_expand | tail -1
done
