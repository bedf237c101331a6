#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_dist_gpu.py -x -q -m gpu > gpurun_out/pt.log 2>&1
grep -E "passed|failed" gpurun_out/pt.log | tail -2
one() {
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 c3', round(d['ms_per_step'],4), round(d.get('ms_per_step_without_kernel_timing'),4), d['kernel_ms'])"
python bench.py --topics 10000 --steps 100 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 10k', round(d['ms_per_step'],4), round(d.get('ms_per_step_without_kernel_timing'),4), d['kernel_ms'])"
}
for r in 1 2; do
unset BMQ_LIB; one new
export BMQ_LIB=/root/repo/bifromq_amd/variants/libbmq_prev.so; one prev
done
unset BMQ_LIB
BMQ_DEBUG=2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>&1 >/dev/null | grep 'k_walk waves' | tail -1
