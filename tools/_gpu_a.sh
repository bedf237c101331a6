#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/a
export TMPDIR=/tmp
one() {
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d.get('ms_per_step_without_kernel_timing'),4), d['kernel_ms'])"
}
for r in 1 2; do
unset BMQ_LIB BMQ_QCAP BMQ_PCAP; one base
export BMQ_QCAP=128 BMQ_PCAP=128; one base_q128
export BMQ_LIB=/root/repo/bifromq_amd/variants/libbmq_mw5.so; one mw5_q128
done
