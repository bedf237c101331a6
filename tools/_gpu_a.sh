#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/a
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_batcher_gpu.py -m gpu -q > gpurun_out/a/pytest_gpu3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/a/pytest_gpu3.log
tail -3 gpurun_out/a/pytest_gpu3.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --batcher-threads 256 > gpurun_out/a/batcher256.json 2> gpurun_out/a/batcher256.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/a/batcher256.json").read().strip().splitlines()[-1])
    print(json.dumps(d.get("batching_front"), indent=1))
except Exception as ex:
    print("no bench line", ex)
PY
tail -3 gpurun_out/a/batcher256.err
g++ -O2 -std=c++17 -pthread -o /tmp/cache_perf tools/cache_fuzz.cpp bifromq_amd/csrc/bmq_codec.cpp && for t in 16 64 256; do /tmp/cache_perf perf $t 400000; done
