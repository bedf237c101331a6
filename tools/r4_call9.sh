#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_zz_churn_gpu.py tests/test_formats_gpu.py -x -q -k "not full_size" > $O/pytest_dist.log 2>&1; tail -3 $O/pytest_dist.log | head -2
bash tools/ab_quick.sh dedup "X=1" nodedup "BMQ_DEDUP_MIN=4294967295" dedup_dbg "BMQ_DEBUG=2"
