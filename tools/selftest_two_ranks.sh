#!/bin/bash
# The N > 1 code path of bench.py on a 1-GPU box: two ranks, both on GPU 0, collectives over gloo (BMQ_BENCH_ONE_GPU=1).  Numbers are
# meaningless; what is checked is that sharding, the step loop with the exchange, the node-wide batch (device partition, hot-tenant split,
# fan-out all-reduce) and the JSON line work with world size 2.
export BMQ_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout ${1:-240} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 \
  --topics 200000 --no-cpu-baseline 2> gpurun_out/two_ranks.err | tail -1 > gpurun_out/two_ranks.json
echo "rc=$?"; tail -c 1500 gpurun_out/two_ranks.json; echo; tail -5 gpurun_out/two_ranks.err | cut -c1-300
