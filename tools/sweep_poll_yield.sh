#!/bin/bash
# Experiment (GPU box): blocking single-topic calls/s through the batching front for settings of the leader's yield (BMQ_POLL_YIELD), the generations in
# flight (BMQ_BATCHER_INFLIGHT) and the wake-up tree (BMQ_BATCHER_FANOUT); BMQ_LIB = a -DBMQ_EXPERIMENTS=1 build.  usage: tools/sweep_poll_yield.sh "y..." "infl..." "fan..." reps
export BMQ_LIB=${BMQ_LIB:-$PWD/build/variants/libbmq_head_x.so}
for r in $(seq 1 ${4:-4}); do
  for y in ${1:-0 64}; do for i in ${2:-6}; do for f in ${3:-8}; do
    echo "cfg y$y i$i f$f $(BMQ_POLL_YIELD=$y BMQ_BATCHER_INFLIGHT=$i BMQ_BATCHER_FANOUT=$f PYTHONPATH=. python tools/batcher_sweep.py child 64 | tail -1)"
  done; done; done
done
