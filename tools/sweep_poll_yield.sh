export BMQ_LIB=$PWD/build/variants/libbmq_head_x.so
for r in 1 2 3 4; do
  for y in 0 1 4 16 64 256; do
    echo "yield $y $(BMQ_POLL_YIELD=$y PYTHONPATH=. python tools/batcher_sweep.py child 64 | tail -1)"
    echo "yield16t $y $(BMQ_POLL_YIELD=$y PYTHONPATH=. python tools/batcher_sweep.py child 16 | tail -1)"
  done
done
