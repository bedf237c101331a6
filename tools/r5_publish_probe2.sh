# the headline batch with the counters through the copy engine (0) and through k_publish (2), same box, 20 steps each, twice
mkdir -p gpurun_out/r05f; B="--steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-host-path --batcher-threads 0"
for m in 0 2 0 2; do BMQ_PUBLISH_KERNEL=$m timeout 12 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode $m', d['value'], d['ms_per_step'], d['p50_batch_ms'], d['kernel_ms']['all_kernels'])" | tee -a gpurun_out/r05f/headline_ab.txt; done
