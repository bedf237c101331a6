mkdir -p gpurun_out/r05e; O=gpurun_out/r05e; export TMPDIR=/tmp
timeout 58 python -m pytest tests -m gpu -q -x -k "not full_size and not table_growth and not two_ranks" > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
B="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-host-path --batcher-topics 100000"
timeout 14 python bench.py $B --batcher-threads 64 > $O/bf_mode1.json 2>/dev/null
BMQ_PUBLISH_KERNEL=0 timeout 14 python bench.py $B --batcher-threads 64 > $O/bf_mode0.json 2>/dev/null
BMQ_PUBLISH_KERNEL=2 timeout 12 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-host-path --batcher-threads 0 > $O/c3_mode2.json 2>/dev/null
python - <<'PY'
import json
for f in ("bf_mode1","bf_mode0","c3_mode2"):
    try:
        d=json.loads(open("gpurun_out/r05e/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("batching_front",{}).get("calls_per_s"), d.get("batching_front",{}).get("mean_topics_per_launch"))
    except Exception as ex: print(f, "failed", ex)
PY
