cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --no-host-path --steps 10 --exchange-selftest > gpurun_out/b_ex1.log 2>&1; tail -1 gpurun_out/b_ex1.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['exchange'][:60]); print(d.get('node_batch'))" || tail -20 gpurun_out/b_ex1.log
