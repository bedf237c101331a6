cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ex in fanout ids; do timeout 600 python bench.py --no-cpu-baseline --no-host-path --steps 10 --exchange-selftest --exchange-impl lib --exchange $ex --node-batch-steps 0 > gpurun_out/b_lib_$ex.log 2>&1; tail -1 gpurun_out/b_lib_$ex.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$ex', round(d['value']/1e6,1), d['ms_per_step'], d['config']['exchange_impl'])" || tail -8 gpurun_out/b_lib_$ex.log; done
