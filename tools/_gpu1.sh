cd $GRAFT_REPO_ROOT
for ev in 0 1 0; do for n in 1000000 10000; do BMQ_KERNEL_EVENTS=$ev timeout 300 python bench.py --no-cpu-baseline --no-host-path --steps 40 --topics $n 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('events $ev', $n, round(d['value']/1e6,1), round(d['ms_per_step'],4), round(d['kernel_ms']['all_kernels'],4))"; done; done
