cd $GRAFT_REPO_ROOT
run() { BMQ_LIB=$PWD/bifromq_amd/$1 BMQ_QCAP=$2 BMQ_PCAP=$3 timeout 300 python bench.py --no-cpu-baseline --no-host-path --steps 20 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 q$2 p$3', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})"; }
run libbmq.so 192 160
run libbmq_w5.so 192 160
run libbmq_w6.so 192 160
run libbmq_w5.so 128 128
run libbmq_w6.so 128 128
run libbmq.so 128 128
