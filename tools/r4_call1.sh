#!/bin/bash
# round 4, GPU call 1: SQ / TCC / TCP counters of the C3 kernels + occupancy A/B of k_walk
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
python tools/pmc_sq.py r04 c3 > $O/pmc_sq.log 2>&1; tail -3 $O/pmc_sq.log
out=$O/ab_occ.txt; : > $out
run() { name=$1; shift
  r=$(env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path --no-extras --batcher-threads 0 2>$O/ab_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms'], 'step', round(d['ms_per_step'],4))")
  echo "$name: $r" >> $out; }
V=$PWD/build/variants
run default X=1
run default_q128 BMQ_QCAP=128 BMQ_PCAP=128
run mw4fl8 BMQ_LIB=$V/libbmq_mw4fl8.so BMQ_QCAP=128 BMQ_PCAP=128
run mw5 BMQ_LIB=$V/libbmq_mw5.so BMQ_QCAP=128 BMQ_PCAP=128
run mw6 BMQ_LIB=$V/libbmq_mw6.so BMQ_QCAP=128 BMQ_PCAP=128
run mw8 BMQ_LIB=$V/libbmq_mw8.so BMQ_QCAP=128 BMQ_PCAP=128
cat $out
