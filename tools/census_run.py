#!/usr/bin/env python3
"""C3 index + one 1M-publish grouped batch through the engine alone (no torch, no exchange, no bench loop): BMQ_DEBUG=8 BMQ_CENSUS_FILE=... to
look at the residency of k_walk's waves outside bench.py."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bifromq_amd as B
w = B.Workload(0xB1F20003, 1000, 10_000, 1)
eng = B.Engine(device=0, kernel_timing=True)
eng.rebuild(packed=w.keys_packed())
data, off, tt = w.topics(0xB1F20003 + 1000, 1_000_000, grouped=True)
tn = w.tenants()
for i in range(4):
    row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
    st = eng.stats()
    print("batch", i, "ids", len(ids), "walk ms", st.ms_walk, "expand ms", st.ms_expand)
eng.close()
