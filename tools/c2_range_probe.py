"""Experiment (GPU box): the lengths of the matched id ranges k_expand turns into ids on C2 (1 tenant x 1 M routes) and C3 -- which of its two paths
(short ranges flattened into one element space / ranges of EXP_LONG = 64 ids and more streamed one after the other) carries the bytes.
    python tools/c2_range_probe.py > gpurun_out/c2_ranges.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bifromq_amd as B  # noqa: E402

for name, (nt, per, seed, n) in (("C2", (1, 1_000_000, 0xB1F20002, 20000)), ("C3 (100 tenants of it)", (100, 10_000, 0xB1F20003, 100000))):
    w = B.Workload(seed, nt, per, 1)
    eng = B.Engine(device=0).rebuild(w.keys())
    data, off, tt = w.topics(11, n)
    tdata, toff = w.tenants_packed()
    cap = 64 * n
    while True:
        rptr = np.zeros(n + 1, dtype=np.uint32)
        ranges = np.zeros((cap, 2), dtype=np.uint32)
        side = np.zeros(cap, dtype=np.uint32)
        row = np.zeros(n + 1, dtype=np.uint32)
        try:
            info = eng.match_wait_ranges(eng.match_submit_fmt(tdata, toff.astype(np.uint32), w.n_tenants, tt.astype(np.uint32), data, off.astype(np.uint32), n, eng.FMT_RANGES), rptr, ranges, side, row)
            break
        except B.BmqError as ex:
            if ex.code != -3:
                raise
            cap = int(max(ex.info.n_ranges, ex.info.n_side_ids)) + 16
    nr = int(info.n_ranges)
    cnt = (ranges[:nr, 1] & 0x7FFFFFFF).astype(np.int64)
    ids = int(cnt.sum())
    print("%s: %d rows, %d ranges (%.1f per row), %d ids (%.1f per row, %.1f per range); side (indirect) ids %d" % (name, n, nr, nr / n, ids, ids / n, ids / max(nr, 1), int(info.n_side_ids)))
    edges = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096, 1 << 40]
    lo = 1
    for hi in edges[1:]:
        m = (cnt >= lo) & (cnt < hi)
        print("   ranges of %5d .. %-6s ids: %5.1f %% of the ranges, %5.1f %% of the ids" % (lo, hi - 1 if hi < (1 << 40) else "", 100.0 * m.sum() / max(nr, 1), 100.0 * cnt[m].sum() / max(ids, 1)))
        lo = hi
    eng.close()
