#!/usr/bin/env python3
"""Offline look at a k_walk residency census (BMQ_DEBUG=8 BMQ_CENSUS_FILE=<file>): uint4 per wave {start lo, start hi, duration, HW_ID | XCC_ID << 16}."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 4)
st = a[:, 0].astype(np.uint64) | (a[:, 1].astype(np.uint64) << np.uint64(32))
ok = st > 0
print("waves", len(a), "with a start stamp", int(ok.sum()))
a, st = a[ok], st[ok]
dur = a[:, 2].astype(np.int64)
st = (st - st.min()).astype(np.int64)
en = st + dur
hw = a[:, 3]
wave_id, simd, pipe, cu, sh, se, xcc = hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, hw >> 16
print("span", int(en.max()), "ticks; mean duration", dur.mean(), "; mean resident", dur.sum() / en.max(), "=", dur.sum() / en.max() / 1024, "per SIMD")
print("wave_id histogram", np.bincount(wave_id))
print("xcc", np.bincount(xcc), "se", np.bincount(se), "sh", np.bincount(sh), "cu", np.bincount(cu), "simd", np.bincount(simd), "pipe", np.bincount(pipe))
# resident waves over time, chip-wide
T = int(en.max())
grid = np.linspace(0, T, 41)[:-1]
res = [(int(((st <= t) & (en > t)).sum())) for t in grid]
print("resident waves at 40 instants:", res)
key = (xcc.astype(np.int64) << 16) | (se << 13) | (sh << 12) | (cu << 8) | (simd << 4)
cus = (xcc.astype(np.int64) << 16) | (se << 13) | (sh << 12) | (cu << 8)
print("distinct SIMDs", len(np.unique(key)), "distinct CUs", len(np.unique(cus)))
one = key == key[0]
order = np.argsort(st[one])
print("one SIMD's waves (start, end, slot):", [(int(s), int(e), int(w)) for s, e, w in zip(st[one][order][:24], en[one][order][:24], wave_id[one][order][:24])])
