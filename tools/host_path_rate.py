"""PCIe-inclusive rate of the host-buffer entry point bmq_match_batch on the C3 workload."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bifromq_amd as B
from bifromq_amd import _lib
from bifromq_amd.engine import _ptr

w = B.Workload(0xB1F20003, 1000, 10000, 1)
eng = B.Engine(device=0)
kb, ko = w.keys_packed()
eng.rebuild_raw(kb.ctypes.data, ko.ctypes.data, w.n_keys)
tdata, toff = w.tenants_packed()
n = 1_000_000
data, off, tt = w.topics(0xB1F20003 + 1000, n, grouped=True)
row = np.zeros(n + 1, dtype=np.uint32)
ids = np.zeros(32 * n, dtype=np.uint32)
need = C.c_uint64()
L = _lib.lib()
times = []
for i in range(8):
    t0 = time.perf_counter()
    rc = L.bmq_match_batch(eng.h, _ptr(tdata), _ptr(toff), w.n_tenants, _ptr(tt), _ptr(data), _ptr(off), n, _ptr(row), _ptr(ids),
                           len(ids), C.byref(need))
    times.append(time.perf_counter() - t0)
    assert rc == 0, rc
best = min(times[2:])
print("bmq_match_batch host path: %.2f ms per 1M-topic batch (best of 6) -> %.1f M topics/s; in %.1f MB, out %.1f MB"
      % (best * 1e3, n / best / 1e6, (off[-1] + 8 * n) / 1e6, (4 * (n + 1) + 4 * need.value) / 1e6))
