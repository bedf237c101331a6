import time, numpy as np, ctypes as C
import bifromq_amd as B
from bifromq_amd.engine import pinned, _ptr
w = B.Workload(7, 100, 2000, 1)
eng = B.Engine(device=0).rebuild(w.keys())
tn = w.tenants(); tdata, toff = w.tenants_packed()
L = B._lib.lib()
for n in (1, 64, 256):
    data, off, tt = w.topics(5, n)
    for pin in (False, True):
        if pin:
            mk = lambda a: (lambda p: (p.__setitem__(slice(None), a), p)[1])(pinned(len(a), a.dtype))
        else:
            mk = lambda a: a.copy()
        pt_, pto, pd = mk(np.concatenate([tdata, np.zeros(16, np.uint8)])), mk(toff.astype(np.uint32)), mk(np.concatenate([data, np.zeros(16, np.uint8)]))
        po, ptt = mk(off.astype(np.uint32)), mk(tt.astype(np.uint32))
        row = pinned(n + 1, np.uint32) if pin else np.zeros(n + 1, np.uint32)
        ids = pinned(64 * n + 1024, np.uint32) if pin else np.zeros(64 * n + 1024, np.uint32)
        need = C.c_uint64(); k = C.c_int()
        def blocking():
            assert L.bmq_match_batch(eng.h, _ptr(pt_), _ptr(pto), len(tn), _ptr(ptt), _ptr(pd), _ptr(po), n, _ptr(row), _ptr(ids), len(ids), C.byref(need)) == 0
        def subwait():
            assert L.bmq_match_submit(eng.h, _ptr(pt_), _ptr(pto), len(tn), _ptr(ptt), _ptr(pd), _ptr(po), n, C.byref(k)) == 0
            assert L.bmq_match_wait(eng.h, k.value, _ptr(row), _ptr(ids), len(ids), C.byref(need)) == 0
        def sub2():
            k2 = C.c_int()
            assert L.bmq_match_submit(eng.h, _ptr(pt_), _ptr(pto), len(tn), _ptr(ptt), _ptr(pd), _ptr(po), n, C.byref(k)) == 0
            assert L.bmq_match_submit(eng.h, _ptr(pt_), _ptr(pto), len(tn), _ptr(ptt), _ptr(pd), _ptr(po), n, C.byref(k2)) == 0
            assert L.bmq_match_wait(eng.h, k.value, _ptr(row), _ptr(ids), len(ids), C.byref(need)) == 0
            assert L.bmq_match_wait(eng.h, k2.value, _ptr(row), _ptr(ids), len(ids), C.byref(need)) == 0
        tot = pinned(2, np.uint64)
        def subdev():
            assert L.bmq_match_submit_dev(eng.h, _ptr(pt_), _ptr(pto), len(tn), _ptr(ptt), _ptr(pd), _ptr(po), n, _ptr(row), _ptr(ids), len(ids), _ptr(tot), C.byref(k)) == 0
            assert L.bmq_match_wait_dev(eng.h, k.value, C.byref(need)) == 0
        def subdev2():
            k2 = C.c_int()
            assert L.bmq_match_submit_dev(eng.h, _ptr(pt_), _ptr(pto), len(tn), _ptr(ptt), _ptr(pd), _ptr(po), n, _ptr(row), _ptr(ids), len(ids), _ptr(tot), C.byref(k)) == 0
            assert L.bmq_match_submit_dev(eng.h, _ptr(pt_), _ptr(pto), len(tn), _ptr(ptt), _ptr(pd), _ptr(po), n, _ptr(row), _ptr(ids), len(ids), _ptr(tot), C.byref(k2)) == 0
            assert L.bmq_match_wait_dev(eng.h, k.value, C.byref(need)) == 0
            assert L.bmq_match_wait_dev(eng.h, k2.value, C.byref(need)) == 0
        if pin:
            blocking(); ref = (row.copy(), ids[:need.value].copy()); row[:] = 0; subdev()
            assert (row == ref[0]).all() and (ids[:need.value] == ref[1]).all()
        for name, f, per in (("blocking", blocking, 1),) + ((("submit_dev+wait_dev", subdev, 1), ("2x submit_dev, 2x wait", subdev2, 2)) if pin else ()) + ( ("submit+wait", subwait, 1), ("2x submit, 2x wait", sub2, 2)):
            if name == "blocking" and pin: continue
            for _ in range(20): f()
            t0 = time.perf_counter()
            for _ in range(300): f()
            print("n=%d pinned=%d %-20s %.1f us per batch" % (n, pin, name, (time.perf_counter() - t0) / 300 / per * 1e6))
