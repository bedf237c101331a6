"""Experiment (round 5): does the ORDER in which k_retain_walk takes the filters on matter?  The C4 batch as generated, and the same batch
with the filters a crude proxy calls heavy ('+' at level >= 2: a node range of hundreds of children) in front.  Prints kernel ms."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bifromq_amd as B
from bifromq_amd.workload import unpack

seed = 0xB1F20004
w = B.Workload(seed, 1, 1, 0)
data, off, tt = w.retain(seed, 1_000_000, filters=False)
eng = B.Engine(device=0, kernel_timing=True)
eng.retain_rebuild(w.tenants(), tt, packed_topics=(data, off))
fdata, foff, ft = w.retain(seed + 1, 100_000, filters=True)
filters = unpack(fdata, foff)

def weight(f):
    lv = f.split(b'/')
    for i, l in enumerate(lv):
        if l == b'+':
            return (3 if i >= 3 else 2 if i == 2 else 1 if i == 1 else 0)
    return 0
wts = np.array([weight(f) for f in filters])
orders = {"as generated": np.arange(len(filters)), "heavy first": np.argsort(-wts, kind="stable"), "heavy last": np.argsort(wts, kind="stable"),
          "random": np.random.RandomState(1).permutation(len(filters))}
tdata, toff = w.tenants_packed()
dev = torch.device("cuda:0")
d_tenants = torch.from_numpy(tdata.copy()).to(dev)
d_tenant_off = torch.from_numpy(toff.astype(np.int32)).to(dev)
for name, order in orders.items():
    fl = [filters[i] for i in order]
    fbytes = b"".join(fl)
    fo = np.zeros(len(fl) + 1, dtype=np.uint32)
    fo[1:] = np.cumsum([len(x) for x in fl])
    fb = np.frombuffer(fbytes + b"\0" * (32 - len(fbytes) % 16), dtype=np.uint8)
    d_f = torch.from_numpy(fb.copy()).to(dev)
    d_fo = torch.from_numpy(fo.astype(np.int32)).to(dev)
    d_ft = torch.zeros(len(fl), dtype=torch.int32, device=dev)
    n = len(fl)
    cap = 64 * n
    d_row = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    d_ids = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    ms = []
    for it in range(10):
        while True:
            eng.retain_match_batch_device(d_tenants.data_ptr(), d_tenant_off.data_ptr(), 1, d_ft.data_ptr(), d_f.data_ptr(), d_fo.data_ptr(), n,
                                          d_row.data_ptr(), d_ids.data_ptr(), cap, d_total.data_ptr())
            try:
                eng.finish()
                break
            except B.BmqError as ex:
                if ex.code != -3:
                    raise
                cap = int(d_total.item()) * 2
                d_ids = torch.zeros(cap, dtype=torch.int32, device=dev)
        st = eng.stats()
        ms.append((st.ms_walk, st.ms_expand))
    print("%-14s walk %.4f ms  expand %.4f ms   (weights: %s)" % (name, np.mean([m[0] for m in ms[3:]]), np.mean([m[1] for m in ms[3:]]), np.bincount(wts).tolist()))
