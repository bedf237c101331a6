"""Experiment (GPU box): does the ORDER of the filters in a C4 batch matter to k_retain_walk?  The kernel hands the batch out in quads of consecutive
filters, dynamically; its tail was read as "single heavy filters walked alone at the end" (DESIGN 7.4).  This runs bench.py's C4 batch 0 as generated,
sorted heaviest first (by the number of topics a filter matched in a first run: the best cost estimate there is), and heaviest last.
    python tools/c4_order_probe.py > gpurun_out/c4_order.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bifromq_amd as B  # noqa: E402

dev = torch.device("cuda", 0)
seed = 0xB1F20004
w = B.Workload(seed, 1, 1, 0)
n_topics, n = 1_000_000, 100_000
data, off, tt = w.retain(seed, n_topics, filters=False)
eng = B.Engine(device=0, kernel_timing=True)
eng.retain_rebuild(w.tenants(), tt, packed_topics=(data, off))
tdata, toff = w.tenants_packed()
d_tenants = torch.from_numpy(tdata.copy()).to(dev)
d_tenant_off = torch.from_numpy(toff.astype(np.int32)).to(dev)
fdata, foff, ft = w.retain(seed + 1, n, filters=True)
foff = foff.astype(np.int64)
cap = 64 * n
d_row = torch.zeros(n + 1, dtype=torch.int32, device=dev)
d_ids = torch.zeros(cap, dtype=torch.int32, device=dev)
d_total = torch.zeros(1, dtype=torch.int64, device=dev)


def upload(order):
    lens = (foff[1:] - foff[:-1])[order]
    noff = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=noff[1:])
    out = np.zeros(int(noff[-1]) + 32, dtype=np.uint8)
    # gather the filters' bytes in the new order
    src = np.repeat(foff[:-1][order], lens) + (np.arange(int(noff[-1])) - np.repeat(noff[:-1], lens))
    out[:int(noff[-1])] = fdata[src]
    return (torch.from_numpy(out).to(dev), torch.from_numpy(noff.astype(np.int32)).to(dev), torch.from_numpy(ft[order].astype(np.int32)).to(dev))


def run(bt, k=8):
    global d_ids, cap
    ms = []
    for _ in range(k):
        while True:
            eng.retain_match_batch_device(d_tenants.data_ptr(), d_tenant_off.data_ptr(), 1, bt[2].data_ptr(), bt[0].data_ptr(), bt[1].data_ptr(), n,
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
    a = np.array(ms[2:])
    return a.mean(axis=0), a.min(axis=0)


ident = np.arange(n)
base = upload(ident)
m, mn = run(base)
rows = d_row.cpu().numpy().astype(np.int64)
cnt = rows[1:] - rows[:-1]
with open("gpurun_out/c4_block_ids.txt", "w") as fh:  # ids per 64-row block, for tools/ubench_write_skew.hip
    pad = np.concatenate([cnt, np.zeros((-n) % 64, dtype=cnt.dtype)])
    fh.write("\n".join(str(int(x)) for x in pad.reshape(-1, 64).sum(axis=1)) + "\n")
levels = np.array([bytes(fdata[foff[i]:foff[i + 1]]).count(b"/") + 1 for i in range(n)])
plus = np.array([bytes(fdata[foff[i]:foff[i + 1]]).split(b"/").count(b"+") for i in range(n)])
print("as generated:        walk %.4f ms (min %.4f)  expand %.4f" % (m[0], mn[0], m[1]))
print("matches per filter: mean %.0f p50 %.0f p99 %.0f max %d; levels mean %.2f; '+' levels mean %.2f max %d" % (cnt.mean(), np.median(cnt), np.percentile(cnt, 99), cnt.max(), levels.mean(), plus.mean(), plus.max()))
for name, order in (("heaviest first (by matches)", np.argsort(-cnt, kind="stable")), ("heaviest last", np.argsort(cnt, kind="stable")),
                    ("most '+' levels first", np.argsort(-plus, kind="stable")), ("shuffled", np.random.default_rng(1).permutation(n))):
    m, mn = run(upload(order))
    print("%-28s walk %.4f ms (min %.4f)  expand %.4f" % (name + ":", m[0], mn[0], m[1]))
# the floor a single wave sets: the heaviest filters alone (a batch of k filters: the kernel's time is its slowest wave's)
for k in (1, 8, 64, 1024):
    order = np.argsort(-cnt, kind="stable")[:k]
    lens = (foff[1:] - foff[:-1])[order]
    noff = np.zeros(k + 1, dtype=np.int64)
    np.cumsum(lens, out=noff[1:])
    out = np.zeros(int(noff[-1]) + 32, dtype=np.uint8)
    src = np.repeat(foff[:-1][order], lens) + (np.arange(int(noff[-1])) - np.repeat(noff[:-1], lens))
    out[:int(noff[-1])] = fdata[src]
    bt = (torch.from_numpy(out).to(dev), torch.from_numpy(noff.astype(np.int32)).to(dev), torch.from_numpy(ft[order].astype(np.int32)).to(dev))
    n_save = n
    n = k
    m, mn = run(bt)
    n = n_save
    print("the %4d heaviest filters alone (%.0f matches each on average): walk %.4f ms (min %.4f)  expand %.4f" % (k, cnt[order].mean(), m[0], mn[0], m[1]))
# which SHAPE of filter carries the walk's time: batches of one class each (the level of the '+', '#' at the end or not).  Two batch sizes: 100 k filters drawn from the
# class (throughput) and 2048 (one quad per wave at most: the time is the class's slowest filters')
split = [bytes(fdata[foff[i]:foff[i + 1]]).split(b"/") for i in range(n)]
ppos = np.array([lv.index(b"+") if b"+" in lv else -1 for lv in split])
hashed = np.array([lv[-1] == b"#" for lv in split])
rng = np.random.default_rng(7)


def run_subset(order):
    global n
    k = len(order)
    lens = (foff[1:] - foff[:-1])[order]
    noff = np.zeros(k + 1, dtype=np.int64)
    np.cumsum(lens, out=noff[1:])
    out = np.zeros(int(noff[-1]) + 32, dtype=np.uint8)
    src = np.repeat(foff[:-1][order], lens) + (np.arange(int(noff[-1])) - np.repeat(noff[:-1], lens))
    out[:int(noff[-1])] = fdata[src]
    bt = (torch.from_numpy(out).to(dev), torch.from_numpy(noff.astype(np.int32)).to(dev), torch.from_numpy(ft[order].astype(np.int32)).to(dev))
    n_save, n = n, k
    try:
        return run(bt, k=5)
    except B.BmqError as ex:
        if ex.code != -6:
            raise
        return (np.array([float("nan"), float("nan")]), None)
    finally:
        n = n_save


print("class: '+' at level, '#' at the end | share of the batch | mean matches | walk ms: 100 k filters of the class / 2048 of them / the 2048 with the most matches | ")
for pp in range(-1, 8):
    for hh in (False, True):
        idx = np.nonzero((ppos == pp) & (hashed == hh))[0]
        if len(idx) < 200:
            continue
        big = run_subset(rng.choice(idx, 100_000 if not hh or pp < 0 else 20_000, replace=True))[0]
        small = run_subset(rng.choice(idx, 2048, replace=False if len(idx) >= 2048 else True))[0]
        top = run_subset(idx[np.argsort(-cnt[idx], kind="stable")[:2048]])[0]
        print("  '+' at %2d  '#' %-5s  %5.1f %%  matches %9.0f   walk %.4f (%s filters) / %.4f / %.4f ms" % (pp, hh, 100.0 * len(idx) / n, cnt[idx].mean(), big[0], "100 k" if not hh or pp < 0 else "20 k", small[0], top[0]))
# longest-job-first with the cost a pre-pass could know: the size of the frontier behind the filter's '+' (distinct children of its literal prefix among the
# retained topics) x the levels that follow it -- computed here on the host from the topics
t0 = time.time()
kids = {}
tb = bytes(data)
for i in range(n_topics):
    lv = tb[off[i]:off[i + 1]].split(b"/")
    for k in range(len(lv)):
        kids.setdefault(b"/".join(lv[:k]), set()).add(lv[k])
cost = np.zeros(n)
for i, lv in enumerate(split):
    if b"+" in lv:
        pp = lv.index(b"+")
        tail = len(lv) - 1 - pp - (1 if lv[-1] == b"#" else 0)
        cost[i] = len(kids.get(b"/".join(lv[:pp]), ())) * max(tail, 0)
print("cost estimate (frontier behind the '+' x literal levels behind it): mean %.1f p99 %.0f max %.0f; %d filters above 256, %d above 1024  (%.0f s on the host)" % (cost.mean(), np.percentile(cost, 99), cost.max(), int((cost > 256).sum()), int((cost > 1024).sum()), time.time() - t0))
for name, order in (("costliest first (all sorted)", np.argsort(-cost, kind="stable")),
                    ("the filters above 256 first, the rest as generated", np.concatenate([np.nonzero(cost > 256)[0][np.argsort(-cost[cost > 256], kind="stable")], np.nonzero(cost <= 256)[0]])),
                    ("costliest last", np.argsort(cost, kind="stable"))):
    m, mn = run(upload(order))
    print("%-52s walk %.4f ms (min %.4f)  expand %.4f" % (name + ":", m[0], mn[0], m[1]))
# is it CLUMPING -- waves that happen to hold two or three costly filters at once?  The costly filters (estimate above 1024) dealt out evenly instead of where chance puts them
hv = np.nonzero(cost > 1024)[0]
hv = hv[np.argsort(-cost[hv], kind="stable")]
lt = np.nonzero(cost <= 1024)[0]
def interleave(step, lead):  # one costly filter at the head of every `step`-th quad from quad `lead` on, light filters everywhere else
    out, li, hi = [], 0, 0
    q = 0
    while li < len(lt) or hi < len(hv):
        quad = []
        if hi < len(hv) and q >= lead and (q - lead) % step == 0:
            quad.append(hv[hi]); hi += 1
        while len(quad) < 4 and li < len(lt):
            quad.append(lt[li]); li += 1
        if not quad:
            quad = list(hv[hi:hi + 4]); hi += len(quad)
        out.extend(quad); q += 1
    return np.array(out)
for name, order in (("one costly filter per quad, from the first quad on", interleave(1, 0)), ("one per 2nd quad", interleave(2, 0)), ("one per 4th quad (spread over the whole batch)", interleave(4, 0)),
                    ("one per quad, behind 8192 quads of light filters", interleave(1, 8192))):
    assert len(order) == n and len(set(order.tolist())) == n
    m, mn = run(upload(order))
    print("%-62s walk %.4f ms (min %.4f)  expand %.4f" % (name + ":", m[0], mn[0], m[1]))
# floor or slope?  prefixes of the batch as generated
for k in (1563, 3125, 6250, 12500, 25000, 50000, 100000):
    m = run_subset(np.arange(k))[0]
    print("the first %6d filters of the batch: walk %.4f ms  expand %.4f" % (k, m[0], m[1]))
m, mn = run(base)
print("as generated again:  walk %.4f ms (min %.4f)  expand %.4f" % (m[0], mn[0], m[1]))
