"""N > 1 path on CPU: tenant sharding + the one exchange step (all-gather of CSR) with the gloo backend, world_size 2.
Each rank's local result comes from the oracle here (no GPU in this tier); the merged node-wide result must equal the
oracle run over the unsharded key set."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bifromq_amd as B
from bifromq_amd import shard
from bifromq_amd.workload import unpack
from oracle import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = B.Workload(0xB1F20003, 6, 400, 1)
        tn = w.tenants()
        keys = w.keys()
        data, off, tt = w.topics(11, 600)
        topics = unpack(data, off)
        mine = [t for t in tn if shard.tenant_rank(t, world) == rank]
        my_keys = [k for k in keys if B.decode_route_key(k)[1] in mine]
        kv = O.KV(my_keys)  # this rank's shard of the index
        parts = shard.route_batch(tn, tt, world)
        idx = parts[rank]
        rows = [sorted(kv.match_all(tn[tt[i]], [topics[i]]).per_topic()[0]) for i in idx]
        n_pad = max(len(p) for p in parts)  # fixed-size row_ptr per rank for the gather
        row_ptr = np.zeros(n_pad + 1, dtype=np.int32)
        row_ptr[1:len(rows) + 1] = np.cumsum([len(r) for r in rows])
        row_ptr[len(rows) + 1:] = row_ptr[len(rows)]
        ids = np.array([x for r in rows for x in r] or [0], dtype=np.int32)
        total = int(row_ptr[-1])
        rows_all, ids_all, cnts = shard.exchange_csr(dist, torch.from_numpy(row_ptr), torch.from_numpy(ids), total, world)
        merged = shard.merge_rows(parts, rows_all.numpy(), ids_all.numpy(), len(topics))
        # rank-local ids -> keys, so that the result is comparable across shards
        shard_keys = []
        for r in range(world):
            sk = [k for k in keys if shard.tenant_rank(B.decode_route_key(k)[1], world) == r]
            shard_keys.append(sorted(sk))
        owner = [shard.tenant_rank(tn[t], world) for t in tt]
        merged_keys = [[shard_keys[owner[i]][x] for x in row] for i, row in enumerate(merged)]
        q.put((rank, cnts.tolist(), merged_keys))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_exchange_merge():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every rank ends up with the same node-wide result, equal to the unsharded oracle
    w = B.Workload(0xB1F20003, 6, 400, 1)
    keys = w.keys()
    kv = O.KV(keys)
    data, off, tt = w.topics(11, 600)
    tn = w.tenants()
    topics = unpack(data, off)
    exp = [[keys[x] for x in sorted(kv.match_all(tn[tt[i]], [topics[i]]).per_topic()[0])] for i in range(len(topics))]
    for rank, cnts, merged_keys in got:
        assert merged_keys == exp
        assert sum(cnts) == sum(len(r) for r in exp)
    assert got[0][1] == got[1][1]


def test_tenant_hash_is_stable_and_spreads():
    assert shard.tenant_hash("tenantA") == shard.tenant_hash(b"tenantA")
    assert shard.tenant_hash("") == shard.FNV_OFFSET  # FNV-1a 64 offset basis
    ranks = [shard.tenant_rank("tenant%06d" % i, 8) for i in range(8000)]
    counts = np.bincount(ranks, minlength=8)
    assert counts.min() > 800 and counts.max() < 1200
    parts = shard.route_batch(["a", "b", "c"], np.array([0, 1, 2, 0, 0, 2]), 2)
    assert sorted(np.concatenate(parts).tolist()) == [0, 1, 2, 3, 4, 5]


def _worker_split(rank, world, port, q):
    """Node-wide batch, hot tenant split by filter: partition on the 'device' (torch tensors), per-rank match (oracle stands in),
    fan-out all-reduce; plus the all-gatherv of the CSR."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = B.Workload(0xB1F20003, 6, 400, 1)
        tn = w.tenants()
        keys = w.keys()
        key_tenant = np.searchsorted(w.tenant_first(), np.arange(len(keys)), side="right") - 1
        data, off, tt = w.topics(11, 800)
        share = np.bincount(tt, minlength=len(tn)) / len(tt)
        hot = shard.pick_hot_tenants(share, world, 0.5)
        assert hot, "the Zipf head must qualify as hot in this workload"
        my_keys = shard.shard_keys(keys, key_tenant, tn, hot, world, rank)
        kv = O.KV(my_keys)
        owner = torch.from_numpy(shard.topic_targets(tn, hot, world))
        sel, d2, o2, tt2 = shard.partition_batch(owner, torch.from_numpy(tt.astype(np.int64)), torch.from_numpy(data),
                                                 torch.from_numpy(off.astype(np.int64)), rank)
        raw = d2.numpy().tobytes()
        o2n = o2.numpy()
        my_topics = [raw[o2n[i]:o2n[i + 1]] for i in range(len(sel))]
        all_topics = unpack(data, off)
        assert my_topics == [all_topics[int(g)] for g in sel]
        rows = [sorted(kv.match_all(tn[int(t)], [tp]).per_topic()[0]) for t, tp in zip(tt2.numpy(), my_topics)]
        counts = torch.tensor([len(r) for r in rows], dtype=torch.int32)
        fan = shard.exchange_fanout(dist, counts, sel, len(tt))
        row_ptr = np.zeros(len(rows) + 1, dtype=np.int32)
        row_ptr[1:] = np.cumsum([len(r) for r in rows])
        # all-gatherv needs equal-sized row pointer vectors: pad to the largest part
        n_max = torch.tensor([len(rows)], dtype=torch.int64)
        dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
        rp = np.full(int(n_max) + 1, row_ptr[-1], dtype=np.int32)
        rp[:len(row_ptr)] = row_ptr
        ids = np.array([x for r in rows for x in r] or [0], dtype=np.int32)
        rows_all, ids_list, totals = shard.exchange_csr_v(dist, torch.from_numpy(rp), torch.from_numpy(ids), int(row_ptr[-1]), world)
        q.put((rank, hot, fan.tolist(), totals, [t.tolist() for t in ids_list], sel.tolist(), len(my_keys)))
    finally:
        dist.destroy_process_group()


def test_two_rank_hot_tenant_split_and_fanout_exchange():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_split, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    w = B.Workload(0xB1F20003, 6, 400, 1)
    keys = w.keys()
    kv = O.KV(keys)
    data, off, tt = w.topics(11, 800)
    tn = w.tenants()
    topics = unpack(data, off)
    exp = [len(kv.match_all(tn[tt[i]], [topics[i]]).per_topic()[0]) for i in range(len(topics))]
    assert got[0][2] == exp and got[1][2] == exp             # the summed fan-out equals the unsharded oracle on every rank
    assert got[0][6] + got[1][6] == len(keys)                 # the shards partition the route keys (hot tenant split by key hash)
    hot = set(got[0][1])
    both = set(got[0][5]) & set(got[1][5])
    assert both == {i for i in range(len(tt)) if int(tt[i]) in hot}   # split tenants' publishes go to every rank, the rest to one
    assert got[0][3] == got[1][3] and got[0][4] == got[1][4]  # all-gatherv: every rank holds every rank's ids, exact sizes
    assert [len(x) for x in got[0][4]] == got[0][3] and sum(got[0][3]) == sum(exp)



def test_fanout_split_hinter_follows_the_mutations():
    """DW/hinter/FanoutSplitHinter.java restated over tenants (bifromq_amd/shard.py): a tenant is split by filter when its live routes
    reach split_at_scale, merged back below half of it, and after every executed decision the ranks' shards still partition the route
    keys and the summed per-rank fan-outs equal the unsharded oracle."""
    world, scale = 3, 40
    rng = np.random.default_rng(77)
    h = shard.FanoutSplitHinter(world, scale)
    tenants = ["tenant%02d" % i for i in range(5)]
    live = {t: set() for t in tenants}                    # the node-wide truth
    held = [{t: set() for t in tenants} for _ in range(world)]  # what every rank indexes
    serial = 0
    decisions = []

    def apply(ops):  # ops: (tenant, key, delete)
        eff = []
        for t, k, d in ops:
            if d and k in live[t]:
                live[t].discard(k)
                held[h.key_owner(t, k)][t].discard(k)
                eff.append((t, True))
            elif not d and k not in live[t]:
                live[t].add(k)
                held[h.key_owner(t, k)][t].add(k)
                eff.append((t, False))
        to_split, to_merge = h.record_mutate([e[0] for e in eff], [e[1] for e in eff])
        for t, s in [(t, True) for t in to_split] + [(t, False) for t in to_merge]:
            moves = h.plan(t, s, [sorted(held[r][t]) for r in range(world)])
            for r, (dels, adds) in enumerate(moves):
                held[r][t] -= set(dels)
                held[r][t] |= set(adds)
            decisions.append((t, s))

    def check():
        for t in tenants:
            parts = [held[r][t] for r in range(world)]
            assert set().union(*parts) == live[t] and sum(len(p) for p in parts) == len(live[t])  # a partition of the tenant's keys
            if t in h.split:
                assert all(shard.key_rank(k, world) == r for r in range(world) for k in parts[r])
                assert h.owner(t) == -1
            else:
                own = shard.tenant_rank(t, world)
                assert all(not parts[r] for r in range(world) if r != own) and h.owner(t) == own
            assert h.routes.get(t, 0) == len(live[t])
        # fan-out: every rank matches the publishes it is a target of against its shard; the sums equal the unsharded oracle
        whole = O.KV(sorted(k for t in tenants for k in live[t]))
        shards = [O.KV(sorted(k for t in tenants for k in held[r][t])) for r in range(world)]
        for t in tenants:
            for topic in (b"a/b/c", b"a/x", b"q"):
                exp = len(whole.match_all(t, [topic]).per_topic()[0])
                targets = range(world) if h.owner(t) < 0 else [h.owner(t)]
                assert sum(len(shards[r].match_all(t, [topic]).per_topic()[0]) for r in targets) == exp

    def new_keys(t, n):
        nonlocal serial
        out = []
        for _ in range(n):
            serial += 1
            f = ["a/b/c", "a/+/c", "a/#", "+/x", "#", "q"][serial % 6]
            out.append(B.route_key(t, f, 1, "0\0inbox%d\0d%d" % (serial, serial % 7)))
        return out

    # grow tenant 0 past the scale in three batches, the others stay small
    for n in (15, 15, 15):
        apply([(tenants[0], k, False) for k in new_keys(tenants[0], n)] + [(t, k, False) for t in tenants[1:] for k in new_keys(t, 3)])
        check()
    assert decisions == [(tenants[0], True)] and h.split == {tenants[0]}
    assert h.load() == {"fanout_topicfilters": 1, "fanout_scale": 45}
    # a repeated subscribe changes nothing and is not recorded
    k0 = sorted(live[tenants[0]])[0]
    apply([(tenants[0], k0, False)])
    assert h.routes[tenants[0]] == 45
    # shrink it: still split at 20 routes (half the scale), merged back at 19
    victims = sorted(live[tenants[0]])
    apply([(tenants[0], k, True) for k in victims[:25]])
    check()
    assert h.split == {tenants[0]} and len(live[tenants[0]]) == 20
    apply([(tenants[0], victims[25], True)])
    check()
    assert decisions[-1] == (tenants[0], False) and not h.split
    # random churn across all tenants: decisions come and go, the invariants hold after every batch
    for _ in range(30):
        ops = []
        for t in tenants:
            if rng.random() < 0.6:
                ops += [(t, k, False) for k in new_keys(t, int(rng.integers(1, 30)))]
            if live[t] and rng.random() < 0.5:
                ks = sorted(live[t])
                ops += [(t, ks[int(i)], True) for i in rng.choice(len(ks), size=min(len(ks), int(rng.integers(1, 40))), replace=False)]
        apply(ops)
        check()
    assert len(decisions) > 6 and any(not s for _, s in decisions[2:])
    # one rank: nothing is ever split
    h1 = shard.FanoutSplitHinter(1, 2)
    assert h1.record_mutate(["t"] * 10, [False] * 10) == ([], []) and h1.owner("t") == 0
