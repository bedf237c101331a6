"""Multi-GPU shape of the match path: tenants never interact, so the filter index is partitioned by
tenant across the ranks of a node (one process per GPU) and every rank matches only the publishes of its own tenants.
The single exchange step afterwards goes over RCCL/xGMI (`torch.distributed`, backend "nccl" on GPUs -- which IS RCCL on
ROCm --, "gloo" in the CPU tests), in one of two forms:
  * exchange_fanout: per-topic fan-out counts (4 B per topic).  That is all the reference sends upstream: the dist worker
    that matched delivers itself, BatchDistReply carries fan-out per topic (DW/DistWorkerCoProc.java:535-538) and dist-server
    SUMS the fan-outs of the ranges a tenant spans (BatchDistServerCall.java:186-205);
  * exchange_csr_v: the complete (topic -> route ids) CSR of every rank on every rank, as a true all-gatherv: counts first,
    then one grouped broadcast per rank with its exact size (no padding to the largest rank).
Hot tenants (Zipf) are split by FILTER: their route keys are spread over all ranks by hash(route key) mod N and their
publishes go to every rank; the per-rank fan-outs add up (the reference's analogue: FanoutSplitHinter splits a hot range,
DW/hinter/FanoutSplitHinter.java:49, and dist-server sums).

The reference's analogue is range sharding of the tenant-prefixed key space across KV ranges
(bifromq-dist-worker-spi SplitKey.java:34-57) with dist-server fanning a batch out per range
(BatchDistServerCall.java:127-165) and summing the per-range fan-outs (:186-205).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

FNV_OFFSET, FNV_PRIME, MASK64 = 0xCBF29CE484222325, 0x100000001B3, (1 << 64) - 1


def tenant_hash(tenant) -> int:
    """FNV-1a 64 over the UTF-8 bytes of the tenant id."""
    b = tenant if isinstance(tenant, (bytes, bytearray)) else tenant.encode("utf-8")
    h = FNV_OFFSET
    for c in b:
        h = ((h ^ c) * FNV_PRIME) & MASK64
    return h


def tenant_rank(tenant, world: int) -> int:
    return tenant_hash(tenant) % world


def route_batch(tenants: Sequence, topic_tenant: np.ndarray, world: int) -> List[np.ndarray]:
    """Indices of the batch's topics each rank has to match (publishes follow their tenant's shard)."""
    owner = np.array([tenant_rank(t, world) for t in tenants], dtype=np.int64)
    topic_owner = owner[np.asarray(topic_tenant, dtype=np.int64)]
    return [np.nonzero(topic_owner == r)[0] for r in range(world)]


def exchange_csr(dist, row_ptr, ids, total: int, world: int):
    """All-gather of every rank's CSR (row_ptr[n+1] int32 tensor, ids tensor with >= total valid entries).
    Returns (rows_all [world, n+1], ids_all [world, max_total], totals [world]).  Three collectives per batch:
    counts (8 B per rank), row pointers (fixed size), ids padded to the largest rank."""
    import torch

    dev = row_ptr.device
    cnt = torch.tensor([total], dtype=torch.int64, device=dev)
    cnts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt)
    mx = max(int(cnts.max().item()), 1)
    rows_all = torch.empty(world * row_ptr.numel(), dtype=row_ptr.dtype, device=dev)
    dist.all_gather_into_tensor(rows_all, row_ptr.contiguous())
    if ids.numel() < mx:
        ids = torch.cat([ids, torch.zeros(mx - ids.numel(), dtype=ids.dtype, device=dev)])
    ids_all = torch.empty(world * mx, dtype=ids.dtype, device=dev)
    dist.all_gather_into_tensor(ids_all, ids[:mx].contiguous())
    return rows_all.view(world, -1), ids_all.view(world, mx), cnts


def merge_rows(parts: Sequence[np.ndarray], rows_all: np.ndarray, ids_all: np.ndarray, n_topics: int) -> List[List[int]]:
    """Undo route_batch: parts[r][k] is the global index of rank r's k-th topic -> per-topic id lists in batch order.
    Ids are rank-local route ids (ranks of keys inside the rank's shard)."""
    out: List[List[int]] = [[] for _ in range(n_topics)]
    for r, idx in enumerate(parts):
        rp = rows_all[r]
        for k, g in enumerate(idx):
            out[int(g)] = ids_all[r][rp[k]:rp[k + 1]].tolist()
    return out


# ---- fan-out exchange, all-gatherv, hot-tenant split, device-side partition --------------------------------------------
def key_rank(route_key: bytes, world: int) -> int:
    """Owner of ONE route key of a split (hot) tenant: FNV-1a 64 of the key bytes mod N."""
    return tenant_hash(route_key) % world


def pick_hot_tenants(publish_share: Sequence[float], world: int, factor: float = 0.5) -> List[int]:
    """Tenant indices whose share of the publishes exceeds `factor` of a rank's fair share (1 / world): these are split by filter.
    (The reference decides from observed fan-out load: DW/hinter/FanoutSplitHinter.java:49.)"""
    if world <= 1:
        return []
    return [i for i, s in enumerate(publish_share) if s > factor / world]


def shard_keys(keys: Sequence[bytes], key_tenant: Sequence[int], tenant_names: Sequence, hot: Sequence[int], world: int, rank: int):
    """Route keys rank `rank` indexes: every key of the tenants it owns + its hash share of the hot tenants' keys."""
    hot = set(int(h) for h in hot)
    owner = [tenant_rank(t, world) for t in tenant_names]
    return [k for k, t in zip(keys, key_tenant) if (key_rank(k, world) if int(t) in hot else owner[int(t)]) == rank]


def topic_targets(tenant_names: Sequence, hot: Sequence[int], world: int):
    """owner[t] = rank that matches tenant t's publishes, or -1 = every rank (split tenant)."""
    hot = set(int(h) for h in hot)
    return np.array([-1 if i in hot else tenant_rank(t, world) for i, t in enumerate(tenant_names)], dtype=np.int64)


def partition_batch(owner, topic_tenant, data, off, rank: int):
    """This rank's part of a node-wide publish batch, computed where the batch lives (torch tensors on the GPU -- or on the CPU
    in the tests): topics whose tenant this rank owns or whose tenant is split.  Plain tensor ops: one mask, one compaction of
    the offsets, one gather of the bytes.
    owner: int64 [n_tenants] (topic_targets); topic_tenant: int [n]; data: uint8 [bytes]; off: int [n + 1].
    -> (sel [m] global topic indices, data' uint8 padded by 32 bytes, off' int32 [m + 1], topic_tenant' [m])"""
    import torch

    tt = topic_tenant.long()
    o = owner[tt]
    sel = torch.nonzero((o == rank) | (o < 0)).flatten()
    off64 = off.long()
    lens = (off64[1:] - off64[:-1])[sel]
    new_off = torch.zeros(sel.numel() + 1, dtype=torch.int64, device=off.device)
    torch.cumsum(lens, 0, out=new_off[1:])
    total = int(new_off[-1].item()) if sel.numel() else 0
    src = torch.repeat_interleave(off64[sel] - new_off[:-1], lens) + torch.arange(total, device=off.device)
    new_data = torch.zeros(total + 32, dtype=torch.uint8, device=data.device)
    if total:
        new_data[:total] = data[src]
    return sel, new_data, new_off.to(torch.int32), topic_tenant[sel].contiguous()


class DevicePartition:
    """bmq_partition_batch_dev (include/bmq.h): the same partition as partition_batch, by kernels of the library on the engine stream
    -- mask, two prefix sums, scatter; one host read (the sizes) per batch instead of a chain of tensor ops with a .item() in the middle.
    Buffers are allocated once for batches of up to n topics / `nbytes` topic bytes."""

    def __init__(self, eng, owner, n: int, nbytes: int, device):
        import torch

        self.eng = eng
        self.owner = owner.to(torch.int32).contiguous()
        self.sel = torch.zeros(n, dtype=torch.int32, device=device)
        self.data = torch.zeros(nbytes + 64, dtype=torch.uint8, device=device)
        self.off = torch.zeros(n + 1, dtype=torch.int32, device=device)
        self.tt = torch.zeros(n, dtype=torch.int32, device=device)

    def __call__(self, topic_tenant, data, off, rank: int):
        """-> (sel [m], data', off' [m + 1], topic_tenant' [m]) views of the internal buffers, m"""
        import ctypes as C

        from . import _lib

        n = int(topic_tenant.numel())
        m, nb = C.c_uint32(), C.c_uint64()
        rc = _lib.lib().bmq_partition_batch_dev(self.eng.h, self.owner.data_ptr(), int(self.owner.numel()), rank, topic_tenant.data_ptr(), data.data_ptr(),
                                                off.data_ptr(), n, self.sel.data_ptr(), self.data.data_ptr(), self.off.data_ptr(), self.tt.data_ptr(),
                                                C.byref(m), C.byref(nb))
        if rc:
            raise RuntimeError("bmq_partition_batch_dev failed: %d %s" % (rc, _lib.lib().bmq_last_error(self.eng.h)))
        m = int(m.value)
        return self.sel[:m], self.data, self.off[:m + 1], self.tt[:m], m


def exchange_fanout(dist, counts_local, sel, n_global: int):
    """Node-wide per-topic fan-out: every rank adds the fan-outs of the topics it matched into a vector over the whole batch
    (one all-reduce SUM of 4 B per topic -- split tenants are matched by every rank against its share of the filters, so their
    counts add up, exactly what dist-server does with per-range fan-outs).  counts_local: int32 [m]; sel: [m] global indices."""
    import torch

    v = torch.zeros(n_global, dtype=torch.int32, device=counts_local.device)
    if sel.numel():
        v[sel.long()] = counts_local.to(torch.int32)
    if dist is not None:
        dist.all_reduce(v)
    return v


def exchange_counts_weak(dist, row_ptr, world: int):
    """Every rank matched its OWN batch of n topics (weak scaling): all-gather of the n fan-out counts of every rank ->
    [world, n].  No host synchronisation, no padding: every rank contributes exactly n values."""
    import torch

    counts = (row_ptr[1:] - row_ptr[:-1]).contiguous()
    out = torch.empty(world * counts.numel(), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(out, counts)
    return out.view(world, -1)


def exchange_csr_v(dist, row_ptr, ids, total: int, world: int):
    """True all-gatherv of every rank's CSR: totals (one small all-gather + one host read), row pointers (fixed size), then the
    ids with their EXACT sizes -- torch.distributed.all_gather with unequal output tensors, which the NCCL/RCCL backend runs
    as one group of per-rank broadcasts.  gloo (CPU tests) cannot take unequal sizes: padded there.
    -> (rows_all [world, n + 1], list of world id tensors, totals list)"""
    import torch

    dev = row_ptr.device
    cnt = torch.tensor([total], dtype=torch.int64, device=dev)
    cnts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt)
    totals = [int(x) for x in cnts.cpu().tolist()]
    rows_all = torch.empty(world * row_ptr.numel(), dtype=row_ptr.dtype, device=dev)
    dist.all_gather_into_tensor(rows_all, row_ptr.contiguous())
    mine = ids[:total].contiguous()
    if dist.get_backend() == "nccl":
        outs = [torch.empty(max(t, 0), dtype=ids.dtype, device=dev) for t in totals]
        dist.all_gather(outs, mine)
    else:
        mx = max(max(totals), 1)
        pad = torch.zeros(mx, dtype=ids.dtype, device=dev)
        pad[:total] = mine
        flat = torch.empty(world * mx, dtype=ids.dtype, device=dev)
        dist.all_gather_into_tensor(flat, pad)
        outs = [flat[r * mx:r * mx + totals[r]] for r in range(world)]
    return rows_all.view(world, -1), outs, totals



# ---- dynamic split decisions ----------------------------------------------------------------------------------------
class FanoutSplitHinter:
    """Which tenants are split by filter, decided continuously from the route mutations that pass by -- DW/hinter/FanoutSplitHinter.java
    restated for a node whose shards are tenants instead of KV ranges:
      * the reference looks at mutations only (recordQuery is empty, :81-83): every BatchMatch / BatchUnmatch re-estimates the record
        count under the touched key prefix (doEstimate, :174-204) -- here the prefix is the tenant (all of a tenant's route keys share
        it: SplitKey.java:34-57), the record count its live routes;
      * a prefix whose scale reaches `split_at_scale` becomes a split candidate (:183-185), one that falls below HALF of it is dropped
        again (:186-188): the same hysteresis decides here when a split tenant is merged back;
      * the reference hands ONE split key at a time to base-kv (estimate(), :143-158); here a decision is executed by moving route
        keys between the ranks' indexes: `plan()` lists, per rank, the keys to delete and the keys to add (bmq_routes_apply ops).
    Ranks run the same hinter over the same mutation stream (the ops are broadcast to the owners anyway), so they agree without talking."""

    def __init__(self, world: int, split_at_scale: int):
        if split_at_scale < 2:
            raise ValueError("split_at_scale must be at least 2")
        self.world = int(world)
        self.split_at_scale = int(split_at_scale)
        self.routes = {}    # tenant -> live routes
        self.split = set()  # tenants split by filter right now

    def owner(self, tenant) -> int:
        """rank that matches the tenant's publishes, or -1 = every rank"""
        return -1 if tenant in self.split and self.world > 1 else tenant_rank(tenant, self.world)

    def key_owner(self, tenant, route_key: bytes) -> int:
        return key_rank(route_key, self.world) if tenant in self.split and self.world > 1 else tenant_rank(tenant, self.world)

    def record_mutate(self, tenants: Sequence, deletes: Sequence[bool]):
        """One batch of mutations (recordMutate, :86-129): deletes[i] tells an unsubscribe from a subscribe.  Only mutations that CHANGED
        the index may be recorded (a repeated subscribe of an existing route adds no record).
        -> (tenants to split, tenants to merge back) -- the decisions this batch triggers; `plan()` turns one into key moves."""
        touched = {}
        for t, d in zip(tenants, deletes):
            touched[t] = touched.get(t, 0) + (-1 if d else 1)
        return self.record_counts(touched)

    def record_counts(self, touched):
        """The same for a batch that is already summed up: {tenant: routes added - routes removed} (a bulk load is one such batch)."""
        to_split, to_merge = [], []
        for t, delta in touched.items():
            n = max(self.routes.get(t, 0) + delta, 0)
            if n:
                self.routes[t] = n
            else:
                self.routes.pop(t, None)
            if self.world <= 1:
                continue
            if n >= self.split_at_scale and t not in self.split:
                to_split.append(t)
            elif t in self.split and 2 * n < self.split_at_scale:
                to_merge.append(t)
        return sorted(to_split, key=tenant_hash), sorted(to_merge, key=tenant_hash)

    def plan(self, tenant, split: bool, keys_of_rank: Sequence[Sequence[bytes]]):
        """Executes one decision: keys_of_rank[r] = the tenant's route keys rank r holds now.  -> per rank (keys to delete, keys to add);
        the tenant's placement (`owner`, `key_owner`) switches with this call."""
        if split:
            self.split.add(tenant)
        else:
            self.split.discard(tenant)
        dels = [[] for _ in range(self.world)]
        adds = [[] for _ in range(self.world)]
        for r, ks in enumerate(keys_of_rank):
            for k in ks:
                to = self.key_owner(tenant, k)
                if to != r:
                    dels[r].append(k)
                    adds[to].append(k)
        return [(sorted(d), sorted(a)) for d, a in zip(dels, adds)]

    def load(self):
        """what the reference reports as SplitHint load (:146-148): split prefixes, and their summed scale"""
        return {"fanout_topicfilters": len(self.split), "fanout_scale": sum(self.routes.get(t, 0) for t in self.split)}
