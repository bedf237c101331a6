"""Multi-GPU shape of the match path: tenants never interact, so the filter index is partitioned by
tenant across the ranks of a node (one process per GPU) and every rank matches only the publishes of its own tenants.
The single exchange step afterwards is an all-gather of each rank's CSR result over RCCL/xGMI
(`torch.distributed`, backend "nccl" on GPUs, "gloo" in the CPU tests).

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
