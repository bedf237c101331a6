"""Synthetic workloads of SURVEY.md 8d (ctypes face of bifromq_amd/csrc/bmq_gen.cpp).  Bench/test tooling."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from . import _lib

MODE_LITERAL, MODE_MIXED, MODE_RETAIN_QUERY = 0, 1, 2
SEED_BASE = 0xB1F20000


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    ct = {np.uint8: C.c_uint8, np.uint32: C.c_uint32}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))


class Workload:
    """Route keys for n_tenants x routes_per_tenant (sorted), plus publish batches drawn from them."""

    def __init__(self, seed: int, n_tenants: int, routes_per_tenant: int, mode: int = MODE_MIXED, tenant_base: int = 0,
                 tenant_ids=None):
        """tenant_base: global index of the first tenant of a contiguous block; tenant_ids: explicit ascending list of
        global tenant indices instead (the shard hash(tenantId) mod N gives one rank).  A tenant's routes depend only on
        (seed, global tenant index), so shards of one population are consistent with the unsharded workload."""
        G = _lib.gen()
        if tenant_ids is not None:
            ids = np.ascontiguousarray(sorted(int(i) for i in tenant_ids), dtype=np.uint32)
            n_tenants = len(ids)
            self.h = G.bmqgen_create_list(seed, ids.ctypes.data_as(C.c_void_p), n_tenants, routes_per_tenant, mode)
        else:
            self.h = G.bmqgen_create(seed, tenant_base, n_tenants, routes_per_tenant, mode)
        if not self.h:
            raise MemoryError("workload exceeds the 4 GiB key buffer of the ABI")
        self.n_tenants = n_tenants
        self.n_keys = G.bmqgen_n_keys(self.h)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None and getattr(_lib, "gen", None) is not None:  # (module globals are gone at interpreter exit)
            _lib.gen().bmqgen_destroy(self.h)
            self.h = None

    # views into generator-owned memory (valid while self lives)
    def keys_packed(self) -> Tuple[np.ndarray, np.ndarray]:
        G = _lib.gen()
        off = _arr(G.bmqgen_key_off(self.h), self.n_keys + 1, np.uint32)
        return _arr(G.bmqgen_key_bytes(self.h), int(off[-1]) if self.n_keys else 0, np.uint8), off

    def keys(self) -> List[bytes]:
        data, off = self.keys_packed()
        raw = data.tobytes()
        return [raw[off[i]:off[i + 1]] for i in range(self.n_keys)]

    def tenants_packed(self) -> Tuple[np.ndarray, np.ndarray]:
        G = _lib.gen()
        off = _arr(G.bmqgen_tenant_off(self.h), self.n_tenants + 1, np.uint32)
        return _arr(G.bmqgen_tenant_bytes(self.h), ((int(off[-1]) + 15) & ~15) + 16, np.uint8), off

    def tenants(self) -> List[str]:
        data, off = self.tenants_packed()
        raw = data.tobytes()
        return [raw[off[i]:off[i + 1]].decode() for i in range(self.n_tenants)]

    def tenant_first(self) -> np.ndarray:
        return _arr(_lib.gen().bmqgen_tenant_first(self.h), self.n_tenants + 1, np.uint32).copy()

    def topics(self, seed: int, n_topics: int, tenant_lo: int = 0, tenant_hi: int = 2**32 - 1, hit_permille: int = 900,
               grouped: bool = False):
        """-> (topic bytes (padded), topic offsets, topic_tenant) as numpy COPIES.  topic_tenant indexes THIS
        workload's tenant table.  grouped: ordered by tenant, like the DistPacks of a BatchDistRequest."""
        G = _lib.gen()
        n = G.bmqgen_topics(self.h, seed, n_topics, tenant_lo, min(tenant_hi, self.n_tenants), hit_permille,
                            1 if grouped else 0)
        return self._out(n)

    def retain(self, seed: int, n: int, filters: bool):
        G = _lib.gen()
        return self._out(G.bmqgen_retain(self.h, seed, n, 1 if filters else 0))

    def _out(self, n):
        G = _lib.gen()
        off = _arr(G.bmqgen_topic_off(self.h), n + 1, np.uint32).copy()
        data = _arr(G.bmqgen_topic_bytes(self.h), ((int(off[-1]) + 15) & ~15) + 16 if n else 16, np.uint8).copy()
        tt = _arr(G.bmqgen_topic_tenant(self.h), n, np.uint32).copy()
        return data, off, tt


def unpack(data: np.ndarray, off: np.ndarray) -> List[bytes]:
    raw = data.tobytes()
    return [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]
