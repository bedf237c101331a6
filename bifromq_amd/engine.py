"""Python face of the C-ABI engine (include/bmq.h).  Thin: packs strings, calls libbmq.so, unpacks CSR.

All matching happens in the HIP kernels behind the ABI (bifromq_amd/csrc).  Nothing here matches topics.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

INT_MAX = 2**31 - 1
STATUS = {0: "OK", -1: "INVAL", -2: "NODEVICE", -3: "NOSPACE", -4: "NOMEM", -5: "HIP", -6: "RANGE", -7: "STATE"}


class BmqError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bmq error {STATUS.get(code, code)}: {msg}")
        self.code = code


def _b(s) -> bytes:
    return s if isinstance(s, (bytes, bytearray)) else s.encode("utf-8")


def pack(strings: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """list of str/bytes -> (uint8 bytes padded by >= 16 zero bytes, uint32 offsets[n+1])"""
    bs = [_b(s) for s in strings]
    off = np.zeros(len(bs) + 1, dtype=np.uint32)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64).astype(np.uint32)
    raw = b"".join(bs)
    data = np.zeros(((len(raw) + 15) & ~15) + 16, dtype=np.uint8)
    if raw:
        data[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    return data, off


def pinned(shape, dtype) -> np.ndarray:
    """numpy array over page-locked host memory from bmq_host_alloc (never freed explicitly here: benchmark / test helper)."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = _lib.lib().bmq_host_alloc(max(n, 16))
    if not p:
        raise MemoryError("bmq_host_alloc")
    buf = (C.c_uint8 * max(n, 16)).from_address(p)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- codec (SCHEMA/KVSchemaUtil.java:91-130) -----------------------------------------------------------
def route_key(tenant, topic_filter, flag: int, receiver) -> bytes:
    t, f, r = _b(tenant), _b(topic_filter), _b(receiver)
    out = C.create_string_buffer(len(t) + len(f) + len(r) + 16)
    n = _lib.lib().bmq_route_key_encode(t, len(t), f, len(f), flag, r, len(r), out, len(out))
    return out.raw[:n]


def route_key_from_mqtt(tenant, mqtt_topic_filter: str, receiver_url: str = "") -> bytes:
    """TopicUtil.from + toNormalRouteKey / toGroupRouteKey (UTIL/TopicUtil.java:252-272)."""
    for prefix, flag in (("$share/", 2), ("$oshare/", 3)):
        if mqtt_topic_filter.startswith(prefix):
            rest = mqtt_topic_filter[len(prefix):]
            i = rest.index("/")
            return route_key(tenant, rest[i + 1:], flag, rest[:i])
    return route_key(tenant, mqtt_topic_filter, 1, receiver_url)


def decode_route_key(key: bytes):
    """-> (flag, tenant, mqttTopicFilter incl. $share prefix, receiver) or None (RouteDetailCache.java:53-109)."""
    spans = (C.c_uint32 * 6)()
    flag = _lib.lib().bmq_route_key_decode(key, len(key), spans)
    if flag < 0:
        return None
    tenant = key[spans[0]:spans[0] + spans[1]].decode()
    filt = key[spans[2]:spans[2] + spans[3]].replace(b"\0", b"/").decode()
    recv = key[spans[4]:spans[4] + spans[5]].decode()
    if flag == 2:
        filt = f"$share/{recv}/{filt}"
    elif flag == 3:
        filt = f"$oshare/{recv}/{filt}"
    return flag, tenant, filt, recv


def retain_message_key(tenant, topic) -> bytes:
    """KVSchemaUtil.retainMessageKey of the retain store schema"""
    t, p = _b(tenant), _b(topic)
    out = C.create_string_buffer(len(t) + 2 * len(p) + 16)
    n = _lib.lib().bmq_retain_message_key(t, len(t), p, len(p), out, len(out))
    return out.raw[:n]


def retain_filter_route(tenant, topic_filter):
    """-> (key or key prefix, LevelHash(filterPrefix), levels, has_wildcard, ends_with_hash): MatchCallRangeRouter's per-filter part"""
    t, f = _b(tenant), _b(topic_filter)
    out, hsh = C.create_string_buffer(len(t) + 2 * len(f) + 16), C.create_string_buffer(len(f) + 2)
    kl, hl, lv = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = _lib.lib().bmq_retain_filter_route(t, len(t), f, len(f), out, len(out), C.byref(kl), hsh, len(hsh), C.byref(hl), C.byref(lv))
    if rc < 0:
        raise BmqError(rc, "bmq_retain_filter_route")
    return out.raw[:kl.value], hsh.raw[:hl.value], lv.value, bool(rc & 1), bool(rc & 2)


class RangeRouter:
    """The client-side KV range router: boundaries (start | None, end | None) in BoundaryUtil.compare order
    (KVRangeRouterUtil.java:41-103); lookups return indices into that list."""

    def __init__(self, boundaries):
        self.n = len(boundaries)
        self.flags = np.array([(1 if s is not None else 0) | (2 if e is not None else 0) for s, e in boundaries], dtype=np.uint8)
        self.start, self.start_off = pack([s or b"" for s, _ in boundaries])
        self.end, self.end_off = pack([e or b"" for _, e in boundaries])

    def _args(self):
        return (_ptr(self.flags), _ptr(self.start), _ptr(self.start_off), _ptr(self.end), _ptr(self.end_off), self.n)

    def build(self):
        """bmq_router_create: copy, check and index the boundaries once; the lookups below then go through the object."""
        if getattr(self, "h", None) is None:
            h = C.c_void_p()
            rc = _lib.lib().bmq_router_create(*self._args(), C.byref(h))
            if rc < 0:
                raise BmqError(rc, "bmq_router_create")
            self.h = h
        return self

    def close(self):
        if getattr(self, "h", None) is not None:
            _lib.lib().bmq_router_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def find_by_key(self, key: bytes):
        out = C.c_int32()
        if getattr(self, "h", None) is not None:
            rc = _lib.lib().bmq_router_lookup_key(self.h, key, len(key), C.byref(out))
        else:
            rc = _lib.lib().bmq_router_find_by_key(*self._args(), key, len(key), C.byref(out))
        if rc < 0:
            raise BmqError(rc, "bmq_router_find_by_key")
        return None if out.value < 0 else out.value

    def find_by_boundary(self, start, end):
        first, count = C.c_uint32(), C.c_uint32()
        fl = (1 if start is not None else 0) | (2 if end is not None else 0)
        q = (fl, start or b"", len(start or b""), end or b"", len(end or b""), C.byref(first), C.byref(count))
        if getattr(self, "h", None) is not None:
            rc = _lib.lib().bmq_router_lookup_boundary(self.h, *q)
        else:
            rc = _lib.lib().bmq_router_find_by_boundary(*self._args(), *q)
        if rc < 0:
            raise BmqError(rc, "bmq_router_find_by_boundary")
        return list(range(first.value, first.value + count.value))

    def retain_range_lookup(self, tenant, topic_filters, exact: bool = False):
        """MatchCallRangeRouter.rangeLookup -> per filter the indices of the ranges it is sent to (exact: BMQ_ROUTER_EXACT)"""
        t = _b(tenant)
        fb, fo = pack(topic_filters)
        keep = np.zeros((len(topic_filters), self.n), dtype=np.uint8)
        if getattr(self, "h", None) is not None:
            rc = _lib.lib().bmq_router_retain_lookup(self.h, t, len(t), _ptr(fb), _ptr(fo), len(topic_filters), 1 if exact else 0, _ptr(keep))
        else:
            rc = _lib.lib().bmq_retain_range_lookup(t, len(t), _ptr(fb), _ptr(fo), len(topic_filters), *self._args(), 1 if exact else 0, _ptr(keep))
        if rc < 0:
            raise BmqError(rc, "bmq_retain_range_lookup")
        return [np.nonzero(keep[i])[0].tolist() for i in range(len(topic_filters))]


def java_string_hash(s) -> int:
    b = _b(s)
    return _lib.lib().bmq_java_string_hash(b, len(b))


class Engine:
    """One engine per KV range replica (cf. DistWorkerCoProc's SubscriptionCache)."""

    def __init__(self, device: int = 0, wave_queue_cap: int = 0, wave_pair_cap: int = 0, slow_scratch_mb: int = 0, kernel_timing: bool = False,
                 dedup_min_topics: int = 0, dedup_sorted: bool = False, region_slack: int = 0):
        L = _lib.lib()
        cfg = _lib.Config()
        cfg.struct_size = C.sizeof(_lib.Config)
        cfg.device = device
        cfg.wave_queue_cap = wave_queue_cap
        cfg.wave_pair_cap = wave_pair_cap
        cfg.slow_scratch_mb = slow_scratch_mb
        cfg.kernel_timing = 1 if kernel_timing else 0
        cfg.dedup_min_topics = dedup_min_topics  # 0: default = never; n: batches of >= n topics are de-duplicated on the device first
        cfg.dedup_sorted = 1 if dedup_sorted else 0  # ... by comparing neighbours: the batches arrive ordered by (tenant index, topic)
        cfg.region_slack = region_slack  # 0: default (6: trie regions at load factor 0.2); 1: 0.4 (less memory, more second probes)
        h = C.c_void_p()
        rc = L.bmq_engine_create(C.byref(cfg), C.byref(h))
        if rc:
            raise BmqError(rc, "bmq_engine_create failed (a gfx950 device is required for device >= 0)")
        self.h = h
        self.device = device
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().bmq_engine_destroy(self.h)
            self.h = None
            for arena in getattr(self, "_apply_arenas", ()) if self.device >= 0 else ():  # (apply_async's page-locked buffers: nothing reads them any more)
                for a in arena or ():
                    _lib.lib().bmq_host_free(C.c_void_p(a.ctypes.data))
            self._apply_arenas = [None, None]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise BmqError(rc, (_lib.lib().bmq_last_error(self.h) or b"").decode())

    # ---- index ---------------------------------------------------------------------------------------
    def rebuild(self, keys: Iterable[bytes] = (), packed: Optional[Tuple[np.ndarray, np.ndarray]] = None):
        data, off = pack(list(keys)) if packed is None else packed
        self._check(_lib.lib().bmq_rebuild(self.h, _ptr(data), _ptr(off), len(off) - 1))
        return self

    def rebuild_raw(self, key_bytes_ptr: int, key_off_ptr: int, n: int):
        self._check(_lib.lib().bmq_rebuild(self.h, key_bytes_ptr, key_off_ptr, n))
        return self

    def apply(self, ops: Sequence[Tuple[int, bytes]]):
        """ops: (0 = put | 1 = delete, route key)"""
        data, off = pack([k for _, k in ops])
        op = np.array([o for o, _ in ops], dtype=np.uint8)
        self._check(_lib.lib().bmq_routes_apply(self.h, _ptr(data), _ptr(off), _ptr(op), len(ops)))
        return self

    def apply_async(self, ops: Sequence[Tuple[int, bytes]]):
        """bmq_routes_apply_async: the batch is uploaded and queued behind whatever the engine stream holds, the call returns at once; its
        outcome is apply_wait()'s (or the next index call's).  The op buffers are kept in page-locked memory owned by this object until then."""
        data, off = pack([k for _, k in ops])
        op = np.array([o for o, _ in ops], dtype=np.uint8)
        # Two page-locked arenas taken in turns, grow-only (a fresh set per call was never freed).  The engine reads batch k's
        # buffers until batch k is completed -- by bmq_routes_apply_wait or inside the NEXT bmq_routes_apply_async --, so call k + 1 must not
        # touch arena k: it takes the other one, whose batch k - 1 was completed during call k.
        if not hasattr(self, "_apply_arenas"):
            self._apply_arenas, self._apply_turn = [None, None], 0
        self._apply_turn ^= 1
        need = (len(data), len(off) * 4, max(len(op), 1))
        arena = self._apply_arenas[self._apply_turn]
        if arena is None or any(a.nbytes < n for a, n in zip(arena, need)):
            if arena is not None:
                for a in arena:
                    _lib.lib().bmq_host_free(C.c_void_p(a.ctypes.data))
            # (a host-only engine has no page-locked allocator -- and nothing to upload: ordinary memory, kept alive the same way)
            arena = tuple((pinned if self.device >= 0 else np.empty)(max(2 * n, 4096), np.uint8) for n in need)
            self._apply_arenas[self._apply_turn] = arena
        b_data, b_off, b_op = arena[0][:len(data)], arena[1][:len(off) * 4].view(np.uint32), arena[2][:max(len(op), 1)]
        b_data[:], b_off[:], b_op[:len(op)] = data, off, op
        self._check(_lib.lib().bmq_routes_apply_async(self.h, _ptr(b_data), _ptr(b_off), _ptr(b_op), len(ops)))
        return self

    def apply_wait(self):
        """bmq_routes_apply_wait: the outcome of the batch handed over with apply_async (raises what apply would have raised)"""
        self._check(_lib.lib().bmq_routes_apply_wait(self.h))
        return self

    POLLER_DISABLE, POLLER_ENABLE, POLLER_STOP, POLLER_TEST_IGNORE_DOORBELLS = 0, 1, 2, 3

    def poller_stats(self) -> _lib.PollerStats:
        """bmq_poller_stats_get: the persistent matcher behind the batching front (k_poll)"""
        out = _lib.PollerStats()
        self._check(_lib.lib().bmq_poller_stats_get(self.h, C.byref(out)))
        return out

    def poller_control(self, what: int):
        self._check(_lib.lib().bmq_poller_control(self.h, what))
        return self

    def compact(self):
        """bmq_compact: re-build from the live routes (ids become ranks again, new generation)."""
        self._check(_lib.lib().bmq_compact(self.h))
        return self

    def compact_begin(self):
        """bmq_compact_begin: the next generation of the route index starts being built beside the serving one"""
        self._check(_lib.lib().bmq_compact_begin(self.h))
        return self

    def compact_poll(self, max_ids: int = 8192) -> int:
        """bmq_compact_poll: hands the next max_ids route ids' live keys to the builder -> progress in permille (1000: ready to swap)"""
        done = C.c_uint32()
        self._check(_lib.lib().bmq_compact_poll(self.h, max_ids, C.byref(done)))
        return int(done.value)

    def compact_swap(self) -> Tuple[int, int]:
        """bmq_compact_swap: replays what was mutated meanwhile, swaps the generations -> (keys carried over, ops replayed)"""
        carried, replayed = C.c_uint64(), C.c_uint64()
        self._check(_lib.lib().bmq_compact_swap(self.h, C.byref(carried), C.byref(replayed)))
        return int(carried.value), int(replayed.value)

    def compact_abort(self):
        self._check(_lib.lib().bmq_compact_abort(self.h))
        return self

    def info(self) -> _lib.IndexInfo:
        out = _lib.IndexInfo()
        self._check(_lib.lib().bmq_index_info_get(self.h, C.byref(out)))
        return out

    def route_key(self, route_id: int) -> bytes:
        n = C.c_uint32()
        buf = C.create_string_buffer(512)
        rc = _lib.lib().bmq_route_key(self.h, route_id, buf, len(buf), C.byref(n))
        while rc == -3:  # BMQ_E_NOSPACE: n holds the length (ids may move under concurrent apply: retry)
            buf = C.create_string_buffer(max(n.value, 2 * len(buf)))
            rc = _lib.lib().bmq_route_key(self.h, route_id, buf, len(buf), C.byref(n))
        self._check(rc)
        return buf.raw[:n.value]

    def route_keys(self, route_ids) -> List[bytes]:
        """bmq_route_keys: one device gather for many ids; a dead id gives b""."""
        ids = np.ascontiguousarray(route_ids, dtype=np.uint32)
        n = len(ids)
        off = np.zeros(n + 1, dtype=np.uint64)
        cap = max(4096, 96 * n)
        while True:
            out = np.zeros(cap, dtype=np.uint8)
            rc = _lib.lib().bmq_route_keys(self.h, _ptr(ids), n, _ptr(out), cap, _ptr(off))
            if rc == -3:
                cap = int(off[n])
                continue
            self._check(rc)
            raw = out.tobytes()
            return [raw[int(off[i]):int(off[i + 1])] for i in range(n)]

    def find(self, tenant, topic_filter) -> List[int]:
        t, f = _b(tenant), _b(topic_filter)
        n = C.c_uint32()
        cap = 1024
        while True:
            out = np.zeros(cap, dtype=np.uint32)
            self._check(_lib.lib().bmq_index_find(self.h, t, len(t), f, len(f), _ptr(out), cap, C.byref(n)))
            if n.value <= cap:
                return out[:n.value].tolist()
            cap = n.value

    # ---- match (host buffers) ---------------------------------------------------------------------------
    def match_batch(self, tenants: Sequence, topic_tenant, topics: Sequence = (), packed_topics=None):
        """-> (row_ptr[n+1], route_ids) numpy arrays; ids ascending per row."""
        tdata, toff = pack(tenants)
        pdata, poff = pack(topics) if packed_topics is None else packed_topics
        n = len(poff) - 1
        tt = np.ascontiguousarray(topic_tenant, dtype=np.uint32)
        assert tt.shape == (n,)
        row = np.zeros(n + 1, dtype=np.uint32)
        cap = max(1024, 8 * n)
        need = C.c_uint64()
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            rc = _lib.lib().bmq_match_batch(self.h, _ptr(tdata), _ptr(toff), len(toff) - 1, _ptr(tt), _ptr(pdata),
                                            _ptr(poff), n, _ptr(row), _ptr(ids), cap, C.byref(need))
            if rc == -3 and need.value > cap:
                cap = need.value
                continue
            self._check(rc)
            return row, ids[:need.value]

    # ---- asynchronous host-buffer match: up to three batches in flight (bmq_match_submit / bmq_match_wait) ------------------------
    def match_submit(self, tdata, toff, n_tenants, tt, pdata, poff, n_topics) -> int:
        """numpy arrays (ideally views of bmq_host_alloc memory, see pinned()); they must stay alive until match_wait."""
        k = C.c_int()
        self._check(_lib.lib().bmq_match_submit(self.h, _ptr(tdata), _ptr(toff), n_tenants, _ptr(tt), _ptr(pdata), _ptr(poff), n_topics,
                                                C.byref(k)))
        return k.value

    def match_wait(self, ticket: int, row: np.ndarray, ids: np.ndarray) -> int:
        """-> number of ids written (BmqError code -3 with .needed set if ids is too small)"""
        need = C.c_uint64()
        rc = _lib.lib().bmq_match_wait(self.h, ticket, _ptr(row), _ptr(ids), len(ids), C.byref(need))
        if rc:
            err = BmqError(rc, (_lib.lib().bmq_last_error(self.h) or b"").decode())
            err.needed = need.value
            raise err
        return need.value

    # ---- result formats that fit the wire (include/bmq.h: BMQ_FMT_*) ------------------------------------------------------------
    FMT_IDS, FMT_COUNTS, FMT_RANGES, FMT_GROUPED = 0, 1, 2, 3
    RANGE_SIDE = 0x80000000

    def match_submit_fmt(self, tdata, toff, n_tenants, tt, pdata, poff, n_topics, fmt: int) -> int:
        k = C.c_int()
        self._check(_lib.lib().bmq_match_submit_fmt(self.h, _ptr(tdata), _ptr(toff), n_tenants, _ptr(tt), _ptr(pdata), _ptr(poff), n_topics, fmt,
                                                    C.byref(k)))
        return k.value

    def match_wait_counts(self, ticket: int, row: np.ndarray) -> int:
        """row[n + 1] <- the id row pointers (fan-out of topic i = row[i + 1] - row[i]); -> total ids"""
        tot = C.c_uint64()
        self._check(_lib.lib().bmq_match_wait_counts(self.h, ticket, _ptr(row), C.byref(tot)))
        return tot.value

    def match_wait_ranges(self, ticket: int, range_ptr: np.ndarray, ranges: np.ndarray, side: np.ndarray, row: Optional[np.ndarray] = None):
        """ranges: uint32 [cap, 2] (begin, count); side: uint32 [cap]; -> RangesInfo (BmqError -3 carries .info with the sizes)"""
        info = _lib.RangesInfo()
        rc = _lib.lib().bmq_match_wait_ranges(self.h, ticket, _ptr(row) if row is not None else None, _ptr(range_ptr), _ptr(ranges), len(ranges),
                                              _ptr(side), len(side), C.byref(info))
        if rc:
            err = BmqError(rc, (_lib.lib().bmq_last_error(self.h) or b"").decode())
            err.info = info
            raise err
        return info

    def match_wait_grouped(self, ticket: int, out_topic: np.ndarray, out_route: np.ndarray, group_off: np.ndarray, group_rep: np.ndarray):
        """-> (total pairs, n_groups, special)"""
        ng, sp, tot = C.c_uint32(), C.c_uint32(), C.c_uint64()
        rc = _lib.lib().bmq_match_wait_grouped(self.h, ticket, _ptr(out_topic), _ptr(out_route), len(out_topic), _ptr(group_off), _ptr(group_rep),
                                               len(group_rep), C.byref(ng), C.byref(sp), C.byref(tot))
        if rc:
            err = BmqError(rc, (_lib.lib().bmq_last_error(self.h) or b"").decode())
            err.needed = tot.value
            raise err
        return tot.value, ng.value, sp.value

    @staticmethod
    def expand_ranges(range_ptr: np.ndarray, ranges: np.ndarray, side: np.ndarray, n: int) -> List[np.ndarray]:
        """What a consumer of BMQ_FMT_RANGES does: the id row of every topic (ordered where ranges overlap)."""
        out = []
        for i in range(n):
            parts, last, overlap = [], -1, False
            for b, c in ranges[range_ptr[i]:range_ptr[i + 1]]:
                b, c = int(b), int(c)
                ids = side[b:b + (c & 0x7FFFFFFF)] if c & 0x80000000 else np.arange(b, b + c, dtype=np.uint32)
                if len(ids) == 0:
                    continue
                overlap |= int(ids[0]) <= last
                last = int(ids[-1])
                parts.append(ids)
            row = np.concatenate(parts).astype(np.uint32) if parts else np.zeros(0, dtype=np.uint32)
            out.append(np.sort(row) if overlap else row)
        return out

    def batcher(self, max_batch_topics: int = 0) -> "Batcher":
        """The batching front of SURVEY.md 8f-1 over this engine (close it before the engine)."""
        return Batcher(self, max_batch_topics)

    def match_tenant(self, tenant, topics: Sequence) -> List[List[int]]:
        row, ids = self.match_batch([tenant], np.zeros(len(topics), dtype=np.uint32), topics)
        return [ids[row[i]:row[i + 1]].tolist() for i in range(len(topics))]

    def match_all(self, tenant, topics: Sequence, max_persistent_fanout: int = INT_MAX,
                  max_group_fanout: int = INT_MAX):
        """ITenantRouteMatcher.matchAll for one tenant incl. MatchedRoutes fan-out caps.
        -> (per-topic id lists, events [(type, topic idx, route id, max)])"""
        t = _b(tenant)
        pdata, poff = pack(topics)
        n = len(topics)
        row = np.zeros(n + 1, dtype=np.uint32)
        cap = max(1024, 8 * n)
        need = C.c_uint64()
        ev_cap = 4096
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            ev = np.zeros(4 * ev_cap, dtype=np.int32)
            nev = C.c_uint32()
            rc = _lib.lib().bmq_match_all(self.h, t, len(t), _ptr(pdata), _ptr(poff), n, max_persistent_fanout,
                                          max_group_fanout, _ptr(row), _ptr(ids), cap, C.byref(need), _ptr(ev), ev_cap,
                                          C.byref(nev))
            if rc == -3 and need.value > cap:
                cap = need.value
                continue
            self._check(rc)
            if nev.value > ev_cap:
                ev_cap = nev.value
                continue
            events = [tuple(int(x) for x in ev[4 * i:4 * i + 4]) for i in range(nev.value)]
            return [ids[row[i]:row[i + 1]].tolist() for i in range(n)], events

    def routes_cap(self, row_ptr, route_ids, max_persistent_fanout: int, max_group_fanout: int):
        """bmq_routes_cap: MatchedRoutes' caps over rows already matched
        -> (per-row id lists, per-row (persistent, group) counts, events [(type, row, route id, max)])"""
        row = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        ids = np.ascontiguousarray(route_ids, dtype=np.uint32)
        n = len(row) - 1
        o_row, o_ids = np.zeros(n + 1, dtype=np.uint32), np.zeros(max(1, len(ids)), dtype=np.uint32)
        cls = np.zeros(2 * max(n, 1), dtype=np.uint32)
        ev = np.zeros(4 * max(1, len(ids)), dtype=np.int32)
        nev = C.c_uint32()
        self._check(_lib.lib().bmq_routes_cap(self.h, _ptr(row), _ptr(ids), n, max_persistent_fanout, max_group_fanout, _ptr(o_row), _ptr(o_ids),
                                              _ptr(cls), _ptr(ev), len(ids), C.byref(nev)))
        return ([o_ids[o_row[i]:o_row[i + 1]].tolist() for i in range(n)], [(int(cls[2 * i]), int(cls[2 * i + 1])) for i in range(n)],
                [tuple(int(x) for x in ev[4 * i:4 * i + 4]) for i in range(nev.value)])

    # ---- match (device-resident; torch tensors are only carriers of device pointers) ----------------------
    def match_batch_device(self, d_tenants, d_tenant_off, n_tenants, d_topic_tenant, d_topics, d_topic_off, n_topics,
                           d_row_ptr, d_ids, capacity, d_total):
        """All d_* are device pointers (ints).  Asynchronous; call finish()."""
        self._check(_lib.lib().bmq_match_batch_dev(self.h, d_tenants, d_tenant_off, n_tenants, d_topic_tenant, d_topics,
                                                   d_topic_off, n_topics, d_row_ptr, d_ids, capacity, d_total))

    # ---- fan-out grouping --------------------------------------------------------------------------------
    def fanout_group(self, row_ptr, route_ids, group_cap: int = 256):
        """Segmented sort of a match CSR by DelivererKey -> (out_topic, out_route, group_off, group_rep, special)."""
        row = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        ids = np.ascontiguousarray(route_ids, dtype=np.uint32)
        n, total = len(row) - 1, int(row[-1])
        ot, orr = np.zeros(max(total, 1), dtype=np.uint32), np.zeros(max(total, 1), dtype=np.uint32)
        ng, sp = C.c_uint32(), C.c_uint32()
        while True:
            goff, grep = np.zeros(group_cap + 1, dtype=np.uint32), np.zeros(max(group_cap, 1), dtype=np.uint32)
            rc = _lib.lib().bmq_fanout_group(self.h, _ptr(row), _ptr(ids), n, _ptr(ot), _ptr(orr), total, _ptr(goff), _ptr(grep), group_cap,
                                             C.byref(ng), C.byref(sp))
            if rc == -3 and ng.value > group_cap:
                group_cap = ng.value
                continue
            self._check(rc)
            return ot[:total], orr[:total], goff[:ng.value + 1], grep[:ng.value], sp.value

    def fanout_group_device(self, d_row_ptr, d_ids, n_topics, total, d_out_topic, d_out_route, d_group_off, d_group_rep, group_cap):
        """All d_* are device pointers (ints).  -> (n_groups, special); BmqError -3 (.needed = groups) if group_cap is too small."""
        ng, sp = C.c_uint32(), C.c_uint32()
        rc = _lib.lib().bmq_fanout_group_dev(self.h, d_row_ptr, d_ids, n_topics, total, d_out_topic, d_out_route, d_group_off, d_group_rep,
                                             group_cap, C.byref(ng), C.byref(sp))
        if rc:
            err = BmqError(rc, (_lib.lib().bmq_last_error(self.h) or b"").decode())
            err.needed = ng.value
            raise err
        return ng.value, sp.value

    def finish(self) -> int:
        total = C.c_uint64()
        self._check(_lib.lib().bmq_match_finish(self.h, C.byref(total)))
        return total.value

    def sync(self):
        self._check(_lib.lib().bmq_sync(self.h))

    def set_kernel_timing(self, on: bool):
        """HIP events around k_walk / k_expand (stats().ms_walk / ms_expand); ~16 us per batch, off by default."""
        self._check(_lib.lib().bmq_set_kernel_timing(self.h, 1 if on else 0))

    def stats(self) -> _lib.Stats:
        out = _lib.Stats()
        self._check(_lib.lib().bmq_stats_get(self.h, C.byref(out)))
        return out

    # ---- retain direction (IRetainTopicIndex, RS/index/IRetainTopicIndex.java:27-35) --------------------------------
    def retain_rebuild(self, tenants: Sequence, topic_tenant, topics: Sequence = (), packed_topics=None, timestamps=None, expiry=None):
        """Load the retained-topic index; topic id = rank of (tenant, level list).  timestamps (HLC) / expiry (seconds): what
        IRetainTopicIndex.add carries; both None = the topics never expire."""
        tdata, toff = pack(tenants)
        pdata, poff = pack(topics) if packed_topics is None else packed_topics
        tt = np.ascontiguousarray(topic_tenant, dtype=np.uint32)
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, dtype=np.uint64)
        ex = None if expiry is None else np.ascontiguousarray(expiry, dtype=np.uint32)
        self._check(_lib.lib().bmq_retain_rebuild_ex(self.h, _ptr(tdata), _ptr(toff), len(toff) - 1, _ptr(tt), _ptr(pdata),
                                                     _ptr(poff), len(poff) - 1, _ptr(ts), _ptr(ex)))
        return self

    def retain_compact_begin(self):
        """bmq_retain_compact_begin: snapshot of the live retained topics -- the next generation starts from it"""
        self._check(_lib.lib().bmq_retain_compact_begin(self.h))
        return self

    def retain_compact_build(self):
        """bmq_retain_compact_build: loads the snapshot into an index of its own; no engine lock held (matching / add / remove go on)"""
        self._check(_lib.lib().bmq_retain_compact_build(self.h))
        return self

    def retain_compact_swap(self) -> Tuple[int, int]:
        """bmq_retain_compact_swap: upload, replay of what was added / removed meanwhile, swap -> (topics carried over, ops replayed)"""
        carried, replayed = C.c_uint64(), C.c_uint64()
        self._check(_lib.lib().bmq_retain_compact_swap(self.h, C.byref(carried), C.byref(replayed)))
        return int(carried.value), int(replayed.value)

    def retain_compact_abort(self):
        self._check(_lib.lib().bmq_retain_compact_abort(self.h))
        return self

    def retain_apply(self, tenant, ops: Sequence):
        """ops: (0 = add | 1 = remove, topic[, timestamp_hlc, expiry_seconds]) -- IRetainTopicIndex.add / remove"""
        t = _b(tenant)
        data, off = pack([o[1] for o in ops])
        op = np.array([o[0] for o in ops], dtype=np.uint8)
        if any(len(o) > 2 for o in ops):
            ts = np.array([o[2] if len(o) > 2 else 0 for o in ops], dtype=np.uint64)
            ex = np.array([o[3] if len(o) > 3 else 0xFFFFFFFF for o in ops], dtype=np.uint32)
            self._check(_lib.lib().bmq_retain_apply_ex(self.h, t, len(t), _ptr(data), _ptr(off), _ptr(op), _ptr(ts), _ptr(ex), len(ops)))
        else:
            self._check(_lib.lib().bmq_retain_apply(self.h, t, len(t), _ptr(data), _ptr(off), _ptr(op), len(ops)))
        return self

    def retain_apply_batch(self, tenants: Sequence, op_tenant, ops: Sequence, packed_topics=None, op_codes=None, timestamps=None, expiry=None):
        """bmq_retain_apply_batch: ops of several tenants in one call.  ops: (0 = add | 1 = remove, topic[, timestamp_hlc, expiry_seconds])
        -- or packed_topics + op_codes (+ timestamps, expiry) for large batches.  -> the topic id of every op (0xFFFFFFFF: no-op)"""
        tdata, toff = pack(tenants)
        if packed_topics is None:
            data, off = pack([o[1] for o in ops])
            op = np.array([o[0] for o in ops], dtype=np.uint8)
            if any(len(o) > 2 for o in ops):
                timestamps = np.array([o[2] if len(o) > 2 else 0 for o in ops], dtype=np.uint64)
                expiry = np.array([o[3] if len(o) > 3 else 0xFFFFFFFF for o in ops], dtype=np.uint32)
        else:
            data, off = packed_topics
            op = np.ascontiguousarray(op_codes, dtype=np.uint8)
        n = len(off) - 1
        ot = None if op_tenant is None else np.ascontiguousarray(op_tenant, dtype=np.uint32)
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, dtype=np.uint64)
        ex = None if expiry is None else np.ascontiguousarray(expiry, dtype=np.uint32)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self._check(_lib.lib().bmq_retain_apply_batch(self.h, _ptr(tdata), _ptr(toff), len(tenants), _ptr(ot), _ptr(data), _ptr(off), _ptr(op),
                                                      _ptr(ts), _ptr(ex), n, _ptr(out)))
        return out[:n]

    def retain_compact(self):
        self._check(_lib.lib().bmq_retain_compact(self.h))
        return self

    def retain_info(self) -> "_lib.RetainInfo":
        st = _lib.RetainInfo()
        self._check(_lib.lib().bmq_retain_info_get(self.h, C.byref(st)))
        return st

    def retain_live_ids(self, tenant=None) -> List[int]:
        """ids of the retained topics, ascending (tenant None: of every tenant) -- IRetainTopicIndex.findAll()"""
        t = None if tenant is None else _b(tenant)
        cap = 1024
        n = C.c_uint32()
        while True:
            out = np.zeros(cap, dtype=np.uint32)
            rc = _lib.lib().bmq_retain_live_ids(self.h, t, len(t) if t is not None else 0, _ptr(out), cap, C.byref(n))
            if rc == -3:
                cap = n.value
                continue
            self._check(rc)
            return out[:n.value].tolist()

    def retain_topics(self, topic_ids) -> List[Tuple[str, str]]:
        """many ids -> [(tenant, topic)] (("", "") for an unknown id)"""
        ids = np.ascontiguousarray(topic_ids, dtype=np.uint32)
        n = len(ids)
        off, tl = np.zeros(n + 1, dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint32)
        cap = max(4096, 64 * n)
        while True:
            out = np.zeros(cap, dtype=np.uint8)
            rc = _lib.lib().bmq_retain_topics(self.h, _ptr(ids), n, _ptr(out), cap, _ptr(off), _ptr(tl))
            if rc == -3:
                cap = int(off[n]) + 16
                continue
            self._check(rc)
            raw = out.tobytes()
            return [(raw[int(off[i]):int(off[i]) + int(tl[i])].decode(), raw[int(off[i]) + int(tl[i]):int(off[i + 1])].decode()) for i in range(n)]

    def retain_topic_info(self, topic_id: int) -> Tuple[int, int, int]:
        """-> (timestamp_hlc, expiry_seconds, expire_at_ms)"""
        ts, ex, at = C.c_uint64(), C.c_uint32(), C.c_uint64()
        self._check(_lib.lib().bmq_retain_topic_info(self.h, topic_id, C.byref(ts), C.byref(ex), C.byref(at)))
        return ts.value, ex.value, at.value

    def retain_find_all(self) -> Tuple[int, int]:
        """IRetainTopicIndex.findAll(): (number of retained topics, retain epoch); retain_live_ids lists the ids"""
        n, ep = C.c_uint64(), C.c_uint64()
        self._check(_lib.lib().bmq_retain_find_all(self.h, C.byref(n), C.byref(ep)))
        return n.value, ep.value

    def retain_expired(self, tenant, now_ms: int, override_expiry_seconds: int = -1) -> List[int]:
        """ids whose message has expired at now_ms (tenant None: all tenants) -- the GC scan of RetainStoreCoProc."""
        t = None if tenant is None else _b(tenant)
        cap = 1024
        n = C.c_uint32()
        while True:
            out = np.zeros(cap, dtype=np.uint32)
            rc = _lib.lib().bmq_retain_expired(self.h, t, len(t) if t is not None else 0, now_ms, override_expiry_seconds, _ptr(out), cap,
                                               C.byref(n))
            if rc == -3:
                cap = n.value
                continue
            self._check(rc)
            return out[:n.value].tolist()

    def retain_topic(self, topic_id: int) -> Tuple[str, str]:
        buf = C.create_string_buffer(140000)
        n, tl = C.c_uint32(), C.c_uint32()
        self._check(_lib.lib().bmq_retain_topic(self.h, topic_id, buf, len(buf), C.byref(n), C.byref(tl)))
        raw = buf.raw[:n.value]
        return raw[:tl.value].decode(), raw[tl.value:].decode()

    def retain_match_batch(self, tenants: Sequence, filter_tenant, filters: Sequence = (), packed_filters=None):
        """Batch of IRetainTopicIndex.match(tenant, topicFilter) -> (row_ptr[n+1], topic ids ascending per row)."""
        tdata, toff = pack(tenants)
        pdata, poff = pack(filters) if packed_filters is None else packed_filters
        n = len(poff) - 1
        ft = np.ascontiguousarray(filter_tenant, dtype=np.uint32)
        row = np.zeros(n + 1, dtype=np.uint32)
        cap = max(1024, 16 * n)
        need = C.c_uint64()
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            rc = _lib.lib().bmq_retain_match_batch(self.h, _ptr(tdata), _ptr(toff), len(toff) - 1, _ptr(ft), _ptr(pdata),
                                                   _ptr(poff), n, _ptr(row), _ptr(ids), cap, C.byref(need))
            if rc == -3 and need.value > cap:
                cap = need.value
                continue
            self._check(rc)
            return row, ids[:need.value]

    def retain_match_limited(self, tenants: Sequence, filter_tenant, filters: Sequence, limits, now_ms: int = 0, packed_filters=None):
        """RetainStoreCoProc.match(limit, now) for a batch -> (row_ptr[n+1], kept topic ids, match count per filter):
        row i holds the limits[i] smallest matching topic ids that have not expired at now_ms."""
        tdata, toff = pack(tenants)
        pdata, poff = pack(filters) if packed_filters is None else packed_filters
        n = len(poff) - 1
        ft = np.ascontiguousarray(filter_tenant, dtype=np.uint32)
        lim = np.ascontiguousarray(limits, dtype=np.uint32)
        assert len(lim) == n and len(ft) == n
        row = np.zeros(n + 1, dtype=np.uint32)
        counts = np.zeros(n, dtype=np.uint32)
        cap = int(min(int(lim.astype(np.uint64).sum()), 1 << 28)) + 1
        need = C.c_uint64()
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            rc = _lib.lib().bmq_retain_match_limited(self.h, _ptr(tdata), _ptr(toff), len(toff) - 1, _ptr(ft), _ptr(pdata), _ptr(poff),
                                                     n, _ptr(lim), now_ms, _ptr(row), _ptr(ids), cap, C.byref(need), _ptr(counts))
            if rc == -3 and need.value > cap:
                cap = need.value
                continue
            self._check(rc)
            return row, ids[:need.value], counts

    def retain_match(self, tenant, topic_filter) -> List[int]:
        row, ids = self.retain_match_batch([tenant], [0], [topic_filter])
        return ids.tolist()

    def retain_match_batch_device(self, d_tenants, d_tenant_off, n_tenants, d_filter_tenant, d_filters, d_filter_off,
                                  n_filters, d_row_ptr, d_ids, capacity, d_total):
        self._check(_lib.lib().bmq_retain_match_batch_dev(self.h, d_tenants, d_tenant_off, n_tenants, d_filter_tenant,
                                                          d_filters, d_filter_off, n_filters, d_row_ptr, d_ids, capacity,
                                                          d_total))

    @property
    def stream(self) -> int:
        return _lib.lib().bmq_stream(self.h) or 0


class Batcher:
    """bmq_batcher_*: collects the single-topic matchAll calls of many threads (TenantRouteCache.java:180-193) into one
    GPU batch.  match_all blocks; call it from as many threads as there are callers (ctypes drops the GIL)."""

    def __init__(self, engine: Engine, max_batch_topics: int = 0):
        cfg = _lib.BatcherConfig()
        cfg.struct_size = C.sizeof(_lib.BatcherConfig)
        cfg.max_batch_topics = max_batch_topics
        h = C.c_void_p()
        rc = _lib.lib().bmq_batcher_create(engine.h, C.byref(cfg), C.byref(h))
        if rc:
            raise BmqError(rc, "bmq_batcher_create failed")
        self.h = h
        self.engine = engine  # keeps the engine alive

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().bmq_batcher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def match_all(self, tenant, topics: Sequence) -> Tuple[List[List[int]], int]:
        """-> (per-topic ascending route id lists, epoch the ids belong to)"""
        t = _b(tenant)
        pdata, poff = pack(topics)
        n = len(topics)
        row = np.zeros(n + 1, dtype=np.uint32)
        cap = max(256, 16 * n)
        need, epoch = C.c_uint64(), C.c_uint64()
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            rc = _lib.lib().bmq_batcher_match_all(self.h, t, len(t), _ptr(pdata), _ptr(poff), n, _ptr(row), _ptr(ids), cap,
                                                  C.byref(need), C.byref(epoch))
            if rc == -3 and need.value > cap:
                cap = int(need.value)
                continue
            if rc:
                raise BmqError(rc, (_lib.lib().bmq_last_error(self.engine.h) or b"").decode())
            return [ids[row[i]:row[i + 1]].tolist() for i in range(n)], int(epoch.value)

    def stats(self) -> "_lib.BatcherStats":
        st = _lib.BatcherStats()
        rc = _lib.lib().bmq_batcher_stats_get(self.h, C.byref(st))
        if rc:
            raise BmqError(rc, "bmq_batcher_stats_get")
        return st

    CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint64)

    def submit(self, tenant, topic, on_done):
        """bmq_batcher_submit: returns at once; on_done(status, ids list, epoch) runs on the batcher's dispatcher thread."""
        t, p = _b(tenant), _b(topic)
        if not hasattr(self, "_live"):
            self._live, self._seq = {}, 0
        self._seq += 1
        key = self._seq

        def tramp(_user, status, ids, n, epoch):
            try:
                on_done(status, [ids[i] for i in range(n)], epoch)
            finally:
                self._live.pop(key, None)

        cb = Batcher.CALLBACK(tramp)
        self._live[key] = cb  # keeps the trampoline alive until it has run
        rc = _lib.lib().bmq_batcher_submit(self.h, t, len(t), p, len(p), C.cast(cb, C.c_void_p), None)
        if rc:
            self._live.pop(key, None)
            raise BmqError(rc, "bmq_batcher_submit")

    def drive_singletons(self, tenants: Sequence, topic_tenant: np.ndarray, topics_packed: Tuple[np.ndarray, np.ndarray],
                         n_threads: int, asynchronous: bool = False):
        """n_threads native threads issue one single-topic call per topic (the production call pattern), blocking
        (bmq_batcher_match_all) or asynchronous (bmq_batcher_submit + callback).
        -> (ids per topic, row hash per topic, seconds)"""
        tdata, toff = pack(tenants)
        pdata, poff = topics_packed
        n = len(poff) - 1
        cnt = np.zeros(n, dtype=np.uint32)
        hsh = np.zeros(n, dtype=np.uint64)
        sec = C.c_double()
        fn = C.cast(_lib.lib().bmq_batcher_submit if asynchronous else _lib.lib().bmq_batcher_match_all, C.c_void_p)
        tt = np.ascontiguousarray(topic_tenant, dtype=np.uint32)
        drive = _lib.gen().bmqgen_drive_async if asynchronous else _lib.gen().bmqgen_drive_singletons
        rc = drive(fn, self.h, _ptr(tdata), _ptr(toff), len(tenants), _ptr(tt), _ptr(pdata), _ptr(poff),
                                                n, n_threads, _ptr(cnt), _ptr(hsh), C.byref(sec))
        if rc:
            raise BmqError(rc, "bmqgen_drive_singletons")
        return cnt, hsh, sec.value


class RouteCache:
    """bmq_route_cache_*: ISubscriptionCache (DW/cache/ISubscriptionCache.java:30-40) on the engine's side of the boundary -- topic ->
    matched routes per tenant, loads through the batching front, TopicIndex-style invalidation by route mutations."""

    def __init__(self, batcher: Batcher, max_routes_per_tenant: int = 0, expiry_ms: int = 0, mutation_log_entries: int = 0, shards_per_tenant: int = 0,
                 max_persistent_fanout: int = 0, max_group_fanout: int = 0, tenant_idle_ms: int = 0, direct_batch_topics: int = 0):
        """max_persistent_fanout / max_group_fanout: the default caps of tenants without set_caps (0 = the reference's defaults,
        Setting.java:60-61: INT_MAX / 100)"""
        cfg = _lib.RouteCacheConfig()
        cfg.struct_size = C.sizeof(_lib.RouteCacheConfig)
        cfg.max_routes_per_tenant = max_routes_per_tenant
        cfg.expiry_ms = expiry_ms
        cfg.mutation_log_entries = mutation_log_entries
        cfg.shards_per_tenant = shards_per_tenant
        cfg.default_max_persistent_fanout = max_persistent_fanout
        cfg.default_max_group_fanout = max_group_fanout
        cfg.tenant_idle_ms = tenant_idle_ms
        cfg.direct_batch_topics = direct_batch_topics
        h = C.c_void_p()
        rc = _lib.lib().bmq_route_cache_create(batcher.engine.h, batcher.h, C.byref(cfg), C.byref(h))
        if rc:
            raise BmqError(rc, "bmq_route_cache_create failed")
        self.h = h
        self.batcher = batcher  # keeps batcher and engine alive

    def close(self):
        if getattr(self, "h", None):
            _lib.lib().bmq_route_cache_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise BmqError(rc, what + ": " + (_lib.lib().bmq_last_error(self.batcher.engine.h) or b"").decode())

    def get(self, tenant, topic, now_ms: int = 0) -> Tuple[List[int], int]:
        """ISubscriptionCache.get -> (ascending route ids, engine epoch they were matched at)"""
        t, p = _b(tenant), _b(topic)
        cap = 64
        n, ep = C.c_uint32(), C.c_uint64()
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            rc = _lib.lib().bmq_route_cache_get(self.h, t, len(t), p, len(p), now_ms, _ptr(ids), cap, C.byref(n), C.byref(ep))
            if rc == -3:
                cap = n.value + 16
                continue
            self._check(rc, "bmq_route_cache_get")
            return ids[:n.value].tolist(), int(ep.value)

    def get_batch(self, tenants: Sequence, topic_tenant, topics: Sequence, now_ms: int = 0):
        """bmq_route_cache_get_batch: a whole BatchDistRequest -> (row_ptr, route ids, hit flags); misses share ONE launch"""
        tdata, toff = pack(tenants)
        pdata, poff = pack(topics)
        n = len(topics)
        tt = np.ascontiguousarray(topic_tenant, dtype=np.uint32)
        row, hit = np.zeros(n + 1, dtype=np.uint32), np.zeros(max(n, 1), dtype=np.uint8)
        cap, need = max(256, 8 * n), C.c_uint64()
        first_hit = None
        while True:
            ids = np.zeros(cap, dtype=np.uint32)
            rc = _lib.lib().bmq_route_cache_get_batch(self.h, _ptr(tdata), _ptr(toff), len(tenants), _ptr(tt), _ptr(pdata), _ptr(poff), n, now_ms,
                                                      _ptr(row), _ptr(ids), cap, C.byref(need), _ptr(hit))
            if first_hit is None:  # a call that ran out of room has loaded (and cached) its misses already: the retry would call them hits
                first_hit = hit[:n].astype(bool)
            if rc == -3 and need.value > cap:
                cap = int(need.value)
                continue
            self._check(rc, "bmq_route_cache_get_batch")
            return row, ids[:need.value], first_hit

    CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint64)

    def get_async(self, tenant, topic, on_done, now_ms: int = 0):
        """bmq_route_cache_get_async: on_done(status, ids list, epoch) runs inline on a hit, on the batcher's dispatcher thread after a miss"""
        t, p = _b(tenant), _b(topic)
        if not hasattr(self, "_live"):
            self._live, self._seq = {}, 0
        self._seq += 1
        key = self._seq

        def tramp(_user, status, ids, n, epoch):
            try:
                on_done(status, [ids[i] for i in range(n)], epoch)
            finally:
                self._live.pop(key, None)

        cb = RouteCache.CALLBACK(tramp)
        self._live[key] = cb  # keeps the trampoline alive until it has run
        rc = _lib.lib().bmq_route_cache_get_async(self.h, t, len(t), p, len(p), now_ms, C.cast(cb, C.c_void_p), None)
        if rc:
            self._live.pop(key, None)
            raise BmqError(rc, "bmq_route_cache_get_async")

    def is_cached(self, tenant, topic_filter) -> bool:
        t, f = _b(tenant), _b(topic_filter)
        rc = _lib.lib().bmq_route_cache_is_cached(self.h, t, len(t), f, len(f))
        if rc < 0:
            raise BmqError(rc, "bmq_route_cache_is_cached")
        return rc == 1

    def apply(self, ops: Sequence[Tuple[int, bytes]]):
        """ISubscriptionCache.refresh: [(0 = put | 1 = delete, route key)] -> engine, then invalidation"""
        data, off = pack([k for _, k in ops])
        op = np.array([o for o, _ in ops], dtype=np.uint8)
        self._check(_lib.lib().bmq_route_cache_apply(self.h, _ptr(data), _ptr(off), _ptr(op), len(ops)), "bmq_route_cache_apply")

    def rebuild(self, keys: Iterable[bytes]):
        data, off = pack(sorted(keys))
        self._check(_lib.lib().bmq_route_cache_rebuild(self.h, _ptr(data), _ptr(off), len(off) - 1), "bmq_route_cache_rebuild")

    def reset(self):
        self._check(_lib.lib().bmq_route_cache_reset(self.h), "bmq_route_cache_reset")

    def expire(self, now_ms: int) -> int:
        """drop every entry not accessed for expiry_ms -> number dropped"""
        n = C.c_uint64()
        self._check(_lib.lib().bmq_route_cache_expire(self.h, now_ms, C.byref(n)), "bmq_route_cache_expire")
        return int(n.value)

    def stats(self) -> "_lib.RouteCacheStats":
        st = _lib.RouteCacheStats()
        self._check(_lib.lib().bmq_route_cache_stats_get(self.h, C.byref(st)), "bmq_route_cache_stats_get")
        return st

    def tenant_stats(self, tenant) -> Optional["_lib.RouteCacheTenantStats"]:
        """the tenant's meters (TenantRouteCache.java:141-147), or None when the tenant has no cache"""
        t = _b(tenant)
        st = _lib.RouteCacheTenantStats()
        rc = _lib.lib().bmq_route_cache_tenant_stats_get(self.h, t, len(t), C.byref(st))
        if rc == -7:
            return None
        self._check(rc, "bmq_route_cache_tenant_stats_get")
        return st

    def set_caps(self, tenant, max_persistent_fanout: int, max_group_fanout: int):
        """the tenant's MaxPersistentFanout / MaxGroupFanout settings"""
        t = _b(tenant)
        self._check(_lib.lib().bmq_route_cache_set_caps(self.h, t, len(t), max_persistent_fanout, max_group_fanout), "bmq_route_cache_set_caps")

    EVENT_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32, C.c_int32, C.c_uint32, C.c_int32)

    def collect_events(self) -> list:
        """installs an event sink (IEventCollector) that appends (tenant, topic, type, route id, max count) to the returned list;
        type 0 = PersistentFanoutThrottled, 1 = GroupFanoutThrottled"""
        events: list = []

        def sink(_user, tenant, tl, topic, pl, typ, rid, mx):
            events.append((bytes(tenant[:tl]), bytes(topic[:pl]), int(typ), int(rid), int(mx)))

        self._event_cb = RouteCache.EVENT_CB(sink)  # kept alive with the cache
        self._check(_lib.lib().bmq_route_cache_set_event_sink(self.h, C.cast(self._event_cb, C.c_void_p), None), "bmq_route_cache_set_event_sink")
        return events

    def drive(self, tenants: Sequence, topic_tenant: np.ndarray, topics_packed: Tuple[np.ndarray, np.ndarray], n_threads: int, passes: int = 2,
              asynchronous: bool = False):
        """n_threads native threads call bmq_route_cache_get (or _get_async) once per topic, `passes` times over the batch.
        -> (ids per topic, row hash per topic, seconds per pass)"""
        tdata, toff = pack(tenants)
        pdata, poff = topics_packed
        n = len(poff) - 1
        cnt, hsh = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint64)
        sec = (C.c_double * passes)()
        tt = np.ascontiguousarray(topic_tenant, dtype=np.uint32)
        fn, drv = ((_lib.lib().bmq_route_cache_get_async, _lib.gen().bmqgen_drive_cache_async) if asynchronous else
                   (_lib.lib().bmq_route_cache_get, _lib.gen().bmqgen_drive_cache))
        rc = drv(C.cast(fn, C.c_void_p), self.h, _ptr(tdata), _ptr(toff), len(tenants), _ptr(tt),
                                           _ptr(pdata), _ptr(poff), n, n_threads, passes, _ptr(cnt), _ptr(hsh), sec)
        if rc:
            raise BmqError(rc, "bmqgen_drive_cache")
        return cnt, hsh, list(sec)
