"""ctypes binding of the CPU oracle (oracle/bmq_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (bifromq_amd/) never does.  See bmq_oracle.cpp for the reference
file:line each function follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbmq_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bmq_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p, u32p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_int)
        vp = C.c_void_p
        sig = {
            "orc_java_hash": (C.c_int32, [C.c_char_p, C.c_uint32]),
            "orc_route_key": (C.c_uint32, [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint8, C.c_char_p,
                                           C.c_uint32, C.c_char_p, C.c_uint32]),
            "orc_tenant_route_start_key": (C.c_uint32, [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p,
                                                        C.c_uint32]),
            "orc_parse_route_key": (C.c_int, [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, u32p]),
            "orc_semantic_match": (C.c_int, [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32]),
            "orc_buf_new": (vp, []), "orc_buf_free": (None, [vp]), "orc_buf_count": (C.c_uint32, [vp]),
            "orc_buf_bytes": (u8p, [vp]), "orc_buf_off": (u32p, [vp]),
            "orc_test_expand": (None, [C.c_char_p, C.c_uint32, vp]),
            "orc_trie_new": (vp, [C.c_int]), "orc_trie_free": (None, [vp]),
            "orc_trie_add": (None, [vp, C.c_char_p, C.c_uint32, C.c_int]),
            "orc_iter_new": (vp, [vp]), "orc_iter_free": (None, [vp]),
            "orc_iter_seek": (None, [vp, C.c_char_p, C.c_uint32, C.c_int]),
            "orc_iter_next": (None, [vp]), "orc_iter_valid": (C.c_int, [vp]),
            "orc_iter_key": (C.c_uint32, [vp, C.c_char_p, C.c_uint32]),
            "orc_iter_values": (C.c_uint32, [vp, i32p, C.c_uint32]),
            "orc_iter_value_topics": (None, [vp, vp]),
            "orc_kv_new": (vp, [vp, vp, C.c_uint32]), "orc_kv_free": (None, [vp]),
            "orc_kv_size": (C.c_uint32, [vp]), "orc_kv_key": (C.c_uint32, [vp, C.c_uint32, C.c_char_p, C.c_uint32]),
            "orc_result_new": (vp, []), "orc_result_free": (None, [vp]),
            "orc_result_rowptr": (u32p, [vp]), "orc_result_routes": (u32p, [vp]),
            "orc_result_nroutes": (C.c_uint32, [vp]), "orc_result_nevents": (C.c_uint32, [vp]),
            "orc_result_event": (None, [vp, C.c_uint32, i32p]),
            "orc_result_seeks": (C.c_uint64, [vp]), "orc_result_nexts": (C.c_uint64, [vp]),
            "orc_result_livelocks": (C.c_uint64, [vp]),
            "orc_match_all": (None, [vp, C.c_char_p, C.c_uint32, vp, vp, C.c_uint32, C.c_int, C.c_int, vp]),
            "orc_match_singletons": (C.c_double, [vp, vp, vp, vp, vp, vp, C.c_uint32, C.c_int, vp]),
            "orc_match_bruteforce": (None, [vp, C.c_char_p, C.c_uint32, vp, vp, C.c_uint32, vp]),
            "orc_match_semantic_batch": (C.c_double, [vp, vp, vp, vp, vp, vp, C.c_uint32, C.c_int, vp]),
            "orc_count_visits": (None, [vp, vp, vp, vp, vp, vp, C.c_uint32, vp]),
            "orc_ltrie_new": (vp, [C.c_int]), "orc_ltrie_free": (None, [vp]),
            "orc_ltrie_add": (None, [vp, C.c_char_p, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32, C.c_int]),
            "orc_ltrie_remove": (None, [vp, C.c_char_p, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32, C.c_int]),
            "orc_ltrie_match": (C.c_uint32, [vp, C.c_char_p, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32, C.c_int,
                                             i32p, C.c_uint32]),
            "orc_ltrie_visits": (C.c_uint64, [vp]),
            "orc_ltrie_match_batch": (C.c_double, [vp, vp, vp, vp, vp, vp, C.c_uint32, C.c_int, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


INT_MAX = 2**31 - 1


def _b(s) -> bytes:
    return s if isinstance(s, (bytes, bytearray)) else s.encode("utf-8")


def pack(strings: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """list of str/bytes -> (uint8 bytes, uint32 offsets[n+1])"""
    bs = [_b(s) for s in strings]
    off = np.zeros(len(bs) + 1, dtype=np.uint32)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64).astype(np.uint32)
    data = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)  # keep a valid pointer
    return data, off


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def java_hash(s) -> int:
    b = _b(s)
    return lib().orc_java_hash(b, len(b))


# ---- route-key codec (SCHEMA/KVSchemaUtil.java:91-130) ---------------------------------------
FLAG_NORMAL, FLAG_UNORDERED, FLAG_ORDERED = 1, 2, 3


def receiver_url(sub_broker_id: int, receiver_id: str, deliverer_key: str) -> str:
    """KVSchemaUtil.toReceiverUrl (SCHEMA/KVSchemaUtil.java:56-58)"""
    return f"{sub_broker_id}\0{receiver_id}\0{deliverer_key}"


def route_key(tenant, topic_filter, flag: int, receiver) -> bytes:
    """topic_filter: MQTT filter without any $share prefix; receiver: receiverUrl (flag 1) or group (flag 2/3)."""
    t, f, r = _b(tenant), _b(topic_filter), _b(receiver)
    out = C.create_string_buffer(len(t) + len(f) + len(r) + 32)
    n = lib().orc_route_key(t, len(t), f, len(f), flag, r, len(r), out, len(out))
    return out.raw[:n]


def route_key_from_mqtt(tenant, mqtt_topic_filter: str, receiver_url_: str = "") -> bytes:
    """TopicUtil.from + toNormalRouteKey/toGroupRouteKey (UTIL/TopicUtil.java:252-272)."""
    if mqtt_topic_filter.startswith("$share/") or mqtt_topic_filter.startswith("$oshare/"):
        ordered = mqtt_topic_filter.startswith("$oshare/")
        rest = mqtt_topic_filter[len("$oshare/" if ordered else "$share/"):]
        i = rest.index("/")
        return route_key(tenant, rest[i + 1:], FLAG_ORDERED if ordered else FLAG_UNORDERED, rest[:i])
    return route_key(tenant, mqtt_topic_filter, FLAG_NORMAL, receiver_url_)


def tenant_route_start_key(tenant, topic_filter) -> bytes:
    t, f = _b(tenant), _b(topic_filter)
    out = C.create_string_buffer(len(t) + len(f) + 16)
    n = lib().orc_tenant_route_start_key(t, len(t), f, len(f), out, len(out))
    return out.raw[:n]


def parse_route_key(key: bytes):
    """-> (flag, tenant, mqttTopicFilter, receiver) or None"""
    out = C.create_string_buffer(len(key) + 64)
    lens = (C.c_uint32 * 3)()
    flag = lib().orc_parse_route_key(key, len(key), out, len(out), lens)
    if flag < 0:
        return None
    a, b, c = lens[0], lens[1], lens[2]
    raw = out.raw
    return flag, raw[:a].decode(), raw[a:a + b].decode(), raw[a + b:a + b + c].decode()


# ---- semantic matcher / expansion -----------------------------------------------------------
def semantic_match(topic, topic_filter) -> bool:
    t, f = _b(topic), _b(topic_filter)
    return bool(lib().orc_semantic_match(t, len(t), f, len(f)))


class _Buf:
    def __init__(self):
        self.h = lib().orc_buf_new()

    def strings(self) -> List[bytes]:
        L = lib()
        n = L.orc_buf_count(self.h)
        off = np.ctypeslib.as_array(L.orc_buf_off(self.h), shape=(n + 1,)).copy()
        total = int(off[-1])
        if total == 0:
            return [b""] * n
        data = bytes(np.ctypeslib.as_array(L.orc_buf_bytes(self.h), shape=(total,)))
        return [data[off[i]:off[i + 1]] for i in range(n)]

    def __del__(self):
        if lib is not None and self.h:
            lib().orc_buf_free(self.h)
            self.h = None


def test_expand(topic) -> List[str]:
    """TRIET/TestUtil.java:70-107 -> '/'-joined filters in order"""
    b = _Buf()
    t = _b(topic)
    lib().orc_test_expand(t, len(t), b.h)
    return [s.replace(b"\0", b"/").decode() for s in b.strings()]


test_expand.__test__ = False  # not a pytest test


class TopicTrie:
    """TRIE/TopicTrieNode.java builder"""

    def __init__(self, is_global: bool = False):
        self.h = lib().orc_trie_new(1 if is_global else 0)

    def add_topic(self, topic, value: int):
        t = _b(topic)
        lib().orc_trie_add(self.h, t, len(t), value)
        return self

    def __del__(self):
        if self.h:
            lib().orc_trie_free(self.h)
            self.h = None


class TopicFilterIterator:
    """TRIE/TopicFilterIterator.java (seek/next/key/value)"""

    def __init__(self, trie: TopicTrie):
        self._trie = trie
        self.h = lib().orc_iter_new(trie.h)

    def seek(self, topic_filter: Optional[str]):
        """topic_filter None => seek(emptyList)"""
        if topic_filter is None:
            lib().orc_iter_seek(self.h, b"", 0, 1)
        else:
            f = _b(topic_filter)
            lib().orc_iter_seek(self.h, f, len(f), 0)

    def next(self):
        lib().orc_iter_next(self.h)

    def is_valid(self) -> bool:
        return bool(lib().orc_iter_valid(self.h))

    def key(self) -> str:
        out = C.create_string_buffer(70000)
        n = lib().orc_iter_key(self.h, out, len(out))
        return out.raw[:n].decode()

    def values(self) -> List[int]:
        out = (C.c_int * 4096)()
        n = lib().orc_iter_values(self.h, out, 4096)
        return list(out[:n])

    def value_topics(self) -> List[str]:
        b = _Buf()
        lib().orc_iter_value_topics(self.h, b.h)
        return [s.decode() for s in b.strings()]

    def all_keys(self) -> List[str]:
        out = []
        while self.is_valid():
            out.append(self.key())
            self.next()
        return out

    def __del__(self):
        if self.h:
            lib().orc_iter_free(self.h)
            self.h = None


# ---- KV + matchAll ----------------------------------------------------------------------------
class MatchResult:
    def __init__(self, n_topics: int):
        self.h = lib().orc_result_new()
        self.n = n_topics

    @property
    def row_ptr(self) -> np.ndarray:
        return np.ctypeslib.as_array(lib().orc_result_rowptr(self.h), shape=(self.n + 1,)).copy()

    @property
    def routes(self) -> np.ndarray:
        n = lib().orc_result_nroutes(self.h)
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(lib().orc_result_routes(self.h), shape=(n,)).copy()

    def per_topic(self) -> List[List[int]]:
        rp, r = self.row_ptr, self.routes
        return [r[rp[i]:rp[i + 1]].tolist() for i in range(self.n)]

    @property
    def events(self) -> List[Tuple[int, int, int, int]]:
        """(type 0=PersistentFanoutThrottled 1=GroupFanoutThrottled, topic idx, route rank, maxCount)"""
        out = []
        buf = (C.c_int * 4)()
        for i in range(lib().orc_result_nevents(self.h)):
            lib().orc_result_event(self.h, i, buf)
            out.append(tuple(buf))
        return out

    @property
    def seek_count(self) -> int:
        return lib().orc_result_seeks(self.h)

    @property
    def next_count(self) -> int:
        return lib().orc_result_nexts(self.h)

    @property
    def livelocks(self) -> int:
        """times the reference's probe-then-seek loop would have spun forever (see bmq_oracle.cpp)"""
        return lib().orc_result_livelocks(self.h)

    def __del__(self):
        if self.h:
            lib().orc_result_free(self.h)
            self.h = None


class KV:
    """Sorted route-key array: stand-in for the KV range (TreeMapKVReader of the reference's tests).
    Route id == rank of the key in unsigned-byte order."""

    def __init__(self, keys: Iterable[bytes] = (), packed: Optional[Tuple[np.ndarray, np.ndarray]] = None):
        if packed is None:
            data, off = pack(list(keys))
        else:
            data, off = packed
        self.h = lib().orc_kv_new(_ptr(data), _ptr(off), len(off) - 1)

    def __len__(self):
        return lib().orc_kv_size(self.h)

    def key(self, rank: int) -> bytes:
        out = C.create_string_buffer(70000)
        n = lib().orc_kv_key(self.h, rank, out, len(out))
        return out.raw[:n]

    def match_all(self, tenant, topics: Sequence, max_persistent_fanout: int = INT_MAX,
                  max_group_fanout: int = INT_MAX) -> MatchResult:
        """One TenantRouteMatcher.matchAll call (DW/cache/TenantRouteMatcher.java:67-161)."""
        t = _b(tenant)
        data, off = pack(topics)
        res = MatchResult(len(topics))
        lib().orc_match_all(self.h, t, len(t), _ptr(data), _ptr(off), len(topics), max_persistent_fanout,
                            max_group_fanout, res.h)
        return res

    def match_singletons(self, tenants: Sequence, topic_tenant: np.ndarray, topics_packed, threads: int = 1):
        """Production call pattern: matchAll(singleton(topic)) per topic. -> (MatchResult, seconds)"""
        tdata, toff = pack(tenants)
        data, off = topics_packed
        tt = np.ascontiguousarray(topic_tenant, dtype=np.uint32)
        n = len(off) - 1
        res = MatchResult(n)
        sec = lib().orc_match_singletons(self.h, _ptr(tdata), _ptr(toff), _ptr(tt), _ptr(data), _ptr(off), n,
                                         threads, res.h)
        return res, sec

    def match_bruteforce(self, tenant, topics: Sequence) -> MatchResult:
        """Semantic oracle (A): every key of the tenant tested against every topic."""
        t = _b(tenant)
        data, off = pack(topics)
        res = MatchResult(len(topics))
        lib().orc_match_bruteforce(self.h, t, len(t), _ptr(data), _ptr(off), len(topics), res.h)
        return res

    def match_semantic_batch(self, tenants: Sequence, topic_tenant: np.ndarray, topics_packed, threads: int = 1):
        """Semantic oracle (A) for a whole multi-tenant batch on `threads` host threads: per topic every key of ITS tenant is tested
        (a tenant's keys are one run of the sorted KV).  -> (MatchResult, seconds)"""
        tdata, toff = pack(tenants)
        data, off = topics_packed
        tt = np.ascontiguousarray(topic_tenant, dtype=np.uint32)
        n = len(off) - 1
        res = MatchResult(n)
        sec = lib().orc_match_semantic_batch(self.h, _ptr(tdata), _ptr(toff), _ptr(tt), _ptr(data), _ptr(off), n, threads, res.h)
        return res, sec

    def count_visits(self, tenants: Sequence, topic_tenant: np.ndarray, topics_packed) -> np.ndarray:
        tdata, toff = pack(tenants)
        data, off = topics_packed
        tt = np.ascontiguousarray(topic_tenant, dtype=np.uint32)
        n = len(off) - 1
        out = np.zeros(n, dtype=np.uint32)
        lib().orc_count_visits(self.h, _ptr(tdata), _ptr(toff), _ptr(tt), _ptr(data), _ptr(off), n, _ptr(out))
        return out

    def __del__(self):
        if self.h:
            lib().orc_kv_free(self.h)
            self.h = None


# ---- retain direction ---------------------------------------------------------------------------
class LevelTrie:
    """TopicLevelTrie + selector. sys_level=0: DW/TopicIndex.java (no tenant level);
    sys_level=1: RS/index/RetainTopicIndex.java (level 0 = tenant)."""

    def __init__(self, sys_level: int):
        self.sys_level = sys_level
        self.h = lib().orc_ltrie_new(sys_level)

    def add(self, tenant, topic, value: int):
        t, p = _b(tenant or ""), _b(topic)
        lib().orc_ltrie_add(self.h, t, len(t), self.sys_level, p, len(p), value)

    def remove(self, tenant, topic, value: int):
        t, p = _b(tenant or ""), _b(topic)
        lib().orc_ltrie_remove(self.h, t, len(t), self.sys_level, p, len(p), value)

    def match(self, tenant, topic_filter) -> List[int]:
        t, f = _b(tenant or ""), _b(topic_filter)
        cap = 1 << 16
        out = (C.c_int * cap)()
        n = lib().orc_ltrie_match(self.h, t, len(t), self.sys_level, f, len(f), 0, out, cap)
        if n > cap:
            out = (C.c_int * n)()
            n = lib().orc_ltrie_match(self.h, t, len(t), self.sys_level, f, len(f), 0, out, n)
        return list(out[:n])

    def find_all(self) -> List[int]:
        cap = 1 << 16
        out = (C.c_int * cap)()
        n = lib().orc_ltrie_match(self.h, b"", 0, self.sys_level, b"", 0, 1, out, cap)
        return list(out[:n])

    def match_batch(self, tenants: Sequence, filter_tenant: np.ndarray, filters_packed, threads: int = 1):
        tdata, toff = pack(tenants)
        data, off = filters_packed
        ft = np.ascontiguousarray(filter_tenant, dtype=np.uint32)
        n = len(off) - 1
        res = MatchResult(n)
        sec = lib().orc_ltrie_match_batch(self.h, _ptr(tdata), _ptr(toff), _ptr(ft), _ptr(data), _ptr(off), n,
                                          threads, res.h)
        return res, sec

    @property
    def visits(self) -> int:
        return lib().orc_ltrie_visits(self.h)

    def __del__(self):
        if self.h:
            lib().orc_ltrie_free(self.h)
            self.h = None


def retain_expire_at(timestamp_hlc: int, expiry_seconds: int) -> int:
    """RetainStoreCoProc.expireAt (RS/RetainStoreCoProc.java:298-304): physical part of the HLC timestamp (base-hlc
    HLC.java:145-151: the upper 48 bits, milliseconds) + expirySeconds, in milliseconds."""
    return (timestamp_hlc >> 16) + expiry_seconds * 1000


def retain_store_match(lt: "LevelTrie", tenant, topic_filter, limit: int, now: int, expire_at_of) -> List[int]:
    """RetainStoreCoProc.match(tenantId, topicFilter, limit, now, reader) (RS/RetainStoreCoProc.java:167-190): the FULL match
    set of the index, walked until `limit` messages that have not expired (expireAt > now) are collected.  The reference
    walks a HashSet (UTIL/index/StrategySet.java:31) -- which `limit` survive is unspecified there; this restatement (and the
    engine) walk in ascending topic id.  expire_at_of(id) -> expiry instant in ms."""
    if limit == 0:
        return []
    out = []
    for i in sorted(lt.match(tenant, topic_filter)):
        if len(out) >= limit:
            break
        if expire_at_of(i) > now:
            out.append(i)
    return out


def range_lookup(tenant: str, topic: str, candidates) -> List[int]:
    """TenantRangeLookupCache.lookup(CacheKey) (bifromq-dist-server .../scheduler/TenantRangeLookupCache.java:69-106) for one
    topic: candidates in boundary order, each None (no Fact: kept), () (Fact without first/last: an empty range, dropped) or
    (first_levels, last_levels) -- GLOBAL filter levels, tenant id first (Fact.proto:27-34).  Walks the structural restatement of
    the expansion iterator over the one-topic global trie.  -> indices of the candidates kept."""
    trie = TopicTrie(True).add_topic(tenant + "/" + topic, 0)
    itr = TopicFilterIterator(trie)
    out = []
    for idx, cand in enumerate(candidates):
        if cand is None:
            out.append(idx)
            continue
        if len(cand) == 0:
            continue
        first, last = cand
        itr.seek("/".join(first))
        if itr.is_valid():
            key = itr.key().split("/")
            if key == list(first) or "\0".join(key) <= "\0".join(last):
                out.append(idx)
        else:
            break
    return out


# ---- retain store key schema (bifromq-retain-store-schema KVSchemaUtil.java:44-73, LevelHash.java:30-50) ---------------------
def retain_level_hash(levels) -> bytes:
    """LevelHash.hash: per level FNV-1a 32 over the UTF-16 code units (Java chars), lowest byte"""
    out = bytearray()
    for lv in levels:
        h = 0x811C9DC5
        u16 = lv.encode("utf-16-le")
        for i in range(0, len(u16), 2):
            h ^= u16[i] | (u16[i + 1] << 8)
            h = (h * 0x01000193) & 0xFFFFFFFF
        out.append(h & 0xFF)
    return bytes(out)


def retain_tenant_begin_key(tenant: str) -> bytes:
    t = tenant.encode()
    return b"\x00" + len(t).to_bytes(2, "big") + t


def retain_message_key(tenant: str, topic: str) -> bytes:
    levels = topic.split("/")
    return retain_tenant_begin_key(tenant) + len(levels).to_bytes(2, "big") + retain_level_hash(levels) + topic.replace("/", "\0").encode()


def retain_filter_prefix(levels):
    """KVSchemaUtil.filterPrefix"""
    if "+" in levels:
        return levels[:levels.index("+")]
    if levels[-1] == "#":
        return levels[:-1]
    return levels


def retain_key_prefix_of_filter(tenant: str, topic_filter: str) -> bytes:
    """KVSchemaUtilTest.toRetainMessageKeyPrefix: retainKeyPrefix(tenant, levels, filterPrefix(parse(filter)))"""
    levels = topic_filter.split("/")
    n = len(levels) - 1 if levels[-1] == "#" else len(levels)
    return retain_tenant_begin_key(tenant) + n.to_bytes(2, "big") + retain_level_hash(retain_filter_prefix(levels))



# ---- KV range router (base-kv BoundaryUtil.java:122-195,241-252,299-339; KVRangeRouterUtil.java:41-103) and the retain-server's
# MatchCallRangeRouter.rangeLookup (MatchCallRangeRouter.java:56-134).  A boundary is (start | None, end | None); the router is a list
# of boundaries and every TreeMap operation is restated as a linear scan over it in BoundaryUtil.compare order. ----------------------
def _cmp(a: bytes, b: bytes) -> int:
    return (a > b) - (a < b)  # bytes compare as unsigned, lexicographically: ByteString.unsignedLexicographicalComparator


def boundary_compare_start(a, b) -> int:
    """BoundaryUtil.compareStartKey (:159-170): null is the smallest start"""
    if a is None and b is None:
        return 0
    if a is None:
        return -1
    if b is None:
        return 1
    return _cmp(a, b)


def boundary_compare_end(a, b) -> int:
    """BoundaryUtil.compareEndKeys (:184-195): null is the greatest end"""
    if a is None and b is None:
        return 0
    if a is None:
        return 1
    if b is None:
        return -1
    return _cmp(a, b)


def boundary_compare(b1, b2) -> int:
    """BoundaryUtil.compare(Boundary, Boundary) (:142-145)"""
    c = boundary_compare_start(b1[0], b2[0])
    return c if c else boundary_compare_end(b1[1], b2[1])


def boundary_upper_bound(key: bytes):
    """BoundaryUtil.upperBound (:299-339): strip trailing 0xFF, bump the last byte; None = open end"""
    i = len(key)
    if i == 0:
        return None
    while True:
        i -= 1
        if i < 0 or key[i] < 0xFF:
            break
    if i < 0:
        return None
    return key[:i] + bytes([key[i] + 1])


def boundary_in_range(key: bytes, b) -> bool:
    """BoundaryUtil.inRange(key, boundary) (:241-263)"""
    if b[0] is not None and _cmp(key, b[0]) < 0:
        return False
    if b[1] is not None:
        return _cmp(key, b[1]) < 0
    return True


def _sorted_router(router):
    import functools
    return sorted(range(len(router)), key=functools.cmp_to_key(lambda i, j: boundary_compare(router[i], router[j])))


def router_find_by_key(key: bytes, router):
    """KVRangeRouterUtil.findByKey (:41-52): floorEntry(Boundary{startKey = key}), then inRange -> index into router or None"""
    probe = (key, None)
    floor = None
    for i in _sorted_router(router):
        if boundary_compare(router[i], probe) <= 0:
            floor = i
    if floor is not None and boundary_in_range(key, router[floor]):
        return floor
    return None


def router_find_by_boundary(boundary, router) -> List[int]:
    """KVRangeRouterUtil.findByBoundary (:54-103) -> indices into router, in router order.  TreeMap.subMap throws where
    fromKey > toKey; that case yields [] here."""
    order = _sorted_router(router)
    if not order:
        return []
    start, end = boundary
    if start is None and end is None:
        return order
    if start is None:
        b_end = (end, end)
        return [i for i in order if boundary_compare(router[i], b_end) < 0]  # headMap(boundaryEnd, false)
    b_start = (start, start)
    floor = None
    for i in order:
        if boundary_compare(router[i], b_start) <= 0:
            floor = i
    if floor is None:
        floor = order[0]
    include_from = boundary_compare_end(router[floor][1], start) > 0
    out = []
    for i in order:
        c = boundary_compare(router[i], router[floor])
        if c < 0 or (c == 0 and not include_from):
            continue
        if end is not None and boundary_compare(router[i], (end, end)) >= 0:  # subMap(.., boundaryEnd, false) / tailMap
            continue
        out.append(i)
    return out


def retain_parse_level_hash(key: bytes) -> bytes:
    """KVSchemaUtil.parseLevelHash of the retain schema (KVSchemaUtil.java:79-85)"""
    tl = int.from_bytes(key[1:3], "big")
    lv = 3 + tl
    n = int.from_bytes(key[lv:lv + 2], "big")
    if len(key) < lv + 2 + n:
        raise IndexError("parseLevelHash")
    return key[lv + 2:lv + 2 + n]


def retain_range_lookup(tenant: str, topic_filter: str, router) -> List[int]:
    """MatchCallRangeRouter.rangeLookup for one topic filter (:58-95) + findCandidates (:96-133) -> indices into router"""
    levels = topic_filter.split("/")
    if "+" not in levels and levels[-1] != "#":
        i = router_find_by_key(retain_message_key(tenant, topic_filter), router)
        assert i is not None
        return [i]
    fixed = levels[-1] != "#"
    prefix = retain_filter_prefix(levels)
    n = len(levels) if fixed else len(levels) - 1
    begin = retain_tenant_begin_key(tenant) + n.to_bytes(2, "big") + retain_level_hash(prefix)
    if fixed:
        return router_find_by_boundary((begin, boundary_upper_bound(begin)), router)
    end = boundary_upper_bound(retain_tenant_begin_key(tenant))
    cands = router_find_by_boundary((begin, end), router)
    if not prefix:
        return cands
    lh = retain_level_hash(prefix)
    lh_ub = boundary_upper_bound(lh)  # None: the reference dereferences it (NullPointerException); treated as open here
    out = []
    for i in cands:
        cs, ce = router[i]
        if boundary_compare_start(cs, begin) > 0:
            h = retain_parse_level_hash(cs)
            if lh_ub is not None and _cmp(lh_ub, h) <= 0:
                continue
        if boundary_compare_end(ce, end) <= 0 and _cmp(retain_parse_level_hash(ce), lh) <= 0:
            continue
        out.append(i)
    return out


# ---- fan-out grouping: what happens to the matched routes of a batch behind the match (DW/DeliverExecutorGroup.java:112-241,
# DW/DeliverExecutor.java:85-90, bifromq-deliverer BatchDeliveryCall.java:71-76, DelivererKey.java:22) ---------------------------
def deliverer_key_of(route_key_bytes: bytes):
    """-> (subBrokerId, delivererKey) of a normal route (SCHEMA/cache/ReceiverCache.java:32-37: receiverUrl.split(NUL), parts[0]
    parsed as int, parts[2]), or None for a shared-subscription route (flag 2 / 3)."""
    flag, _tenant, _filt, recv = parse_route_key(route_key_bytes)
    if flag != FLAG_NORMAL:
        return None
    parts = recv.split("\0")
    return int(parts[0]), parts[2]


def fanout_groups(key_of, per_topic_routes):
    """DeliverExecutorGroup.submit for every topic of a batch: each NormalMatching -> DeliverExecutor.send -> a DeliveryCall batched under
    DelivererKey(subBrokerId, delivererKey); BatchDeliveryCall.add files it under tenant -> message pack (topic) -> MatchInfo set.
    key_of(route id) -> route key bytes, or None if the route is gone.
    -> ({DelivererKey: [(topic index, route id), ...] in (topic, route) order}, shared-subscription pairs, dead pairs)"""
    batches, shared, dead = {}, [], []
    for t, routes in enumerate(per_topic_routes):
        for rid in routes:
            k = key_of(rid)
            if k is None:
                dead.append((t, rid))
                continue
            dk = deliverer_key_of(k)
            if dk is None:
                shared.append((t, rid))  # GroupMatching: the receiver is picked per message (DeliverExecutorGroup.java:243-279)
            else:
                batches.setdefault(dk, []).append((t, rid))
    return batches, shared, dead


# ---- TenantRouteCache's patch path (DW/cache/TenantRouteCache.java:208-296) at the level of route identity (= KV route key) ---------
class TenantRouteCacheModel:
    """One tenant's route cache as the reference maintains it: a load stores matchAll(singleton(topic)) (:217-230) and files the topic in
    the TopicIndex; AddRoutes / RemoveRoutes tasks PATCH the cached rows index.match(filterLevels) finds (:243-291) through
    MatchedRoutes.addNormalMatching / removeNormalMatching / putGroupMatching / removeGroupMatching (MatchedRoutes.java:87-141, caps
    INT_MAX).  Used to pin that dropping those rows and re-matching them -- what bmq_route_cache_apply does -- serves the same sets."""

    def __init__(self, tenant: str):
        self.tenant = tenant
        self.index = LevelTrie(0)      # TopicIndex<RouteCacheKey>
        self.topic_no = {}             # topic -> value filed in the index
        self.topics = []
        self.cached = {}               # topic -> set of route keys (a group route = its one key: MatchedRoutes keeps one GroupMatching
                                       # per mqttTopicFilter, putGroupMatching replaces it)

    def load(self, topic: str, kv_keys):
        """LoadEntryTask: matchAll(singleton(topic)) over the CURRENT routes, cacheKey filed in the index"""
        rows = KV(sorted(kv_keys)).match_bruteforce(self.tenant, [topic]).per_topic()[0]
        ks = sorted(kv_keys)
        self.cached[topic] = {ks[r] for r in rows}
        if topic not in self.topic_no:
            self.topic_no[topic] = len(self.topics)
            self.topics.append(topic)
        self.index.add(None, topic, self.topic_no[topic])

    def evict(self, topic: str):
        """Caffeine's removalListener: index.remove(key.topic, key) (:117-118)"""
        if topic in self.cached:
            del self.cached[topic]
            self.index.remove(None, topic, self.topic_no[topic])

    def _hit(self, filter_levels):
        return [self.topics[v] for v in self.index.match(None, "/".join(filter_levels))]

    def add_routes(self, filter_levels, route_keys):
        """AddRoutesTask (:243-266): addNormalMatching / putGroupMatching on every cached row the filter matches"""
        for topic in self._hit(filter_levels):
            if topic in self.cached:
                self.cached[topic] |= set(route_keys)

    def remove_routes(self, filter_levels, route_keys, group_still_has_members=False):
        """RemoveRoutesTask (:268-291): removeNormalMatching; a group route is removed only when its last member left"""
        for topic in self._hit(filter_levels):
            if topic in self.cached and not group_still_has_members:
                self.cached[topic] -= set(route_keys)


# ---- MatchedRoutes + getMatch with finite fan-out caps (DW/cache/MatchedRoutes.java:36-200, DW/cache/TenantRouteCache.java:116-139,299-301) ----
class MatchedRoutesModel:
    """MatchedRoutes at the level of route identity (= KV route key), method by method:
    addNormalMatching :87-109, removeNormalMatching :111-117, putGroupMatching :119-141, removeGroupMatching :143-155, adjust :157-200.
    `events` collects what the reference hands to IEventCollector.report: (type 0 = PersistentFanoutThrottled | 1 = GroupFanoutThrottled,
    route key, maxCount)."""
    ADDED, EXISTS, EXCEED = "Added", "Exists", "ExceedFanoutLimit"
    ADJUSTED, CLAMPED, RELOAD = "Adjusted", "Clamped", "ReloadNeeded"

    def __init__(self, max_persistent_fanout: int = INT_MAX, max_group_fanout: int = INT_MAX):
        self.max_pf, self.max_gf = max_persistent_fanout, max_group_fanout
        self.all = set()        # allMatchings
        self.groups = {}        # groupMatchings: mqttTopicFilter (incl. $share/<group>/) -> route key
        self.persistent = 0     # persistentFanout
        self.events = []

    @staticmethod
    def _kind(key: bytes):
        flag, _tenant, mqtt_filter, receiver = parse_route_key(key)
        if flag == FLAG_NORMAL:
            return "normal", mqtt_filter, receiver.split("\0")[0] == "1"   # subBrokerId() == 1: the inbox (persistent session) broker
        return "group", mqtt_filter, False

    def add(self, key: bytes):
        """what TenantRouteMatcher.matchAll / an AddRoutesTask does with one Matching"""
        kind, mqtt_filter, is_persistent = self._kind(key)
        if kind == "normal":
            if key in self.all:
                return self.EXISTS
            self.all.add(key)
            if is_persistent:
                if self.persistent < self.max_pf:
                    self.persistent += 1
                    return self.ADDED
                self.all.discard(key)
                self.events.append((0, key, self.max_pf))
                return self.EXCEED
            return self.ADDED
        prev = self.groups.get(mqtt_filter)
        self.groups[mqtt_filter] = key
        if prev is None:
            if len(self.groups) <= self.max_gf:
                self.all.add(key)
                return self.ADDED
            del self.groups[mqtt_filter]
            self.events.append((1, key, self.max_gf))
            return self.EXCEED
        self.all.discard(prev)
        self.all.add(key)
        return self.EXISTS

    def remove(self, key: bytes):
        kind, mqtt_filter, is_persistent = self._kind(key)
        if kind == "normal":
            if key in self.all:
                self.all.discard(key)
                if is_persistent:
                    self.persistent -= 1
        else:
            existing = self.groups.pop(mqtt_filter, None)
            if existing is not None:
                self.all.discard(existing)

    def adjust(self, new_pf: int, new_gf: int) -> str:
        if self.max_pf < new_pf and self.persistent == self.max_pf:
            return self.RELOAD
        if self.max_gf < new_gf and len(self.groups) == self.max_gf:
            return self.RELOAD
        clamped = False
        if self.max_pf > new_pf and self.persistent > new_pf:
            # the reference removes the first `toRemove` persistent routes a ConcurrentHashMap key set yields: WHICH ones is unspecified
            victims = [k for k in sorted(self.all) if self._kind(k) == ("normal", self._kind(k)[1], True)][:self.persistent - new_pf]
            for k in victims:
                self.remove(k)
            clamped = True
        if self.max_gf > new_gf and len(self.groups) > new_gf:
            for f in sorted(self.groups)[:len(self.groups) - new_gf]:
                self.remove(self.groups[f])
            clamped = True
        self.max_pf, self.max_gf = new_pf, new_gf
        return self.CLAMPED if clamped else self.ADJUSTED

    def routes(self):
        return set(self.all)


def matched_routes_load(tenant: str, topic: str, kv_keys, max_persistent_fanout: int, max_group_fanout: int) -> MatchedRoutesModel:
    """LoadEntryTask / ReloadEntryTask (TenantRouteCache.java:217-241): matchAll(singleton(topic), maxPF, maxGF) -- the KV iterator yields
    the route keys ascending, every key whose filter matches goes through addNormalMatching / putGroupMatching in that order
    (TenantRouteMatcher.java:106-126)."""
    mr = MatchedRoutesModel(max_persistent_fanout, max_group_fanout)
    for key in sorted(kv_keys):
        flag, key_tenant, mqtt_filter, _receiver = parse_route_key(key)
        if key_tenant != tenant:
            continue
        plain = mqtt_filter
        if flag != FLAG_NORMAL:  # "$share/<group>/<filter>" / "$oshare/..."
            plain = mqtt_filter.split("/", 2)[2]
        if semantic_match(topic, plain):
            mr.add(key)
    return mr


class CappedTenantRouteCacheModel:
    """TenantRouteCache.getMatch (TenantRouteCache.java:299-301) for one tenant with finite caps, as the reference serves it:
      * a miss loads matchAll(singleton(topic), maxPF, maxGF) and caches the MatchedRoutes (:217-230);
      * AddRoutes / RemoveRoutes tasks PATCH the cached MatchedRoutes first-come (:243-291) -- a route added while a cap is reached is
        thrown away (with an event) although a fresh load would admit it in front of a later key;
      * the periodic fan-out check (asyncReload, :124-138) calls adjust(new caps) and reloads when it says ReloadNeeded.
    get_match(topic, reload=True) is the state after that entry's next (re)load -- what bmq_route_cache_* serves at once, because it
    drops patched rows and re-matches them."""

    def __init__(self, tenant: str, max_persistent_fanout: int = INT_MAX, max_group_fanout: int = 100):
        self.tenant = tenant
        self.max_pf, self.max_gf = max_persistent_fanout, max_group_fanout
        self.kv = set()
        self.cached = {}   # topic -> MatchedRoutesModel
        self.events = []   # (topic, type, route key, maxCount) in report order

    def _drain(self, topic, mr):
        self.events += [(topic,) + e for e in mr.events]
        mr.events = []

    def get_match(self, topic: str, reload: bool = False):
        mr = self.cached.get(topic)
        if mr is None or reload or (mr.max_pf, mr.max_gf) != (self.max_pf, self.max_gf) and mr.adjust(self.max_pf, self.max_gf) == mr.RELOAD:
            mr = matched_routes_load(self.tenant, topic, self.kv, self.max_pf, self.max_gf)
            self.cached[topic] = mr
        self._drain(topic, mr)
        return mr.routes()

    def refresh(self, added=(), removed=()):
        """DistWorkerCoProc.mutate's post-commit refresh (DW/DistWorkerCoProc.java:188-209): the KV has changed, cached rows are patched"""
        for key in removed:
            self.kv.discard(key)
        for key in added:
            self.kv.add(key)
        for topic, mr in self.cached.items():
            for key in removed:
                flag, _t, mqtt_filter, _r = parse_route_key(key)
                if semantic_match(topic, mqtt_filter if flag == FLAG_NORMAL else mqtt_filter.split("/", 2)[2]):
                    mr.remove(key)
            for key in added:
                flag, _t, mqtt_filter, _r = parse_route_key(key)
                if semantic_match(topic, mqtt_filter if flag == FLAG_NORMAL else mqtt_filter.split("/", 2)[2]):
                    mr.add(key)
            self._drain(topic, mr)
