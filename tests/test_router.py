"""the KV range router either side of the retain match -- KVRangeRouterUtil.findByKey / findByBoundary and
MatchCallRangeRouter.rangeLookup -- product (bifromq_amd/csrc/bmq_router.cpp through the C ABI) against the oracle restatement
(oracle/oracle.py) and the reference's own test vectors:
  base-kv/base-kv-store-client/src/test/java/org/apache/bifromq/basekv/client/KVRangeRouterUtilTest.java:57-257
  bifromq-retain/bifromq-retain-server/src/test/java/org/apache/bifromq/retain/server/scheduler/MatchRetainedRequestRangeRouterTest.java:72-147
Host arithmetic only: runs without a GPU."""
import random

import pytest

from bifromq_amd.engine import BmqError, RangeRouter
from oracle import oracle as O

FULL = (None, None)
NULL_BOUNDARY = (None, b"")
TENANT = "testTenant"


@pytest.fixture(autouse=True, params=["arrays", "object"])
def router_form(request, monkeypatch):
    """Every test runs twice: lookups over the caller's arrays (the router is indexed per call) and over a router OBJECT built once
    (bmq_router_create: the boundaries are copied, checked and indexed once) -- the two forms must agree with the oracle alike."""
    if request.param == "object":
        plain_init = RangeRouter.__init__

        def init_and_build(self, boundaries):
            plain_init(self, boundaries)
            self.build()

        monkeypatch.setattr(RangeRouter, "__init__", init_and_build)
    return request.param


def _abcd_router():
    # (null,"b") ["b","c") ["c","d") ["d",null)   KVRangeRouterUtilTest.java:66-117
    return [(None, b"b"), (b"b", b"c"), (b"c", b"d"), (b"d", None)]


def test_find_by_key_vectors():
    bs = _abcd_router()
    r = RangeRouter(bs)
    for key, want in ((b"a", 0), (b"b", 1), (b"c", 2), (b"d", 3), (b"z", 3)):  # KVRangeRouterUtilTest.java:125-152
        assert r.find_by_key(key) == want
        assert O.router_find_by_key(key, bs) == want
    assert RangeRouter([]).find_by_key(b"a") is None  # emptyRouter :58-61
    assert O.router_find_by_key(b"a", []) is None


def test_find_by_boundary_vectors():
    bs = _abcd_router()
    r = RangeRouter(bs)
    cases = [((None, b"a"), [0]), ((None, b"b"), [0]), ((b"b", None), [1, 2, 3]), ((b"b", b"d"), [1, 2]), ((b"x", b"y"), [3]),
             (FULL, [0, 1, 2, 3]), (NULL_BOUNDARY, [0])]  # KVRangeRouterUtilTest.java:213-256
    for q, want in cases:
        assert r.find_by_boundary(*q) == want, q
        assert O.router_find_by_boundary(q, bs) == want, q
    assert RangeRouter([]).find_by_boundary(None, None) == []


def _split_router(tenant, topics):
    """MatchRetainedRequestRangeRouterTest.createEffectiveRouter (:149-169): FULL_BOUNDARY split at retainMessageKey(topic)"""
    bs = [FULL]
    for t in topics:
        k = O.retain_message_key(tenant, t)
        nxt = []
        for b in bs:
            if O.boundary_in_range(k, b) and b[0] != k:  # a key that already is a split point splits nothing
                nxt += [(b[0], k), (k, b[1])]
            else:
                nxt.append(b)
        bs = nxt
    import functools
    return sorted(bs, key=functools.cmp_to_key(O.boundary_compare))


def _lookup_both(tenant, filters, bs):
    got = RangeRouter(bs).retain_range_lookup(tenant, filters)
    for f, g in zip(filters, got):
        assert g == O.retain_range_lookup(tenant, f, bs), (f, bs)
    return got


def test_range_lookup_reference_cases():
    # testNonWildcardTopicFilter :72-86
    bs = _split_router(TENANT, [])
    assert _lookup_both(TENANT, ["a/b/c"], bs) == [[0]] and bs[0] == FULL
    # testWildcardTopicFilterWithoutMultiWildcard :88-101
    bs = _split_router(TENANT, ["a/a", "a/b", "a/b/c"])
    got = _lookup_both(TENANT, ["a/+/c"], bs)[0]
    assert any(O.boundary_in_range(O.retain_message_key(TENANT, t), bs[i]) for i in got for t in ("a/a/c", "a/c/c"))
    # every topic the filter can match lives in a kept range
    for t in ("a/a/c", "a/c/c", "a//c", "a/zzz/c"):
        assert O.router_find_by_key(O.retain_message_key(TENANT, t), bs) in got
    # testMultiWildcardTopicFilterWithEmptyFilterPrefix :103-114: '#' goes to every range
    assert _lookup_both(TENANT, ["#"], bs) == [list(range(len(bs)))]
    # testMultiWildcardTopicFilterWithNonEmptyFilterPrefix :116-132: one of the four ranges is pruned by findCandidates
    got = _lookup_both(TENANT, ["a/b/#"], bs)[0]
    assert len(bs) == 4 and len(got) == 2
    assert any(O.boundary_in_range(O.retain_message_key(TENANT, t), bs[i]) for i in got for t in ("a/b", "a/b/c"))
    for t in ("a/b", "a/b/c", "a/b/c/d/e"):
        assert O.router_find_by_key(O.retain_message_key(TENANT, t), bs) in got


def _rand_level(rng):
    return rng.choice(["a", "b", "c", "", "sensor", "x1", "你好", "$sys", "zz"])


def _rand_topic(rng):
    return "/".join(_rand_level(rng) for _ in range(rng.randint(1, 5)))


def _rand_filter(rng):
    lv = [_rand_level(rng) for _ in range(rng.randint(1, 5))]
    kind = rng.random()
    if kind < 0.35:
        lv[rng.randrange(len(lv))] = "+"
    if 0.25 < kind < 0.7:
        lv.append("#")
    if kind > 0.9:
        lv = ["#"] if rng.random() < 0.5 else ["+"] + lv[1:]
    return "/".join(lv)


def _exact_ranges(tenant, topic_filter, bs):
    """independent statement of BMQ_ROUTER_EXACT for a filter ending in '#': the ranges that meet one of the key intervals
    [tenant | L | hash(prefix), tenant | L | upperBound(hash(prefix))), L >= levels (brute force over L)"""
    levels = topic_filter.split("/")
    prefix = O.retain_filter_prefix(levels)
    n = len(levels) - 1
    tb = O.retain_tenant_begin_key(tenant)
    h = O.retain_level_hash(prefix)
    hub = O.boundary_upper_bound(h)
    out = set()
    for L in list(range(n, 12)) + [0xFFFF]:
        base = tb + L.to_bytes(2, "big")
        iv = (base + h, base + hub if hub is not None else O.boundary_upper_bound(base))
        for i, b in enumerate(bs):
            s = iv[0] if O.boundary_compare_start(b[0], iv[0]) < 0 else b[0]
            e = iv[1] if O.boundary_compare_end(b[1], iv[1]) > 0 else b[1]
            if e is None or s < e:
                out.add(i)
    return sorted(out)


def test_range_lookup_random_vs_oracle_and_soundness():
    rng = random.Random(20260923)
    lost = 0
    for it in range(150):
        tenant = rng.choice(["t", "tenantA", "租户"])
        other = "u" if tenant == "t" else "t"
        topics = [_rand_topic(rng) for _ in range(rng.randint(0, 12))]
        bs = _split_router(tenant, topics)
        # boundaries that belong to other tenants either side, as in a shared store
        for t in (_rand_topic(rng), _rand_topic(rng)):
            k = O.retain_message_key(other, t)
            bs = [x for b in bs for x in (((b[0], k), (k, b[1])) if O.boundary_in_range(k, b) and b[0] != k else (b,))]
        filters = [_rand_filter(rng) for _ in range(20)] + [_rand_topic(rng) for _ in range(5)]
        got = _lookup_both(tenant, filters, bs)  # reference mode == oracle restatement, filter by filter
        exact = RangeRouter(bs).retain_range_lookup(tenant, filters, exact=True)
        probe = topics + [_rand_topic(rng) for _ in range(30)]
        for f, g, x in zip(filters, got, exact):
            lv = f.split("/")
            if lv[-1] == "#" and O.retain_filter_prefix(lv):
                assert x == _exact_ranges(tenant, f, bs), (f, bs)
            else:
                assert x == g  # the modes differ only in findCandidates
            for t in probe:
                if O.semantic_match(t, f) or t == f:
                    home = O.router_find_by_key(O.retain_message_key(tenant, t), bs)
                    assert home in x, (f, t, bs)  # exact mode: the range holding a matching retained topic is always asked
                    lost += home not in g
    assert lost > 0  # ... which the reference's pruning rules do not guarantee (next test)


def test_find_candidates_prunes_a_range_spanning_two_level_counts():
    """MatchCallRangeRouter.findCandidates (:112-128) judges a range by the LevelHash of its start / end key alone.  The last range
    below starts at a 4-level key whose first hash byte sorts behind hash("a") and is open-ended, so it holds every 5-level topic --
    also `a/b/c/d/e`, which `a/#` matches.  Rule one (start key behind the prefix, hash of the start key >= upperBound(hash(prefix)))
    drops it: the reference does not ask that range.  Reference mode reproduces this; BMQ_ROUTER_EXACT keeps the range."""
    tenant = "tenantA"
    bs = _split_router(tenant, ["你好/a/a/zz"])  # LevelHash starts with 0xC5 > hash("a") = 0x2C
    assert O.retain_level_hash(["你好"])[0] > O.retain_level_hash(["a"])[0]
    home = O.router_find_by_key(O.retain_message_key(tenant, "a/b/c/d/e"), bs)
    assert home == 1 and bs[1][1] is None
    r = RangeRouter(bs)
    assert r.retain_range_lookup(tenant, ["a/#"]) == [O.retain_range_lookup(tenant, "a/#", bs)] == [[0]]
    assert r.retain_range_lookup(tenant, ["a/#"], exact=True) == [[0, 1]]


def test_router_argument_checks():
    with pytest.raises(BmqError):  # not in BoundaryUtil.compare order
        RangeRouter([(b"b", b"c"), (None, b"b")]).find_by_key(b"a")
    # a router with a hole: a plain topic nobody holds is an error (the reference asserts)
    k = O.retain_message_key(TENANT, "a/b")
    with pytest.raises(BmqError):
        RangeRouter([(None, k[:4])]).retain_range_lookup(TENANT, ["a/b"])
    # router that does not reach the query start: TreeMap.subMap would throw; nothing is returned
    assert RangeRouter([(b"x", b"y")]).find_by_boundary(b"a", b"b") == []
    assert O.router_find_by_boundary((b"a", b"b"), [(b"x", b"y")]) == []


def test_router_rejects_hostile_offsets(router_form):
    """Offsets that decrease, do not start at 0, or reach beyond the byte arrays are refused (BMQ_E_INVAL) instead of building views
    outside the router object's own copy of the boundaries; a query boundary flagged present with a NULL pointer is refused too."""
    import ctypes as C

    from bifromq_amd import _lib

    for damage in ("decreasing", "nonzero_first", "end_decreasing"):
        r = object.__new__(RangeRouter)
        bs = _abcd_router()
        r.n = len(bs)
        import numpy as np

        from bifromq_amd.engine import pack
        r.flags = np.array([(1 if s is not None else 0) | (2 if e is not None else 0) for s, e in bs], dtype=np.uint8)
        r.start, r.start_off = pack([s or b"" for s, _ in bs])
        r.end, r.end_off = pack([e or b"" for _, e in bs])
        r.h = None
        if damage == "decreasing":
            r.start_off[2] = 0xFFFFFFF0
        elif damage == "nonzero_first":
            r.start_off[0] = 1
        else:
            r.end_off[1], r.end_off[2] = r.end_off[2] + 1, r.end_off[1]
        with pytest.raises(BmqError):
            if router_form == "object":
                r.build()
            else:
                r.find_by_key(b"a")
    good = RangeRouter(_abcd_router())
    first, count = C.c_uint32(), C.c_uint32()
    L = _lib.lib()
    if router_form == "object":
        rc = L.bmq_router_lookup_boundary(good.h, 1, None, 3, b"", 0, C.byref(first), C.byref(count))
    else:
        rc = L.bmq_router_find_by_boundary(*good._args(), 1, None, 3, b"", 0, C.byref(first), C.byref(count))
    assert rc < 0


def test_router_under_sanitizers():
    """tools/router_fuzz.cpp: bmq_router.cpp under ASan + UBSan with hostile boundary keys (empty, truncated inside the key header,
    all 0xFF, other tenants'): a lookup succeeds or reports BMQ_E_INVAL, find_by_boundary returns exactly the overlapping ranges, EXACT
    mode always asks the range holding a matching retained topic."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "bifromq_amd", "csrc"), "routerfuzz"], check=True, capture_output=True, timeout=600)
    for seed in ("1", "7"):
        r = subprocess.run([os.path.join(root, "tools", "router_fuzz"), seed, "500"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "router_fuzz ok" in r.stdout, r.stdout + r.stderr


def test_boundary_util_vectors():
    """base-kv/base-kv-type-proto/src/test/java/org/apache/bifromq/basekv/utils/BoundaryUtilTest.java: findUpperBound (:148-169) and the
    compare* cases (:396-525) -- for the oracle's restatement, and for the product through the order it accepts a router in (it takes the
    boundaries of a TreeMap ordered by BoundaryUtil.compare: strictly ascending or BMQ_E_INVAL)."""
    ub = O.boundary_upper_bound
    assert ub(b"") is None  # MIN_KEY
    assert ub(bytes([1, 2, 3])) == bytes([1, 2, 4])
    assert ub(bytes([1, 2, 0xFF])) == bytes([1, 3])
    assert ub(bytes([1, 0xFF, 0xFF])) == bytes([2])
    assert ub(bytes([0xFF, 0xFF, 0xFF])) is None
    for k in (b"", bytes([1, 2, 3]), bytes([1, 2, 0xFF]), bytes([0xFF] * 3)):
        assert O.boundary_compare_end(k, ub(k)) < 0
    assert O.boundary_compare_start(None, None) == 0 and O.boundary_compare_start(None, b"a") == -1
    assert O.boundary_compare_start(b"a", None) == 1 and O.boundary_compare_start(b"a", b"b") == -1
    assert O.boundary_compare_end(None, None) == 0 and O.boundary_compare_end(None, b"a") == 1
    assert O.boundary_compare_end(b"a", None) == -1 and O.boundary_compare_end(b"a", b"b") == -1
    cases = [((b"a", b"c"), (b"a", b"c"), 0), ((b"a", b"c"), (b"b", b"c"), -1), ((b"a", b"b"), (b"a", b"c"), -1),
             ((b"a", None), (b"a", b"c"), 1), ((None, None), (b"a", b"b"), -1), ((b"a", None), (b"a", b"b"), 1)]
    for b1, b2, want in cases:
        assert O.boundary_compare(b1, b2) == want, (b1, b2)

        def accepted(order):
            try:
                RangeRouter(order).find_by_key(b"zz")
                return True
            except BmqError:
                return False
        assert accepted([b1, b2]) == (want < 0) and accepted([b2, b1]) == (want > 0), (b1, b2)
