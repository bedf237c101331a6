"""Fan-out caps on the ISubscriptionCache seam: what TenantRouteCache.getMatch returns is IMatchedRoutes.routes()
(DW/cache/TenantRouteCache.java:299-301), the set AFTER MatchedRoutes' persistent / group caps (DW/cache/MatchedRoutes.java:87-141).

CPU part: the oracle's restatement of MatchedRoutes (oracle.MatchedRoutesModel) against the reference's own MatchedRoutesTest.java:59-336
(14 cases), the restated getMatch against the structural matchAll oracle, and what separates the reference's in-place patching from a
reload once a cap binds.  GPU part (-m gpu): the same rows and throttle events through bmq_route_cache_get / get_batch / get_async."""
import random
import threading

import numpy as np
import pytest

from oracle import oracle as O

TENANT, TOPIC, GROUP_FILTER = "tenantA", "sensor/temperature", "sensor/+"  # MatchedRoutesTest.java:48-50


def _normal(broker, receiver, deliverer, topic=TOPIC):
    return O.route_key(TENANT, topic, O.FLAG_NORMAL, O.receiver_url(broker, receiver, deliverer))


def _group(name, flt=GROUP_FILTER):
    return O.route_key(TENANT, flt, O.FLAG_UNORDERED, name)


def test_matched_routes_model_against_reference_cases():
    M = O.MatchedRoutesModel
    # addPersistentNormalMatchingWithinLimit :59-70
    r = M(2, 2)
    p = _normal(1, "receiverA", "delivererA")
    assert r.add(p) == M.ADDED and r.persistent == 1 and p in r.routes() and not r.events
    # addPersistentNormalMatchingExceedLimit :72-96
    r = M(1, 2)
    first, second = _normal(1, "receiverA", "delivererA"), _normal(1, "receiverB", "delivererB")
    assert r.add(first) == M.ADDED and r.add(second) == M.EXCEED
    assert r.persistent == 1 and second not in r.routes() and r.events == [(0, second, 1)]
    # addNonPersistentNormalMatchingDoesNotAffectPersistentFanout :98-108
    r = M(1, 2)
    n = _normal(2, "receiverC", "delivererC")
    assert r.add(n) == M.ADDED and r.persistent == 0 and n in r.routes()
    # addDuplicateNormalMatchingReturnsExists :110-123
    r = M(2, 2)
    assert r.add(p) == M.ADDED and r.add(p) == M.EXISTS and r.persistent == 1 and len(r.routes()) == 1
    # removePersistentNormalMatching :125-139, removeNonPersistentNormalMatching :141-151
    r = M(2, 2)
    r.add(p)
    r.remove(p)
    assert r.persistent == 0 and p not in r.routes()
    r.remove(p)
    assert r.persistent == 0
    r.add(n)
    r.remove(n)
    assert r.persistent == 0 and n not in r.routes()
    # putGroupMatchingWithinLimit :153-165, putGroupMatchingExceedLimit :167-193
    r = M(2, 2)
    g = _group("groupA")
    assert r.add(g) == M.ADDED and len(r.groups) == 1 and g in r.routes() and not r.events
    r = M(2, 1)
    ga, gb = _group("groupA"), _group("groupB")
    assert r.add(ga) == M.ADDED and r.add(gb) == M.EXCEED and len(r.groups) == 1 and gb not in r.routes() and r.events == [(1, gb, 1)]
    # putGroupMatchingReplacesExistingGroup :195-210 (same group + filter = the same route key; the membership lives in the value)
    r = M(2, 3)
    assert r.add(ga) == M.ADDED and r.add(ga) == M.EXISTS and len(r.groups) == 1 and ga in r.routes()
    # removeGroupMatching :212-224
    r = M(2, 2)
    r.add(ga)
    r.remove(ga)
    assert len(r.groups) == 0 and ga not in r.routes()
    r.remove(ga)
    assert len(r.groups) == 0
    # adjustReturnsReloadNeededWhenPersistentLimitIncreases :226-234, ...GroupLimitIncreases :236-245
    r = M(1, 2)
    r.add(p)
    assert r.adjust(2, r.max_gf) == M.RELOAD
    r = M(2, 1)
    r.add(ga)
    assert r.adjust(r.max_pf, 2) == M.RELOAD
    # adjustClampsPersistentFanoutWhenLimitDecreases :247-268
    r = M(3, 2)
    for k in (p, second, _normal(1, "receiverC", "delivererC")):
        r.add(k)
    assert r.adjust(1, r.max_gf) == M.CLAMPED and r.persistent == 1 and len(r.routes()) == 1
    assert r.adjust(1, r.max_gf) == M.ADJUSTED
    # adjustClampsGroupFanoutWhenLimitDecreases :270-292
    r = M(2, 3)
    for k in (ga, gb, _group("groupC")):
        r.add(k)
    assert r.adjust(r.max_pf, 1) == M.CLAMPED and len(r.groups) == 1 and len(r.routes()) == 1
    assert r.adjust(r.max_pf, 1) == M.ADJUSTED
    # adjustUpdatesLimitsWhenWithinBounds :294-303
    r = M(2, 2)
    assert r.adjust(4, 3) == M.ADJUSTED and (r.max_pf, r.max_gf) == (4, 3)


def _random_keys(rng, tenant, n, lv=("a", "b", "c", "", "$s")):
    keys = set()
    for _ in range(n):
        d = rng.randint(1, 3)
        f = []
        for k in range(d):
            x = rng.randint(0, 7)
            f.append("+" if x == 0 else ("#" if x == 1 and k == d - 1 else rng.choice(lv)))
        flt = "/".join(f)
        kind = rng.randint(0, 9)
        if kind < 5:
            keys.add(O.route_key(tenant, flt, O.FLAG_NORMAL, O.receiver_url(1, f"inbox{rng.randint(0, 30)}", f"d{rng.randint(0, 3)}")))
        elif kind < 8:
            keys.add(O.route_key(tenant, flt, O.FLAG_NORMAL, O.receiver_url(0, f"mqtt{rng.randint(0, 30)}", "d0")))
        else:
            keys.add(O.route_key(tenant, flt, rng.choice([O.FLAG_UNORDERED, O.FLAG_ORDERED]), f"g{rng.randint(0, 9)}"))
    return keys


def test_restated_load_equals_structural_match_all():
    """matched_routes_load (Python, rule by rule) == orc_match_all (the C++ restatement of TenantRouteMatcher.matchAll with caps):
    same kept routes, same events in the same order"""
    rng = random.Random(5)
    for trial in range(30):
        # (no empty levels here: a filter ending in an empty level can make the reference's probe-then-seek loop skip keys -- quirk (ii) of
        # DESIGN.md section 2 -- which is the structural oracle's business, not MatchedRoutes')
        keys = sorted(_random_keys(rng, "T", 60, lv=("a", "b", "c", "$s")))
        kv = O.KV(keys)
        for topic in ["a/b", "a", "b/c/a", "$s/a", "a//b", "c"]:
            pf, gf = rng.randint(0, 4), rng.randint(0, 3)
            mr = O.matched_routes_load("T", topic, keys, pf, gf)
            res = kv.match_all("T", [topic], pf, gf)
            assert sorted(keys.index(k) for k in mr.routes()) == res.per_topic()[0]
            assert [(t, keys.index(k), m) for t, k, m in mr.events] == [(t, r, m) for t, _i, r, m in res.events]


def test_patch_versus_reload_once_a_cap_binds():
    """The reference patches cached rows first-come (TenantRouteCache.java:243-291); the engine's cache drops and re-matches them.  Both
    serve the same set while no cap binds; when one does, the patched row depends on arrival order until the entry's next load -- and a
    load is all bmq_route_cache_* ever serves."""
    rng = random.Random(11)
    for trial in range(40):
        caps = (rng.randint(1, 4), rng.randint(1, 3)) if trial % 2 else (O.INT_MAX, O.INT_MAX)
        ref = O.CappedTenantRouteCacheModel("T", *caps)
        ref.refresh(added=_random_keys(rng, "T", 25))
        topics = ["a/b", "a", "b/c", "c/a/b"]
        for t in topics:
            ref.get_match(t)
        for step in range(12):
            added = _random_keys(rng, "T", 3)
            removed = set(rng.sample(sorted(ref.kv), min(2, len(ref.kv))))
            ref.refresh(added=added - removed, removed=removed)
            for t in topics:
                patched = ref.get_match(t)
                loaded = O.matched_routes_load("T", t, ref.kv, *caps)
                if caps[0] == O.INT_MAX:
                    assert patched == loaded.routes()  # no cap: patching == re-matching (the invariant bmq_route_cache_apply relies on)
                else:
                    assert len(patched) <= len(loaded.routes())  # a patched row never holds MORE than a load admits ...
                    assert ref.get_match(t, reload=True) == loaded.routes()  # ... and the reference's own reload lands on the load


# ---- GPU: through the C ABI ---------------------------------------------------------------------------------------------------------
def _caps_case():
    keys = set()
    for i in range(7):
        keys.add(O.route_key("T", "s/t" if i % 2 else "s/+", O.FLAG_NORMAL, O.receiver_url(1, f"inbox{i}", "d0")))
    for i in range(3):
        keys.add(O.route_key("T", "s/#", O.FLAG_NORMAL, O.receiver_url(0, f"mqtt{i}", "d1")))
    for i in range(5):
        keys.add(O.route_key("T", "+/t" if i % 2 else "s/t", O.FLAG_UNORDERED if i % 3 else O.FLAG_ORDERED, f"g{i}"))
    keys.add(O.route_key("T", "x", O.FLAG_NORMAL, O.receiver_url(1, "inbox9", "d0")))
    keys |= _random_keys(random.Random(3), "U", 40)
    return keys


@pytest.mark.gpu
def test_caps_through_the_route_cache_gpu():
    """group cap 2 / persistent cap 3 through bmq_route_cache_get*: miss and hit, before and after an apply, after set_caps; rows and
    throttle events equal the restated getMatch (a load per miss)."""
    import bifromq_amd as B
    keys = sorted(_caps_case())
    eng = B.Engine(device=0).rebuild(keys)
    b = eng.batcher()
    c = B.RouteCache(b, max_persistent_fanout=3, max_group_fanout=2)
    events = c.collect_events()
    live = set(keys)

    def key_of(ids):
        return set(eng.route_keys(np.array(ids, dtype=np.uint32))) if ids else set()

    def check(tenant, topic, caps, expect_load):
        nb = b.stats().n_batches
        del events[:]
        ids, _ep = c.get(tenant, topic, now_ms=1)
        want = O.matched_routes_load(tenant, topic, live, *caps)
        assert key_of(ids) == want.routes() and len(ids) == len(want.routes()), (tenant, topic)
        loaded = b.stats().n_batches != nb
        assert loaded == expect_load, (tenant, topic, caps)
        got = [(t.decode(), p.decode(), typ, eng.route_key(rid), mx) for t, p, typ, rid, mx in events]
        assert got == ([(tenant, topic, typ, k, mx) for typ, k, mx in want.events] if loaded else [])

    check("T", "s/t", (3, 2), True)    # 7 persistent -> 3, 5 groups -> 2, 3 transient: 4 + 3 events
    check("T", "s/t", (3, 2), False)   # hit: capped row, silent
    check("T", "x", (3, 2), True)
    st = c.stats()
    assert st.cached_routes == 3 + 2 + 3 + 1  # weighed by the capped size (TenantRouteCache.java:108)
    # the structural oracle agrees with the restatement on the engine's rows (ids are ranks after a rebuild)
    res = O.KV(keys).match_all("T", ["s/t"], 3, 2)
    assert c.get("T", "s/t", now_ms=1)[0] == res.per_topic()[0]
    # refresh: one persistent route whose key sorts in FRONT of the admitted ones, one group behind them, one admitted route removed
    add = [O.route_key("T", "+/t", O.FLAG_NORMAL, O.receiver_url(1, "inbox-new", "d0")), O.route_key("T", "s/t", O.FLAG_UNORDERED, "zz")]
    gone = sorted(k for k in live if O.parse_route_key(k)[3].startswith("1\0") and O.parse_route_key(k)[2] in ("s/+", "s/t"))[0]
    c.apply([(0, add[0]), (0, add[1]), (1, gone)])
    live = (live | set(add)) - {gone}
    check("T", "s/t", (3, 2), True)    # dropped by the mutation: a fresh load, caps in key order, events again
    check("T", "s/t", (3, 2), False)
    check("T", "x", (3, 2), False)     # untouched by the mutation
    # MatchedRoutes.adjust through set_caps
    c.set_caps("T", 4, 2)
    check("T", "s/t", (4, 2), True)    # the row sits at the old persistent cap: reload
    check("T", "x", (4, 2), False)     # nowhere near a cap: adopts the new caps
    c.set_caps("T", 4, 1)
    check("T", "s/t", (4, 1), True)    # lowered below what is cached: reload (clamp in key order)
    ts = c.tenant_stats("T")
    assert ts.max_persistent_fanout == 4 and ts.max_group_fanout == 1 and ts.entries == 2 and ts.hits == 5 and ts.misses == 5
    assert c.tenant_stats("nobody") is None
    # other tenants keep the defaults; get_batch caps per tenant, identical misses are ONE load
    del events[:]
    utopics = ["a/b", "a", "b/c/a", "c"]
    row, ids, _hit = c.get_batch(["T", "U"], [0, 1, 1, 0, 1, 1, 0], ["s/t", utopics[0], utopics[1], "x", utopics[2], utopics[3], "s/t"], now_ms=2)
    want_t = O.matched_routes_load("T", "s/t", live, 4, 1).routes()
    assert key_of(ids[row[0]:row[1]].tolist()) == want_t and key_of(ids[row[6]:row[7]].tolist()) == want_t
    for j, r in ((0, 1), (1, 2), (2, 4), (3, 5)):
        assert key_of(ids[row[r]:row[r + 1]].tolist()) == O.matched_routes_load("U", utopics[j], live, 3, 2).routes()
    want_ev = []
    for j in range(4):
        want_ev += [("U", utopics[j], typ, k, mx) for typ, k, mx in O.matched_routes_load("U", utopics[j], live, 3, 2).events]
    assert sorted((t.decode(), p.decode(), typ, eng.route_key(rid), mx) for t, p, typ, rid, mx in events) == sorted(want_ev)
    # the direct path (big requests skip the cache) caps too
    c2 = B.RouteCache(b, max_persistent_fanout=2, max_group_fanout=1, direct_batch_topics=4)
    row, ids, _hit = c2.get_batch(["T", "U"], [0, 1, 1, 0, 1], ["s/t", utopics[0], utopics[1], "x", utopics[2]], now_ms=2)
    assert key_of(ids[row[0]:row[1]].tolist()) == O.matched_routes_load("T", "s/t", live, 2, 1).routes()
    assert key_of(ids[row[1]:row[2]].tolist()) == O.matched_routes_load("U", utopics[0], live, 2, 1).routes()
    assert c2.stats().entries == 0
    c2.close()
    # get_async: the miss completes on the dispatcher thread with the capped row, the hit inline
    c.apply([(0, O.route_key("T", "s/t", O.FLAG_NORMAL, O.receiver_url(0, "mqtt-late", "d1")))])
    live.add(O.route_key("T", "s/t", O.FLAG_NORMAL, O.receiver_url(0, "mqtt-late", "d1")))
    done = threading.Event()
    got = {}

    def on_done(status, ids_, epoch):
        got["v"] = (status, ids_)
        done.set()

    del events[:]
    c.get_async("T", "s/t", on_done, now_ms=3)
    assert done.wait(10)
    want = O.matched_routes_load("T", "s/t", live, 4, 1)
    assert got["v"][0] == 0 and key_of(got["v"][1]) == want.routes()
    assert [(typ, eng.route_key(rid), mx) for _t, _p, typ, rid, mx in events] == want.events
    done.clear()
    c.get_async("T", "s/t", on_done, now_ms=3)
    assert done.is_set() and key_of(got["v"][1]) == want.routes()
    c.close()
    b.close()
    eng.close()


@pytest.mark.gpu
def test_routes_cap_abi_gpu():
    """bmq_routes_cap == MatchedRoutes over rows somebody else matched; after churn (ids no longer ranks) the caps still follow KEY order"""
    import bifromq_amd as B
    keys = sorted(_caps_case())
    eng = B.Engine(device=0).rebuild(keys)
    late = O.route_key("T", "#", O.FLAG_NORMAL, O.receiver_url(1, "inbox-0-sorts-first", "d0"))  # gets the LAST id, sorts in front
    eng.apply([(0, late)])
    live = set(keys) | {late}
    tn = ["T"]
    row, ids = eng.match_batch(tn, np.zeros(2, dtype=np.uint32), topics=["s/t", "x"])
    rows, counts, ev = eng.routes_cap(row, ids, 3, 2)
    for r, topic in enumerate(["s/t", "x"]):
        want = O.matched_routes_load("T", topic, live, 3, 2)
        got = rows[r]
        assert set(eng.route_keys(np.array(got, dtype=np.uint32))) == want.routes() and got == sorted(got)
        assert [(typ, eng.route_key(rid), mx) for typ, rr, rid, mx in ev if rr == r] == want.events
    assert late in set(eng.route_keys(np.array(rows[0], dtype=np.uint32)))  # admitted first although its id is the largest
    assert counts[0] == (3, 2) and counts[1] == (0xFFFFFFFF, 0xFFFFFFFF)  # row 1 is shorter than either cap: never classified
    eng.close()
