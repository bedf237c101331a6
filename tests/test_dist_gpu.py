"""GPU parity tests of the dist-direction match path (HIP kernels behind the C ABI) against the CPU oracle.

Bar: bit-exact -- for every (tenant, topic) the ascending list of route ids (ranks of the KV keys after a rebuild; stable
handles mapped back through their keys after applies) equals the
oracle's.  Known-answer tests are the reference's own (DWT/cache/TenantRouteMatcherTest.java:89-342,
DWT/DistQoS0Test.java:95-150), re-run through the engine.
"""
import os
import random

import numpy as np
import pytest

import bifromq_amd as B
from bifromq_amd.workload import unpack
from oracle import oracle as O
from oracle import semantic as S
from tests import util as U

pytestmark = pytest.mark.gpu

TENANT, OTHER = "tenantA", "tenantB"


def _normal(tenant, tf, broker, recv, deliverer):
    return B.route_key_from_mqtt(tenant, tf, O.receiver_url(broker, recv, deliverer))


@pytest.fixture(scope="module")
def eng():
    e = B.Engine(device=0)
    yield e
    e.close()


def _keys_of(e, rows):
    return [[e.route_key(r) for r in row] for row in rows]


# ---- the reference's known-answer tests, through the engine ---------------------------------------------------
def test_kat_no_tenant_data(eng):  # TenantRouteMatcherTest.java:89-110
    eng.rebuild([_normal(OTHER, "sensors/+/temp", 1, "receiverX", "delivererX")])
    m = B.TenantRouteMatcher(eng, TENANT).match_all(["sensors/device1/temp", "sensors/device1/humidity"], 10, 10)
    assert set(m) == {"sensors/device1/temp", "sensors/device1/humidity"}
    assert all(v.route_ids == [] for v in m.values())


def test_kat_multiple_topics(eng):  # :112-146
    temp = _normal(TENANT, "sensors/+/temp", 1, "receiverA", "delivererA")
    hum = _normal(TENANT, "sensors/+/humidity", 1, "receiverB", "delivererB")
    eng.rebuild([temp, hum])
    rows, ev = eng.match_all(TENANT, ["sensors/device1/temp", "sensors/device1/humidity", "sensors/device2/temp"], 10, 10)
    assert _keys_of(eng, rows) == [[temp], [hum], [temp]] and ev == []


def test_kat_reuse_cached_filter_matches(eng):  # :148-176
    a = _normal(TENANT, "devices/+/status", 1, "receiverA", "delivererA")
    b = _normal(TENANT, "devices/+/status", 2, "receiverB", "delivererB")
    eng.rebuild([a, b])
    rows, ev = eng.match_all(TENANT, ["devices/a/status", "devices/b/status"], 5, 5)
    assert _keys_of(eng, rows) == [sorted([a, b])] * 2 and ev == []


def test_kat_shared_subscription(eng):  # :178-205
    g = B.route_key_from_mqtt(TENANT, "$share/groupAlpha/alerts/+/+/temperature")
    eng.rebuild([g])
    m = B.TenantRouteMatcher(eng, TENANT).match_all(
        ["alerts/site1/device1/temperature", "alerts/site1/device2/temperature"], 10, 10)
    for v in m.values():
        assert v.routes(eng) == [(2, TENANT, "$share/groupAlpha/alerts/+/+/temperature", "groupAlpha")]
        assert v.group_fanout == 1 and v.persistent_fanout == 0


def test_kat_probe_then_seek(eng):  # :207-237
    keys = [_normal(TENANT, "invalid/%d" % i, 1, "noise%d" % i, "deliverer%d" % i) for i in range(21)]
    valid = _normal(TENANT, "metrics/+/cpu", 1, "receiverA", "delivererA")
    eng.rebuild(keys + [valid])
    rows, ev = eng.match_all(TENANT, ["metrics/server1/cpu"], 10, 10)
    assert _keys_of(eng, rows) == [[valid]] and ev == []


def test_kat_tenant_isolation(eng):  # :239-270
    a = _normal(TENANT, "devices/+/signal", 1, "receiverA", "delivererA")
    b = _normal(OTHER, "devices/+/signal", 1, "receiverB", "delivererB")
    eng.rebuild([a, b])
    assert _keys_of(eng, eng.match_all(TENANT, ["devices/a/signal"], 10, 10)[0]) == [[a]]
    assert _keys_of(eng, eng.match_all(OTHER, ["devices/a/signal"], 10, 10)[0]) == [[b]]
    # both tenants in ONE batch
    row, ids = eng.match_batch([TENANT, OTHER, "nobody"], [0, 1, 2, 1], ["devices/a/signal"] * 4)
    assert _keys_of(eng, U.csr_rows(row, ids)) == [[a], [b], [], [b]]


def test_kat_persistent_fanout_throttling(eng):  # :272-303
    first = _normal(TENANT, "alarms/+/critical", 1, "receiverA", "delivererA")
    second = _normal(TENANT, "alarms/+/critical", 1, "receiverB", "delivererB")
    eng.rebuild([first, second])
    kv = O.KV([first, second])
    exp = kv.match_all(TENANT, ["alarms/device1/critical"], 1, 10)
    rows, ev = eng.match_all(TENANT, ["alarms/device1/critical"], 1, 10)
    assert rows == exp.per_topic() and ev == exp.events
    assert len(rows[0]) == 1 and len(ev) == 1 and (ev[0][0], ev[0][1], ev[0][3]) == (0, 0, 1)
    m = B.TenantRouteMatcher(eng, TENANT).match_all(["alarms/device1/critical"], 1, 10)["alarms/device1/critical"]
    assert m.persistent_fanout == 1 and len(m.throttled) == 1


def test_kat_group_fanout_throttling(eng):  # :305-342
    first = B.route_key_from_mqtt(TENANT, "$share/groupA/jobs/+/progress")
    second = B.route_key_from_mqtt(TENANT, "$share/groupB/jobs/+/progress")
    eng.rebuild([first, second])
    rows, ev = eng.match_all(TENANT, ["jobs/job1/progress"], 10, 1)
    assert _keys_of(eng, rows) == [[second]]  # "second comes before first ... by bucketing key" (:339-340)
    assert len(ev) == 1 and ev[0][0] == 1 and ev[0][3] == 1 and eng.route_key(ev[0][2]) == first


def test_caps_follow_key_order_after_apply(eng):
    """MatchedRoutes caps are first-come in KV KEY order (DW/cache/MatchedRoutes.java:87-141).  Routes subscribed through
    bmq_routes_apply carry later ids than the routes of the last rebuild although their keys may sort FIRST: bmq_match_all must
    order a row by key bytes before it applies a cap.  Expected = the oracle's matchAll over the sorted key set."""
    rnd = random.Random(12)
    base = [_normal(TENANT, "alarms/+/critical", 1, "recv%02d" % i, "d%d" % i) for i in range(10, 20)] + \
           [B.route_key_from_mqtt(TENANT, "$share/g%02d/alarms/#" % i) for i in range(10, 16)]
    late = [_normal(TENANT, "alarms/+/critical", 1, "recv%02d" % i, "d%d" % i) for i in range(0, 10)] + \
           [B.route_key_from_mqtt(TENANT, "$share/g%02d/alarms/#" % i) for i in range(0, 10)] + \
           [_normal(TENANT, "alarms/#", 0, "transient%d" % i, "d") for i in range(5)]
    eng.rebuild(base)
    rnd.shuffle(late)
    eng.apply([(0, k) for k in late])
    keys = sorted(base + late)
    kv = O.KV(keys)
    topic = "alarms/device1/critical"
    for max_pf, max_gf in ((3, 2), (15, 100), (100, 4), (1, 1), (100, 100)):
        exp = kv.match_all(TENANT, [topic], max_pf, max_gf)
        rows, ev = eng.match_all(TENANT, [topic], max_pf, max_gf)
        assert sorted(_keys_of(eng, rows)[0]) == sorted(keys[r] for r in exp.per_topic()[0]), (max_pf, max_gf)
        # the throttle events name the same routes in the same (key) order, with the same type and maximum
        assert [(t, ti, eng.route_key(rid), mx) for t, ti, rid, mx in ev] == [(t, ti, keys[rid], mx) for t, ti, rid, mx in exp.events]
    # the accepted persistent routes are the first three in KEY order (the bucket byte -- a hash of the receiver url -- leads the
    # key's tail, so that is neither receiver-name order nor id order), and at least one of them was added by the apply
    rows, ev = eng.match_all(TENANT, [topic], 3, 100)
    is_pers = lambda k: O.parse_route_key(k)[0] == 1 and O.parse_route_key(k)[3].startswith("1\0")
    assert sorted(k for k in _keys_of(eng, rows)[0] if is_pers(k)) == [k for k in keys if is_pers(k)][:3]
    assert any(k in late for k in [k for k in keys if is_pers(k)][:3])


def test_dist_qos0_vectors(eng):  # DWT/DistQoS0Test.java:95-150
    keys = [_normal(TENANT, "/你好/hello/😄", 0, "inbox1", "batch1"), _normal(TENANT, "/#", 0, "inbox1", "batch1"),
            _normal(TENANT, "/#", 0, "inbox2", "batch2"), _normal(TENANT, "#", 0, "inbox3", "batch3"),
            _normal(TENANT, "$sys/#", 0, "inbox4", "b")]
    eng.rebuild(keys)
    topics = ["/你好/hello/😄", "$sys/bifromq/user/event/abc", "/", ""]
    rows = eng.match_tenant(TENANT, topics)
    assert [len(r) for r in rows] == [4, 1, 3, 3]  # "/#" also matches "" (one empty level, '#' matches the parent)
    assert _keys_of(eng, rows)[1] == [keys[4]]
    assert rows == [sorted(r) for r in O.KV(keys).match_all(TENANT, topics).per_topic()]


# ---- randomised parity: engine == structural oracle == semantic brute force -------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_parity_small(eng, seed):
    rnd = random.Random(seed)
    tenants = ["tA", "tB", "租户"]
    keys = sorted({U.rand_route_key(rnd, rnd.choice(tenants), U.rand_filter(rnd, 5), i) for i in range(4000)})
    eng.rebuild(keys)
    kv = O.KV(keys)
    topics = [U.rand_topic(rnd, 6) for _ in range(3000)] + ["", "/", "//", "a", "$sys", "$sys/a", "+", "#", "a/#"]
    tt = [rnd.randrange(len(tenants) + 1) for _ in topics]  # index 3 = tenant without routes
    tnames = tenants + ["ghost"]
    row, ids = eng.match_batch(tnames, tt, topics)
    got = U.csr_rows(row, ids)
    assert got == U.semantic_rows(kv, tnames, tt, topics)
    # the structural restatement of TenantRouteMatcher.matchAll (whole-batch mode) agrees, except where the
    # reference loses routes to quirk (ii) -- this alphabet is rich in empty levels on purpose
    U.assert_rows_equal_modulo_quirk_ii(keys, tnames, tt, U.oracle_rows(kv, tnames, tt, topics), got)
    for i in range(0, len(topics), 7):  # semantic cross-check (independent python matcher)
        exp = [r for r, k in enumerate(keys)
               if (d := O.parse_route_key(k))[1] == tnames[tt[i]]
               and S.matches(topics[i], d[2].split("/", 2)[2] if d[0] != 1 else d[2])]
        assert got[i] == exp, (topics[i], tnames[tt[i]])
    st = eng.stats()
    assert st.n_topics == len(topics) and st.n_match == len(ids)
    # roofline accounting: the engine's N_visit equals the oracle's count on the same data
    packed = O.pack(topics)
    assert st.n_visit == int(kv.count_visits(tnames, np.array(tt, dtype=np.uint32), packed).sum())


def test_generated_workload_parity(eng):
    """C1-shaped (literal) and C2/C3-shaped (wildcards, multi-tenant) workloads at oracle-friendly sizes."""
    for seed, n_tenants, per, mode, n_topics in [(0xB1F20001, 1, 10000, 0, 20000), (0xB1F20002, 1, 20000, 1, 20000),
                                                 (0xB1F20003, 20, 2000, 1, 30000)]:
        w = B.Workload(seed, n_tenants, per, mode)
        eng.rebuild(packed=w.keys_packed())
        kv = O.KV(packed=w.keys_packed())
        data, off, tt = w.topics(seed + 77, n_topics)
        tn = w.tenants()
        row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
        topics = [t.decode() for t in unpack(data, off)]
        got = U.csr_rows(row, ids)
        assert eng.stats().n_visit == int(kv.count_visits(tn, tt, (data, off)).sum())
        # (1) the reference's production call pattern -- one matchAll(singleton(topic)) per topic -- on EVERY topic;
        # rows may differ only where the reference itself loses routes (quirk ii / the trailing-'/' livelock,
        # see oracle/bmq_oracle.cpp); (2) whole-batch matchAll on a slice; (3) authoritative semantic rows on a sample
        res, _ = kv.match_singletons(tn, tt, (data, off), threads=U.host_threads())
        differ = U.assert_rows_equal_modulo_quirk_ii(w.keys(), tn, tt, [sorted(r) for r in res.per_topic()], got)
        # no tolerance: EVERY row the restatement differs in equals the semantic oracle (the reference's own brute-force TopicMatcher)
        assert [got[i] for i in differ] == U.semantic_rows(kv, tn, [tt[i] for i in differ], [topics[i] for i in differ])
        m = 3000
        U.assert_rows_equal_modulo_quirk_ii(w.keys(), tn, tt[:m], U.oracle_rows(kv, tn, tt[:m], topics[:m]), got[:m])
        idx = list(range(0, n_topics, 40))
        assert [got[i] for i in idx] == U.semantic_rows(kv, tn, [tt[i] for i in idx], [topics[i] for i in idx])


def test_grouped_batch_equals_ungrouped(eng):
    """k_walk has two instantiations: a batch ordered by tenant (the DistPacks of a BatchDistRequest: what bench.py times) is walked one
    tenant at a time with the tenant's region in scalar registers, waves that straddle up to 4 tenants walk them one after the other; a
    batch in any other order is run again through the MIXED instantiation (ST_WANT_MIXED).  The same publishes in both orders must give
    the same rows -- the ungrouped order is what the oracle tests above and below use -- also for tenants with a handful of publishes
    (several tenants per wave) and for small batches (16 / 4 topics per wave)."""
    w = B.Workload(0xB1F20007, 300, 2000, 1)
    eng.rebuild(packed=w.keys_packed())
    kv = O.KV(packed=w.keys_packed())
    tn = w.tenants()
    for n in (200_000, 20_000, 3_000):  # 64 / 16 / 4 topics per wave
        data, off, tt = w.topics(0xB1F20007 + n, n)
        row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
        nv = eng.stats().n_visit
        order = np.argsort(tt, kind="stable")
        gdata, goff = U.sub_packed(data, off, order)
        grow, gids = eng.match_batch(tn, tt[order], packed_topics=(gdata, goff))
        assert eng.stats().n_visit == nv == int(kv.count_visits(tn, tt, (data, off)).sum())
        a_rp, a = U.csr_select(row, ids, order)
        assert np.array_equal(a_rp, grow.astype(np.int64)) and np.array_equal(a, gids)
        # ... and a slice of the grouped batch against the semantic oracle directly
        sel = np.arange(0, n, max(1, n // 2000))
        sd, so = U.sub_packed(gdata, goff, sel)
        res, _ = kv.match_semantic_batch(tn, tt[order][sel], (sd, so), threads=U.host_threads())
        b_rp, b = U.csr_select(grow, gids, sel)
        assert np.array_equal(b_rp, res.row_ptr.astype(np.int64)) and np.array_equal(b, res.routes)


def test_expand_gives_heavy_blocks_to_four_waves_each(eng):
    """k_expand (round 6): a batch ORDERED by (tenant, topic) -- what BatchDistRequest carries -- puts the rows under a hot prefix side by side;
    the 64-row blocks that hold more than ~2 x the ranges / ids of the mean block are listed by k_walk and expanded by four waves each (rows
    0-15 by the block's own wave, three helper waves in front of the grid for the rest).  The ordered batch must give the rows of the batch
    as generated -- which the oracle tests check --, with repeats and without, blocks must in fact have been split, and an engine's first
    large batch (absolute thresholds) must agree with its later ones (thresholds from the batches before)."""
    w = B.Workload(0xB1F20009, 200, 5000, 1)
    e2 = B.Engine(device=0)  # a fresh engine: its first large batch has no history to take thresholds from
    try:
        for engine in (eng, e2):
            engine.rebuild(packed=w.keys_packed())
        kv = O.KV(packed=w.keys_packed())
        tn = w.tenants()
        n = 300_000
        data, off, tt = w.topics(0xB1F20009 + 5, n)
        row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
        assert eng.stats().n_visit == int(kv.count_visits(tn, tt, (data, off)).sum())
        topics = unpack(data, off)
        order = np.asarray(sorted(range(n), key=lambda i: (int(tt[i]), topics[i])), dtype=np.int64)
        head = np.ones(n, dtype=bool)
        head[1:] = [tt[order[i]] != tt[order[i - 1]] or topics[order[i]] != topics[order[i - 1]] for i in range(1, n)]
        for name, sel in (("ordered, repeats kept", order), ("ordered, every topic once", order[head])):
            sd, so = U.sub_packed(data, off, sel)
            want_rp, want = U.csr_select(row, ids, sel)
            splits = []
            for engine in (e2, eng, eng):
                grow, gids = engine.match_batch(tn, tt[sel], packed_topics=(sd, so))
                assert np.array_equal(want_rp, grow.astype(np.int64)) and np.array_equal(want, gids), name
                splits.append(engine.stats().n_split_blocks)
            assert min(splits) > 0, (name, splits)  # the hot prefixes' blocks were split, with either kind of threshold
            per_block = np.add.reduceat(np.diff(want_rp), np.arange(0, len(sel), 64))
            assert per_block.max() > 3 * per_block.mean(), name  # (what makes this batch a test of the path)
        # ... and a slice of the ordered batch against the semantic oracle directly
        pick = order[:: max(1, n // 3000)]
        sd, so = U.sub_packed(data, off, pick)
        res, _ = kv.match_semantic_batch(tn, tt[pick], (sd, so), threads=U.host_threads())
        b_rp, b = U.csr_select(row, ids, pick)
        assert np.array_equal(b_rp, res.row_ptr.astype(np.int64)) and np.array_equal(b, res.routes)
    finally:
        e2.close()


def test_in_batch_dedup_changes_nothing_but_the_work():
    """Identical (tenant, topic) rows of a batch are walked once (k_dedup / k_fill: matchAll takes a Set<String>,
    TenantRouteMatcher.java:67-78); every row keeps its own row.  An engine that de-duplicates every batch and one that never does must
    return the same CSR and the same statistics -- N_visit / ranges / bytes are counted per ROW (the roofline accounting of SURVEY.md 8d
    treats them as properties of the data) -- on batches full of repeats: hot topics, topics of unknown tenants, empty topics, topics
    deeper than FAST_LEVELS (k_walk_slow), in both batch orders and with the smallest LDS lists; and both equal the semantic oracle."""
    rnd = random.Random(11)
    tenants = ["tA", "tB", "租户", "t4", "t5", "t6"]
    alphabet = ["a", "b", "c", "", "$sys", "dev", "x" * 17]
    keys = sorted({U.rand_route_key(rnd, rnd.choice(tenants), U.rand_filter(rnd, 5, alphabet), i) for i in range(6000)} |
                  {U.rand_route_key(rnd, "tA", "/".join(["a"] * 20 + ["#"]), 900001), U.rand_route_key(rnd, "tB", "/".join(["+"] * 18), 900002)})
    kv = O.KV(keys)
    pool = [U.rand_topic(rnd, 6, alphabet) for _ in range(700)] + ["", "/", "a", "/".join(["a"] * 21), "/".join(["a"] * 18)]
    topics = [pool[min(int(rnd.paretovariate(1.1)) - 1, len(pool) - 1)] for _ in range(20000)]  # Zipf-like repeats
    tnames = tenants + ["ghost"]
    tt = [rnd.randrange(len(tnames)) for _ in topics]
    exp = U.semantic_rows(kv, tnames, tt, topics)
    for order in ("as is", "grouped"):
        if order == "grouped":
            o = sorted(range(len(topics)), key=lambda i: tt[i])
            topics, tt, exp = [topics[i] for i in o], [tt[i] for i in o], [exp[i] for i in o]
        stats = []
        for dd, geom in ((1, {}), (0xFFFFFFFF, dict(region_slack=1)), (1, dict(wave_queue_cap=128, wave_pair_cap=128, slow_scratch_mb=1))):  # (region_slack = 1: the regions of rounds 2-5, load factor 0.4)
            e = B.Engine(device=0, dedup_min_topics=dd, **geom).rebuild(keys)
            row, ids = e.match_batch(tnames, tt, topics)
            assert U.csr_rows(row, ids) == exp
            st = e.stats()
            stats.append((st.n_visit, st.n_match, st.n_ranges, st.topic_bytes, st.n_slow_topics > 0))
            e.close()
        assert stats[0] == stats[1] == stats[2]
        assert stats[0][0] == int(kv.count_visits(tnames, np.array(tt, dtype=np.uint32), O.pack(topics)).sum())


def _run_heads(tt, topics):
    """rows that differ from the row before them (tenant index or bytes): what the neighbour compare keeps"""
    return sum(1 for i in range(len(topics)) if i == 0 or tt[i] != tt[i - 1] or topics[i] != topics[i - 1])


def test_dedup_of_an_ordered_batch_by_neighbour_compare():
    """bmq_config.dedup_sorted (bmq_dedup_adj_kernels.h): BatchDistRequest is "sorted by tenantId and topic" (DistWorkerCoProc.proto:75-83), so
    equal rows are neighbours -- the run heads are copied into a dense batch, the walk kernels run on that, every row takes its head's result.
    The CSR and the statistics must equal those of an engine that never de-duplicates, and the semantic oracle, on batches full of repeats
    (hot topics, unknown tenants, empty topics, topics deeper than FAST_LEVELS, long topics): ordered -- then n_walked is the number of DISTINCT
    (tenant, topic) pairs --, grouped by tenant only and not ordered at all (only speed depends on the order: n_walked = the rows that differ
    from their predecessor); with 4 / 16 / 64 rows per wave; with the smallest LDS lists; through the device-resident entry point's growth
    path when the dense batch's bytes outgrow the first guess (ST_NEED_ADJ)."""
    rnd = random.Random(23)
    tenants = ["tA", "tB", "租户", "t4", "t5", "t6"]
    alphabet = ["a", "b", "c", "", "$sys", "dev", "x" * 17]
    keys = sorted({U.rand_route_key(rnd, rnd.choice(tenants), U.rand_filter(rnd, 5, alphabet), i) for i in range(6000)} |
                  {U.rand_route_key(rnd, "tA", "/".join(["a"] * 20 + ["#"]), 900001), U.rand_route_key(rnd, "tB", "/".join(["+"] * 18), 900002),
                   U.rand_route_key(rnd, "tA", "long/#", 900003)})
    kv = O.KV(keys)
    pool = [U.rand_topic(rnd, 6, alphabet) for _ in range(700)] + ["", "/", "a", "/".join(["a"] * 21), "/".join(["a"] * 18)]
    pool += ["long/" + "y" * rnd.randrange(100, 900) for _ in range(40)]  # blocks whose bytes do not fit the LDS image; > 48 bytes per row on average
    tnames = tenants + ["ghost"]
    plain = B.Engine(device=0).rebuild(keys)
    for n, kind in ((37, "zipf"), (900, "long"), (20000, "zipf"), (20000, "long"), (140000, "zipf")):
        topics = [pool[min(int(rnd.paretovariate(1.1)) - 1, len(pool) - 1)] for _ in range(n)]  # Zipf-like repeats
        if kind == "long":  # long topics only: the first guess of the dense buffer is too small (ST_NEED_ADJ); with 16 rows per wave (n = 20000) a
            topics = [rnd.choice(pool[-40:]) for _ in range(n)]  # block's bytes outgrow the LDS image too: the byte-copy path of k_dd_adj_scatter
        tt = [rnd.randrange(len(tnames)) for _ in topics]
        for order in ("ordered", "grouped", "as is"):
            if order == "ordered":
                o = sorted(range(n), key=lambda i: (tt[i], topics[i].encode()))
            elif order == "grouped":
                o = sorted(range(n), key=lambda i: tt[i])
            else:
                o = list(range(n))
            ts, tts = [topics[i] for i in o], [tt[i] for i in o]
            row0, ids0 = plain.match_batch(tnames, tts, ts)
            st0 = plain.stats()
            assert st0.n_walked == n
            if n <= 20000 and order == "ordered":  # (the engine that never de-duplicates is itself checked against the oracle)
                assert U.csr_rows(row0, ids0) == U.semantic_rows(kv, tnames, tts, ts)
            for geom in ({}, dict(wave_queue_cap=128, wave_pair_cap=128, slow_scratch_mb=1)):
                if geom and (n != 20000 or kind != "zipf"):
                    continue
                e = B.Engine(device=0, dedup_min_topics=1, dedup_sorted=True, **geom).rebuild(keys)
                for rep in range(2):  # (the second batch runs on buffers the first one sized)
                    row, ids = e.match_batch(tnames, tts, ts)
                    assert np.array_equal(row, row0) and np.array_equal(ids, ids0), (n, order, rep)
                    st = e.stats()
                    assert (st.n_visit, st.n_match, st.n_ranges, st.topic_bytes, st.n_slow_topics > 0) == \
                           (st0.n_visit, st0.n_match, st0.n_ranges, st0.topic_bytes, st0.n_slow_topics > 0), (n, order, rep)
                    assert st.n_walked == _run_heads(tts, ts), (n, order)
                    if order == "ordered":
                        assert st.n_walked == len(set(zip(tts, ts)))
                e.close()
    plain.close()


# ---- the rare paths: LDS overflow -> per-lane DFS, deep topics, interleaved ranges -> fix-up sort ------------------
def test_slow_path_equals_fast_path():
    rnd = random.Random(5)
    alphabet = ["a", "b", "c"]
    keys = sorted({U.rand_route_key(rnd, "t", U.rand_filter(rnd, 6, alphabet), i) for i in range(6000)})
    topics = [U.rand_topic(rnd, 6, alphabet) for _ in range(2000)]
    topics += ["/".join(rnd.choice(alphabet) for _ in range(n)) for n in (16, 17, 18, 40, 300)]  # deep topics
    deep = sorted({U.rand_route_key(rnd, "t", "/".join(["a"] * n + ["#"]), 900000 + n) for n in (15, 16, 17, 30)} |
                  {U.rand_route_key(rnd, "t", "/".join(["+"] * 17), 800000), U.rand_route_key(rnd, "t", "/".join(["a"] * 40), 800001)})
    topics += ["/".join(["a"] * n) for n in (15, 16, 17, 18, 30, 31, 40, 41)]
    keys = sorted(set(keys) | set(deep))
    kv = O.KV(keys)
    exp = U.semantic_rows(kv, ["t"], [0] * len(topics), topics)
    normal = B.Engine(device=0).rebuild(keys)
    assert normal.match_tenant("t", topics) == exp
    assert normal.stats().n_slow_topics >= 10  # the deep ones
    nv = int(kv.count_visits(["t"], np.zeros(len(topics), dtype=np.uint32), O.pack(topics)).sum())
    assert normal.stats().n_visit == nv
    n_deep = normal.stats().n_slow_topics
    # smallest LDS lists: the work stack and the range buffer overflow into their global spill chains, results and
    # counters are unchanged and nothing but the deep topics takes the slow path
    tiny = B.Engine(device=0, wave_queue_cap=128, wave_pair_cap=128, slow_scratch_mb=1).rebuild(keys)
    assert tiny.match_tenant("t", topics) == exp
    assert tiny.stats().n_slow_topics == n_deep
    assert tiny.stats().n_visit == nv


def test_repair_kernels_come_and_go():
    """k_walk_slow / k_sort_rows are launched only while batches need them: a batch that needs one and did not have it is completed
    (re-run / sorted) by the finish step, 32 batches without need switch the kernel off again -- results identical throughout."""
    rnd = random.Random(77)
    alphabet = ["a", "b"]
    keys = sorted({U.rand_route_key(rnd, "t", U.rand_filter(rnd, 5, alphabet), i) for i in range(3000)} |
                  {U.rand_route_key(rnd, "t", "/".join(["a"] * 20 + ["#"]), 900001), U.rand_route_key(rnd, "t", "/".join(["+"] * 18), 900002)})
    kv = O.KV(keys)
    shallow = [U.rand_topic(rnd, 5, alphabet) for _ in range(300)]
    deep = shallow[:50] + ["/".join(["a"] * n) for n in (17, 18, 21, 25)]
    exp_shallow = U.semantic_rows(kv, ["t"], [0] * len(shallow), shallow)
    exp_deep = U.semantic_rows(kv, ["t"], [0] * len(deep), deep)
    eng = B.Engine(device=0).rebuild(keys)
    for cycle in range(2):
        assert eng.match_tenant("t", deep) == exp_deep          # first batch of the engine / after the kernels were dropped
        assert eng.stats().n_slow_topics == 4
        assert eng.match_tenant("t", deep) == exp_deep          # with the slow kernel in the pipeline
        for _ in range(40):                                     # > REPAIR_IDLE_BATCHES quiet batches
            assert eng.match_tenant("t", shallow) == exp_shallow
            assert eng.stats().n_slow_topics == 0
    # rows that need the fix-up sort: > 32 ranges per row, ranges not ascending (the case of test_many_ranges_per_row)
    keys2 = sorted({U.rand_route_key(rnd, "t", f, i) for i, f in enumerate(
        ["/".join(rnd.choice(["+", "x"]) for _ in range(7)) for _ in range(400)] + ["x/#", "+/#", "#"])})
    kv2 = O.KV(keys2)
    eng2 = B.Engine(device=0).rebuild(keys2)
    t7 = ["/".join(["x"] * 7), "x/x", "y"]
    exp7 = U.semantic_rows(kv2, ["t"], [0] * len(t7), t7)
    for cycle in range(2):
        assert eng2.match_tenant("t", t7) == exp7
        assert eng2.match_tenant("t", t7) == exp7
        for _ in range(40):
            assert eng2.match_tenant("t", t7[1:]) == exp7[1:]


def test_interleaved_key_ranges(eng):
    # keys of filter "x" (bucket byte b) interleave with keys of "x/..." whose next level
    # is empty, so the ids of ONE filter are not a contiguous rank range (indirect ranges in the index).  No single
    # topic can match both "x" and a deeper "x//..." filter, so rows still come out ascending.
    keys = [_normal("t", "x", 0, "r%d" % i, "d") for i in range(300)] + \
           [_normal("t", "x/#", 0, "h%d" % i, "d") for i in range(50)] + \
           [_normal("t", "x//#", 0, "e%d" % i, "d") for i in range(200)] + \
           [_normal("t", "x//" + c + "/#", 0, "q%d" % i, "d") for i in range(40) for c in "aZ0"]
    keys = sorted(set(keys))
    eng.rebuild(keys)
    kv = O.KV(keys)
    topics = ["x", "x/", "x//a", "x//Z/k", "x/y"]
    got = eng.match_tenant("t", topics)
    exp = [kv.match_bruteforce("t", [t]).per_topic()[0] for t in topics]
    assert got == exp
    assert all(r == sorted(r) for r in got)
    xs = eng.find("t", "x")
    assert len(xs) == 300 and xs != list(range(xs[0], xs[0] + 300))  # "x" really is a non-contiguous id set


def test_big_fanout_and_many_ranges(eng):
    # one topic matching > 32 distinct filters (range-list sort skipped -> row sort) and a 5000-receiver filter
    keys = [_normal("t", "big/+", 0, "r%d" % i, "d%d" % (i % 5)) for i in range(5000)]
    lv = ["a", "+"]
    import itertools
    for combo in itertools.product(lv, repeat=6):
        keys.append(_normal("t", "/".join(combo), 0, "c", "d"))
        keys.append(_normal("t", "/".join(combo[:5]) + "/#", 1, "h", "d"))
    keys = sorted(set(keys))
    eng.rebuild(keys)
    kv = O.KV(keys)
    topics = ["big/1", "a/a/a/a/a/a", "a/a/a/a/a/b", "big/2"] + ["a/a/a/a/a/a"] * 70
    exp = [kv.match_bruteforce("t", [t]).per_topic()[0] for t in topics[:4]]
    got = eng.match_tenant("t", topics)
    assert got[:4] == exp and all(g == exp[1] for g in got[4:])
    assert len(got[0]) == 5000 and len(got[1]) == 64 + 32
    assert eng.stats().n_sorted_rows >= 1  # > 32 ranges per row: range ordering skipped, out-of-order rows sorted by k_sort_rows


def test_apply_then_match(eng):
    w = B.Workload(42, 4, 1500, 1)
    keys = w.keys()
    eng.rebuild(keys)
    rnd = random.Random(9)
    live = set(keys)
    tn = w.tenants()
    data, off, tt = w.topics(5, 4000)
    topics = [t.decode() for t in unpack(data, off)]
    for step in range(3):
        ops = []
        for k in rnd.sample(sorted(live), 300):
            ops.append((1, k))
            live.discard(k)
        for i in range(300):
            k = U.rand_route_key(rnd, rnd.choice(tn), rnd.choice(topics).replace("l2_", "+/x")[:60] if i % 3 else "#", 10**6 + step * 1000 + i)
            ops.append((0, k))
            live.add(k)
        eng.apply(ops)
        keys_now = sorted(live)
        kv = O.KV(keys_now)
        assert eng.info().n_routes == len(live)
        row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
        d = np.diff(ids.astype(np.int64))
        d[(row[1:-1][row[1:-1] < len(ids)] - 1)[row[1:-1][row[1:-1] < len(ids)] > 0]] = 1
        assert (d > 0).all()  # rows stay ascending although new routes carry late ids (range ordering / k_sort_rows)
        assert U.rows_as_ranks(eng, row, ids, keys_now) == U.semantic_rows(kv, tn, tt, topics)
    # bmq_compact: the index re-built from its own live routes -- ids are ranks again, garbage gone, same answers
    before = eng.info()
    assert before.garbage_bytes > 0 or before.next_route_id > before.n_routes
    eng.compact()
    after = eng.info()
    assert after.generation == before.generation + 1 and after.n_routes == len(live) == after.next_route_id and after.garbage_bytes == 0
    row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
    assert U.csr_rows(row, ids) == U.semantic_rows(kv, tn, tt, topics)


def test_apply_creates_and_removes_tenants(eng):
    """bmq_routes_apply touching a subset of tenants: in-place inserts, region growth (the tenant is moved into a larger
    region by k_b_rehash), a tenant appearing through apply and a tenant losing its last route; ids are stable handles."""
    t1 = [_normal("t1", "a/%d" % i, 0, "r%d" % i, "d") for i in range(50)]
    t2 = [_normal("t2", "b/+", 0, "r%d" % i, "d") for i in range(5)]
    eng.rebuild(t1 + t2)
    live = set(t1 + t2)
    steps = [
        [(0, _normal("t1", "a/%d/x" % i, 0, "n%d" % i, "d")) for i in range(3)],                    # small, in place
        [(0, _normal("t0", "a/#", 0, "z", "d")), (0, _normal("t3", "+/1", 1, "p", "d"))],           # new tenants (before and after)
        [(1, k) for k in t2],                                                                       # t2 disappears
        [(0, _normal("t1", "g/%d/+/#" % i, 0, "g%d" % i, "d")) for i in range(400)],                # t1 outgrows its region
        [(1, _normal("t1", "a/7", 0, "r7", "d")), (0, _normal("t2", "b/#", 0, "back", "d"))],       # delete + tenant returns
    ]
    topics = ["a/1", "a/7", "a/1/x", "b/q", "g/3/k/m", "zzz/1"]
    for ops in steps:
        eng.apply(ops)
        for o, k in ops:
            (live.discard if o else live.add)(k)
        keys = sorted(live)
        info = eng.info()
        assert info.n_routes == len(keys) and info.n_tenants == len({O.parse_route_key(k)[1] for k in keys})
        got_keys = [k for k in eng.route_keys(list(range(int(info.next_route_id)))) if k]
        assert sorted(got_keys) == keys  # the key store holds exactly the live routes
        kv = O.KV(keys)
        tn = ["t0", "t1", "t2", "t3", "ghost"]
        tt = [i % 5 for i in range(len(topics) * 5)]
        tp = [topics[i // 5] for i in range(len(topics) * 5)]
        row, ids = eng.match_batch(tn, tt, tp)
        assert U.rows_as_ranks(eng, row, ids, keys) == U.semantic_rows(kv, tn, tt, tp)


def test_ops_of_one_batch_apply_in_order_also_when_the_filter_is_new(eng):
    """One batch = one ordered stream (ISubscriptionCache.refresh applies mutations in arrival order): subscribe to a filter nobody had,
    unsubscribe again -- thousands of such pairs, so that the deletes' lanes would run beside the puts' -- leaves nothing; delete / put /
    delete of an existing route leaves it deleted; put / delete / put leaves it there.  (Round 5: the locate stage took puts and deletes
    in one pass, and a delete whose filter node was being created by a put of the same batch was dropped as "no such filter".)"""
    base = [_normal("t1", "a/%d/+" % i, 0, "r%d" % i, "d") for i in range(200)]
    eng.rebuild(base)
    n = 20000
    fresh = [_normal("t%d" % (i % 3), "fresh/%d/x%d/#" % (i, i % 7), 0, "f%d" % i, "d") for i in range(n)]   # t0, t2: tenants born in the batch too
    stay = [_normal("t1", "stay/%d" % i, 0, "s%d" % i, "d") for i in range(500)]
    ops = [(0, k) for k in fresh] + [(1, base[3]), (0, base[3]), (1, base[3])] + [(0, k) for k in stay] + [(1, k) for k in fresh] + \
          [(0, stay[5]), (1, stay[5]), (0, stay[5])]
    eng.apply(ops)
    live = sorted(set(base) - {base[3]} | set(stay))
    info = eng.info()
    assert info.n_routes == len(live)
    assert sorted(k for k in eng.route_keys(np.arange(int(info.next_route_id), dtype=np.uint32)) if k) == live
    row, ids = eng.match_batch(["t0", "t1", "t2"], [0, 1, 2, 1], ["fresh/0/x0/y", "stay/5", "fresh/2/x2", "a/3/q"])
    assert [len(r) for r in U.csr_rows(row, ids)] == [0, 1, 0, 0]


def test_edge_shapes(eng):
    """maximum-size and degenerate inputs: a 65535-byte topic of 32768 empty levels, a 60000-byte single level, batches of
    one topic, only-slash topics, levels longer than the 16-byte inline prefix that differ only in their tail."""
    long_a, long_b = "p" * 40 + "A", "p" * 40 + "B"  # same first 16 bytes, same length: the pool compare must decide
    keys = [_normal("t", "#", 0, "all", "d"), _normal("t", "/#", 0, "slash", "d"), _normal("t", long_a + "/+", 0, "la", "d"),
            _normal("t", long_b + "/+", 0, "lb", "d"), _normal("t", "x" * 60000, 0, "huge", "d"),
            _normal("t", "/".join(["+"] * 9), 0, "nine", "d"), _normal("t", "////", 0, "four", "d")]
    keys = sorted(keys)
    eng.rebuild(keys)
    kv = O.KV(keys)
    topics = ["/" * 65535, "x" * 60000, "x" * 59999, long_a + "/1", long_b + "/1", "p" * 40 + "C/1", "////", "///", "/////",
              "a/b/c/d/e/f/g/h/i", "a"]
    exp = [kv.match_bruteforce("t", [t]).per_topic()[0] for t in topics]
    assert eng.match_tenant("t", topics) == exp
    assert eng.stats().n_slow_topics >= 1  # the 32768-level topic
    for i, t in enumerate(topics):  # batches of one
        assert eng.match_tenant("t", [t]) == [exp[i]]
    row, ids = eng.match_batch([], [0, 0], ["a", "b"])  # no tenant table at all: every row empty
    assert row.tolist() == [0, 0, 0] and len(ids) == 0


def test_concurrent_callers(eng):
    """The reference calls matchAll from a ForkJoinPool (DW/DistWorkerCoProcFactory.java:74-88) while the apply thread
    mutates routes: concurrent host-buffer calls on one engine are serialised inside the library (ctypes drops the GIL)."""
    import threading
    w = B.Workload(7, 3, 800, 1)
    keys = w.keys()
    eng.rebuild(keys)
    tn = w.tenants()
    data, off, tt = w.topics(3, 600)
    topics = [t.decode() for t in unpack(data, off)]
    kv = O.KV(keys)
    exp = U.semantic_rows(kv, tn, tt, topics)
    extra = [_normal(tn[0], "zz/%d" % i, 0, "x%d" % i, "d") for i in range(40)]  # never match the topics above
    errors = []

    def matcher():
        try:
            for _ in range(15):
                row, ids = eng.match_batch(tn, tt, topics)
                got = U.csr_rows(row, ids)
                assert got == exp  # ids are stable handles: routes coming and going elsewhere do not move them
                for r, k in zip(got[0][:3], eng.route_keys(got[0][:3])):
                    assert k == keys[r]  # ... and resolve to the same key whatever the mutator is doing
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    def mutator():
        try:
            for i in range(10):
                eng.apply([(0, k) for k in extra[i * 4:(i + 1) * 4]])
                eng.apply([(1, k) for k in extra[i * 4:(i + 1) * 4]])
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    th = [threading.Thread(target=matcher) for _ in range(4)] + [threading.Thread(target=mutator)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    row, ids = eng.match_batch(tn, tt, topics)
    assert U.csr_rows(row, ids) == exp


def test_output_capacity_protocol(eng):
    import ctypes as C
    from bifromq_amd import _lib
    keys = [_normal("t", "a/+", 0, "r%d" % i, "d") for i in range(100)]
    eng.rebuild(keys)
    tdata, toff = B.pack(["t"])
    pdata, poff = B.pack(["a/b"] * 10)
    tt = np.zeros(10, dtype=np.uint32)
    row = np.zeros(11, dtype=np.uint32)
    ids = np.zeros(10, dtype=np.uint32)
    need = C.c_uint64()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = _lib.lib().bmq_match_batch(eng.h, p(tdata), p(toff), 1, p(tt), p(pdata), p(poff), 10, p(row), p(ids), 10, C.byref(need))
    assert rc == -3 and need.value == 1000  # BMQ_E_NOSPACE, exact size reported
    ids = np.zeros(1000, dtype=np.uint32)
    rc = _lib.lib().bmq_match_batch(eng.h, p(tdata), p(toff), 1, p(tt), p(pdata), p(poff), 10, p(row), p(ids), 1000, C.byref(need))
    assert rc == 0 and row.tolist() == list(range(0, 1001, 100)) and ids.tolist() == list(range(100)) * 10


# ---- full BASELINE size: properties that do not need the oracle to finish the whole batch ---------------------------
def test_full_size_config2_properties(eng):
    """configs[1]: 1 tenant, 1M filters with +/# wildcards, 1M-publish batch.  Checked: CSR well-formed, rows strictly
    ascending, EVERY row bit-exact vs the oracle (production call pattern; the rows the reference loses routes in -- quirk (ii) -- vs the
    semantic oracle, all of them), duplicate topics get identical rows, idempotence across two runs."""
    w = B.Workload(0xB1F20002, 1, 1_000_000, 1)
    eng.rebuild(packed=w.keys_packed())
    data, off, tt = w.topics(0xB1F20002 + 1, 1_000_000)
    tn = w.tenants()
    row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
    assert row[0] == 0 and row[-1] == len(ids) and (np.diff(row.astype(np.int64)) >= 0).all()
    d = np.diff(ids.astype(np.int64))
    starts = row[1:-1][row[1:-1] < len(ids)]
    d[(starts - 1)[starts > 0]] = 1  # ignore row boundaries
    assert (d > 0).all()
    assert ids.max() < w.n_keys
    row2, ids2 = eng.match_batch(tn, tt, packed_topics=(data, off))
    assert (row == row2).all() and (ids == ids2).all()
    rnd = random.Random(3)
    kv = O.KV(packed=w.keys_packed())
    # EVERY row of the 1M-publish batch: the DISTINCT topics (Zipf: identical publishes are checked equal row for row just below) against
    # the oracle in the production call pattern -- one matchAll(singleton(topic)) each -- on all host cores, whole-CSR comparison; every
    # row that differs from that structural restatement (the reference loses routes to quirk (ii) there: hot filters next to
    # "<filter>/" filters) and a sub-sample of the others against the semantic oracle
    raw = data.tobytes()
    first_of, rep = {}, np.zeros(1_000_000, dtype=np.int64)
    for i in range(1_000_000):
        rep[i] = first_of.setdefault(raw[off[i]:off[i + 1]], i)
    cnt = np.diff(row.astype(np.int64))
    assert (cnt == cnt[rep]).all()
    dup = np.nonzero(rep != np.arange(1_000_000))[0]
    a_rp, a = U.csr_select(row, ids, dup)
    b_rp, b = U.csr_select(row, ids, rep[dup])
    assert np.array_equal(a, b)  # identical topic -> identical row
    sample = np.nonzero(rep == np.arange(1_000_000))[0]
    # in chunks of 50 000 distinct topics (a row of this workload has ~1000 ids: the whole CSR at once is half a billion ids per side)
    n_diff = n_lock = 0
    sec_sem = 0.0
    qc = {}
    for c0 in range(0, len(sample), 50_000):
        part = sample[c0:c0 + 50_000]
        sub_data, sub_off = U.sub_packed(raw, off, part)
        stt0 = np.zeros(len(part), dtype=np.uint32)
        res, _ = kv.match_singletons(tn, stt0, (sub_data, sub_off), threads=U.host_threads())
        got_rp, got = U.csr_select(row, ids, part)
        differ = U.assert_csr_equal_modulo_quirk_ii(w.keys, kv.key, tn, None, res.row_ptr.astype(np.int64), res.routes, got_rp, got, quirk_cache=qc)
        # every differing row + a few of the others against the semantic oracle (O(keys of the tenant) each, all host cores)
        rows = np.unique(np.concatenate([differ, np.arange(0, len(part), 5000)]))
        sd, so = U.sub_packed(sub_data, sub_off, rows)
        sem, sec = kv.match_semantic_batch(tn, np.zeros(len(rows), dtype=np.uint32), (sd, so), threads=U.host_threads())
        a_rp, a = U.csr_select(got_rp, got, rows)
        assert np.array_equal(a_rp, sem.row_ptr.astype(np.int64)) and np.array_equal(a, sem.routes)
        n_diff += len(differ)
        n_lock += res.livelocks
        sec_sem += sec
    U.parity_report("c2: 1 tenant x 1M routes, 1M publishes (%d distinct topics, all compared; identical publishes compared row for row)" % len(sample),
                    rows_compared=int(len(sample)), rows_differing_from_reference_restatement=int(n_diff),
                    differing_rows_equal_semantic_oracle=int(n_diff), reference_livelocks=int(n_lock), semantic_oracle_s=round(sec_sem, 2))


def test_full_size_config3_properties(eng):
    """configs[2] on one GPU = the bench default: 1000 tenants x 10k filters (10M route keys), 1M Zipf publishes.
    Checked: CSR well-formed, rows strictly ascending, idempotent, **tenant isolation** (every id of a row lies in the
    id range of the row's tenant), and EVERY row of the batch -- all 1000 tenants, 1 M publishes -- against the oracle in the production
    call pattern (ids are ranks in the KV), every differing row against the semantic oracle."""
    w = B.Workload(0xB1F20003, 1000, 10_000, 1)
    eng.rebuild(packed=w.keys_packed())
    assert eng.info().n_routes == w.n_keys == 10_000_000 and eng.info().n_tenants == 1000
    n = 1_000_000
    data, off, tt = w.topics(0xB1F20003 + 1000, n)
    tn = w.tenants()
    row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
    assert row[0] == 0 and row[-1] == len(ids) and (np.diff(row.astype(np.int64)) >= 0).all()
    d = np.diff(ids.astype(np.int64))
    starts = row[1:-1][row[1:-1] < len(ids)]
    d[(starts - 1)[starts > 0]] = 1  # ignore row boundaries
    assert (d > 0).all()
    row2, ids2 = eng.match_batch(tn, tt, packed_topics=(data, off))
    assert (row == row2).all() and (ids == ids2).all()
    first = np.asarray(w.tenant_first(), dtype=np.int64)  # id range of tenant t: [first[t], first[t + 1])
    counts = np.diff(row.astype(np.int64))
    owner = np.repeat(tt.astype(np.int64), counts)
    assert ((ids >= first[owner]) & (ids < first[owner + 1])).all()
    S = 1000  # EVERY publish of the batch (rounds 2-4: the first 128 tenants, 73 % of it) against the oracle, all host cores
    kb, ko = w.keys_packed()
    hi = int(first[S])
    sub_off = ko[:hi + 1].copy()
    sub_bytes = kb[:int(sub_off[-1]) + 1]
    kv = O.KV(packed=(sub_bytes, sub_off))
    cand = np.nonzero(tt < S)[0]
    assert len(cand) == n
    raw = data.tobytes()
    t_off = np.concatenate([[0], np.cumsum((off[cand + 1] - off[cand]).astype(np.int64))]).astype(np.uint32)
    t_data = np.zeros(int(t_off[-1]) + 32, dtype=np.uint8)
    t_data[:int(t_off[-1])] = np.frombuffer(b"".join(raw[off[i]:off[i + 1]] for i in cand), dtype=np.uint8)
    stt = tt[cand]
    res, _ = kv.match_singletons(tn[:S], stt, (t_data, t_off), threads=U.host_threads())
    got_rp, got = U.csr_select(row, ids, cand)  # ids of the first tenants are ranks in the sub-KV of exactly those tenants
    rawk = sub_bytes.tobytes()
    differ = U.assert_csr_equal_modulo_quirk_ii(lambda: [rawk[sub_off[i]:sub_off[i + 1]] for i in range(hi)], kv.key, tn[:S], stt,
                                                res.row_ptr.astype(np.int64), res.routes, got_rp, got)
    rnd = random.Random(4)
    # authoritative semantic check: EVERY row that differs from the structural restatement + a sub-sample of the others
    U.assert_differing_rows_semantic("c3: 1000 tenants x 10k routes, 1M publishes (every publish of all %d tenants)" % S, kv, tn[:S], stt,
                                     (t_data, t_off), differ, got_rp, got, livelocks=res.livelocks, extra_rows=rnd.sample(range(len(cand)), 200))
    # fan-out grouping of the whole batch (~18 M (topic, route) pairs regrouped by DelivererKey), size-independent properties:
    # a permutation of the pairs; inside a group (topic, route) ascending; one DelivererKey per group and one group per DelivererKey
    # (checked on a sample of pairs of every group through their route keys)
    ot, orr, goff, grep, special = eng.fanout_group(row, ids, group_cap=4096)
    assert goff[0] == 0 and goff[-1] == len(ids) and (np.diff(goff.astype(np.int64)) > 0).all()
    pair_in = (np.repeat(np.arange(n, dtype=np.int64), counts) << 32) | ids.astype(np.int64)
    pair_out = (ot.astype(np.int64) << 32) | orr.astype(np.int64)
    assert (np.sort(pair_out) == pair_in).all()  # the input pairs are already in (topic, route) order
    inside = np.ones(len(pair_out) - 1, dtype=bool)
    inside[goff[1:-1].astype(np.int64) - 1] = False  # group boundaries
    assert (np.diff(pair_out)[inside] > 0).all()
    seen_keys = {}
    for g in range(len(goff) - 1):
        lo, hi_ = int(goff[g]), int(goff[g + 1])
        if int(grep[g]) >= 0xFFFFFFFE:
            assert g >= len(goff) - 3  # the special groups come last
            if int(grep[g]) == 0xFFFFFFFE:  # shared subscriptions: every route of the group is a $share / $oshare route
                smp = orr[lo:hi_][:200]
                assert all(O.parse_route_key(k)[0] in (2, 3) for k in eng.route_keys(smp))
            continue
        pick = np.unique(np.concatenate([orr[lo:lo + 50], orr[hi_ - 50:hi_], orr[lo:hi_][:: max(1, (hi_ - lo) // 100)], [grep[g]]]))
        dks = {O.deliverer_key_of(k) for k in eng.route_keys(pick)}
        assert len(dks) == 1 and None not in dks, g
        dk = dks.pop()
        assert dk not in seen_keys
        seen_keys[dk] = g
    assert len(seen_keys) == 64 and not (special & 2)  # the generator's 64 deliverer keys d0..d63 under broker 0


def test_submit_wait_two_batches_in_flight(eng):
    """bmq_match_submit / bmq_match_wait: two host batches in flight (upload / kernels / download overlapped), page-locked buffers
    from bmq_host_alloc; every batch equals the blocking bmq_match_batch result, also with a bmq_routes_apply squeezed between a
    submit and its wait (stream order: the submitted batch still sees the old routes, the next one the new ones)."""
    from bifromq_amd.engine import pinned
    w = B.Workload(11, 6, 3000, 1)
    keys = w.keys()
    eng.rebuild(keys)
    tn = w.tenants()
    tdata, toff = w.tenants_packed()
    p_t, p_to = pinned(len(tdata), np.uint8), pinned(len(toff), np.uint32)
    p_t[:], p_to[:] = tdata, toff
    batches, expect = [], []
    for b in range(5):
        data, off, tt = w.topics(100 + b, 20000 + 1000 * b)
        pd, po, pt = pinned(len(data), np.uint8), pinned(len(off), np.uint32), pinned(len(tt), np.uint32)
        pd[:], po[:], pt[:] = data, off, tt
        batches.append((pd, po, pt, len(tt)))
        expect.append(eng.match_batch(tn, tt, packed_topics=(data, off)))
    rows = [pinned(40000, np.uint32) for _ in range(2)]
    ids = [pinned(4_000_000, np.uint32) for _ in range(2)]
    tickets = [None, None]
    for i in range(len(batches) + 1):
        if i < len(batches):
            pd, po, pt, n = batches[i]
            tickets[i % 2] = eng.match_submit(p_t, p_to, len(tn), pt, pd, po, n)
            if i == 0:
                with pytest.raises(B.BmqError):  # ids buffer too small for the wait: NOSPACE, the needed size is reported
                    eng.match_wait(tickets[0], rows[0], ids[0][:1])
                tickets[0] = eng.match_submit(p_t, p_to, len(tn), pt, pd, po, n)
        if i >= 1:
            j = (i - 1) % 2
            n = batches[i - 1][3]
            got = eng.match_wait(tickets[j], rows[j], ids[j])
            erow, eids = expect[i - 1]
            assert got == len(eids) and (rows[j][:n + 1] == erow).all() and (ids[j][:got] == eids).all()
    with pytest.raises(B.BmqError):
        eng.match_wait(0, rows[0], ids[0])  # nothing in flight under that ticket
    # a submit while every ticket is in flight is refused; an apply between submit and wait lands behind the submitted batch
    pd, po, pt, n = batches[0]
    t0 = eng.match_submit(p_t, p_to, len(tn), pt, pd, po, n)
    t1 = eng.match_submit(p_t, p_to, len(tn), pt, pd, po, n)
    t2 = eng.match_submit(p_t, p_to, len(tn), pt, pd, po, n)
    assert sorted((t0, t1, t2)) == [0, 1, 2]  # BMQ_MAX_TICKETS
    with pytest.raises(B.BmqError):
        eng.match_submit(p_t, p_to, len(tn), pt, pd, po, n)
    extra = _normal(tn[0], "#", 0, "late", "d")
    eng.apply([(0, extra)])
    for t in (t0, t1, t2):
        got = eng.match_wait(t, rows[0], ids[0])
        assert got == len(expect[0][1]) and (ids[0][:got] == expect[0][1]).all()  # submitted before the apply
    row2, ids2 = eng.match_batch(tn, pt, packed_topics=(pd, po))
    assert len(ids2) > len(expect[0][1])  # the '#' route of tenant 0 now matches its non-'$' topics


def test_apply_async_lands_behind_a_submitted_batch(eng):
    """bmq_routes_apply_async: the batch is uploaded beside, and applied behind, a match batch handed over with bmq_match_submit; its
    outcome is bmq_routes_apply_wait's or the next index call's.  Same rows, same ids as the blocking bmq_routes_apply on a second engine --
    also when the batch needs the host (a tenant the index does not know: the gate closes, the stage-by-stage loops finish the batch) --; a
    batch with a malformed key changes nothing and its error reaches whoever asks next."""
    from bifromq_amd.engine import pinned
    w = B.Workload(23, 5, 2500, 1)
    keys = w.keys()
    tn = w.tenants() + ["a-tenant-the-index-does-not-know-yet"]
    ref = B.Engine(device=0)
    try:
        eng.rebuild(keys)
        ref.rebuild(keys)
        tdata, toff = O.pack(tn)
        p_t, p_to = pinned(len(tdata), np.uint8), pinned(len(toff), np.uint32)
        p_t[:], p_to[:] = tdata, toff
        data, off, tt = w.topics(77, 30000)
        tt = tt.copy()
        tt[::50] = len(tn) - 1  # publishes of the tenant that only appears with the second batch
        pd, po, pt = pinned(len(data), np.uint8), pinned(len(off), np.uint32), pinned(len(tt), np.uint32)
        pd[:], po[:], pt[:] = data, off, tt
        rows, ids = pinned(40000, np.uint32), pinned(8_000_000, np.uint32)
        # (every ticket slot once, so that its buffers have their size and the engine knows the batch is not grouped by tenant: a batch that does
        # not fit, or that wants the other k_walk instantiation, is run AGAIN by bmq_match_wait -- against the index as it is then, mutations
        # queued behind the first run included)
        warm = [eng.match_submit(p_t, p_to, len(tn), pt, pd, po, len(tt)) for _ in range(3)]
        for t in warm:
            eng.match_wait(t, rows, ids)
        rnd = random.Random(5)
        batches = []
        dels = rnd.sample(range(len(keys)), 600)
        batches.append([(1, keys[i]) for i in dels[:300]] + [(0, _normal(tn[q % 5], "l0_%d/#" % (q % 8), 0, "as%d" % q, "d%d" % (q % 7))) for q in range(300)])
        batches.append([(1, keys[i]) for i in dels[300:]] + [(0, _normal(tn[-1], "#" if q % 2 else "+/+/#", 0, "nt%d" % q, "d1")) for q in range(40)])
        for ops in batches:
            rnd.shuffle(ops)
            before = ref.match_batch(tn, tt, packed_topics=(data, off))
            t = eng.match_submit(p_t, p_to, len(tn), pt, pd, po, len(tt))
            eng.apply_async(ops)  # queued behind the submitted batch; the call does not wait
            got = eng.match_wait(t, rows, ids)
            assert got == len(before[1]) and (rows[:len(tt) + 1] == before[0]).all() and (ids[:got] == before[1]).all()  # the batch saw the index as it was
            ref.apply(ops)
            after = ref.match_batch(tn, tt, packed_topics=(data, off))
            row2, ids2 = eng.match_batch(tn, tt, packed_topics=(data, off))  # (fetches the apply's outcome first)
            assert (row2 == after[0]).all() and (ids2 == after[1]).all() and not (len(after[1]) == len(before[1]) and (after[1] == before[1]).all())
            eng.apply_wait()
            assert eng.info().n_routes == ref.info().n_routes and eng.info().next_route_id == ref.info().next_route_id
        # a malformed key: nothing is changed; the error is apply_wait's ...
        bad = [(0, _normal(tn[0], "x/#", 0, "ok", "d")), (0, b"\x00\x07not-a-route-key")]
        n0 = eng.info().n_routes
        eng.apply_async(bad)
        with pytest.raises(B.BmqError) as ei:
            eng.apply_wait()
        assert ei.value.code == -1 and eng.info().n_routes == n0
        # ... or, when nobody waits, the next call's that needs the index; the call after it works again
        eng.apply_async(bad)
        with pytest.raises(B.BmqError):
            eng.match_batch(tn, tt, packed_topics=(data, off))
        row3, ids3 = eng.match_batch(tn, tt, packed_topics=(data, off))
        assert (ids3 == ids2).all() and eng.info().n_routes == n0
    finally:
        ref.close()


def test_dev_protocol_belongs_to_one_thread(eng):
    """bmq_match_batch_dev .. bmq_match_finish is one caller's window: another thread's *_dev launch is refused meanwhile
    (BMQ_E_STATE) instead of overwriting the batch in flight.  Device buffers come straight from the HIP runtime the library
    itself is linked against (ctypes; torch would bring a second copy of the runtime into the process)."""
    import ctypes as C
    import threading
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    bufs = []

    def to_dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes + 64) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        bufs.append(p)
        return p.value

    w = B.Workload(3, 2, 500, 1)
    eng.rebuild(w.keys())
    tn = w.tenants()
    data, off, tt = w.topics(1, 1000)
    tdata, toff = w.tenants_packed()
    d = [to_dev(np.ascontiguousarray(x)) for x in (tdata.copy(), toff.astype(np.uint32), tt.astype(np.uint32), data, off.astype(np.uint32))]
    row, ids, tot = to_dev(np.zeros(1001, dtype=np.uint32)), to_dev(np.zeros(200000, dtype=np.uint32)), to_dev(np.zeros(1, dtype=np.uint64))
    args = (d[0], d[1], len(tn), d[2], d[3], d[4], 1000, row, ids, 200000, tot)
    eng.match_batch_device(*args)
    seen = []

    def other():
        try:
            eng.match_batch_device(*args)
            seen.append("launched")
        except B.BmqError as ex:
            seen.append(ex.code)
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen == [-7]
    n = eng.finish()
    exp_row, exp_ids = eng.match_batch(tn, tt, packed_topics=(data, off))
    got = np.zeros(n, dtype=np.uint32)
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), C.c_void_p(ids), 4 * n, 2) == 0  # hipMemcpyDeviceToHost
    assert n == len(exp_ids) and (got == exp_ids).all()
    for p in bufs:
        hip.hipFree(p)


def test_in_library_rccl_exchange_world_size_one(eng):
    """bmq_exchange_fanout / bmq_exchange_csr (RCCL inside libbmq, bmq_exchange.inc) at world size 1 -- the largest this pool
    offers: the communicator comes up, the collectives run on the exchange stream behind the batch, and what comes back is this
    rank's own fan-out vector / CSR, equal to the oracle's."""
    import ctypes as C
    from bifromq_amd import _lib
    L = _lib.lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    bufs = []

    def to_dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes + 64) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
        bufs.append(p)
        return p.value

    def from_dev(p, n, dtype):
        out = np.zeros(n, dtype=dtype)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes, 2) == 0
        return out

    e2 = B.Engine(device=0)  # a fresh engine: the communicator belongs to it
    w = B.Workload(21, 3, 800, 1)
    keys = w.keys()
    e2.rebuild(keys)
    tn = w.tenants()
    n = 3000
    data, off, tt = w.topics(2, n)
    uid = C.create_string_buffer(128)
    assert L.bmq_comm_unique_id(uid) == 0
    assert L.bmq_comm_init(e2.h, 1, 0, uid) == 0
    tdata, toff = w.tenants_packed()
    d = [to_dev(np.ascontiguousarray(x)) for x in (tdata.copy(), toff.astype(np.uint32), tt.astype(np.uint32), data, off.astype(np.uint32))]
    cap = 400000
    row, ids, tot = to_dev(np.zeros(n + 1, dtype=np.uint32)), to_dev(np.zeros(cap, dtype=np.uint32)), to_dev(np.zeros(1, dtype=np.uint64))
    counts_all, rows_all, ids_all = to_dev(np.zeros(n, dtype=np.uint32)), to_dev(np.zeros(n + 1, dtype=np.uint32)), to_dev(np.zeros(cap, dtype=np.uint32))
    e2.match_batch_device(d[0], d[1], len(tn), d[2], d[3], d[4], n, row, ids, cap, tot)
    total = e2.finish()
    kv = O.KV(keys)
    topics = [t.decode() for t in unpack(data, off)]
    exp = U.semantic_rows(kv, tn, tt, topics)
    assert L.bmq_exchange_fanout(e2.h, row, n, counts_all) == 0
    totals = np.zeros(1, dtype=np.uint64)
    assert L.bmq_exchange_csr(e2.h, row, ids, n, total, rows_all, ids_all, cap, totals.ctypes.data_as(C.c_void_p)) == 0
    assert L.bmq_exchange_wait(e2.h) == 0
    assert from_dev(counts_all, n, np.uint32).tolist() == [len(r) for r in exp]
    assert int(totals[0]) == total == sum(len(r) for r in exp)
    rp = from_dev(rows_all, n + 1, np.uint32)
    got = from_dev(ids_all, total, np.uint32)
    assert U.csr_rows(rp, got) == exp
    small = np.zeros(1, dtype=np.uint64)
    assert L.bmq_exchange_csr(e2.h, row, ids, n, total, rows_all, ids_all, 10, small.ctypes.data_as(C.c_void_p)) == -3  # NOSPACE, totals reported
    assert int(small[0]) == total
    for p in bufs:
        hip.hipFree(p)
    e2.close()


def test_node_wide_path_two_shards_on_one_gpu():
    """The node-wide shape of SURVEY 8e without a second GPU: two engines hold the shards ranks 0 and 1 of a 2-GPU node would hold
    (tenants by hash(tenantId) mod 2, the hottest tenant split by filter: its route keys by hash(route key) mod 2), ONE publish batch
    for all tenants lies in HBM, each "rank" picks its part with bmq_partition_batch_dev (kernels of the library), matches it, and the
    per-topic fan-outs add up.  The sum must equal the fan-out of an unsharded engine -- and the oracle's."""
    import torch

    from bifromq_amd import shard

    W = 2
    w = B.Workload(0xB1F20077, 24, 800, 1)
    keys, tn = w.keys(), w.tenants()
    n = 30000
    data, off, tt = w.topics(5, n)
    share = np.bincount(tt, minlength=len(tn)) / float(n)
    hot = [int(np.argmax(share))]
    owner = shard.topic_targets(tn, hot, W)
    assert (owner < 0).sum() == 1 and set(owner[owner >= 0].tolist()) == {0, 1}
    tidx = {t: i for i, t in enumerate(tn)}
    key_tenant = [tidx[B.decode_route_key(k)[1]] for k in keys]
    dev = torch.device("cuda", 0)
    d_data = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
    d_data[:len(data)] = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(off.astype(np.int32)).to(dev)
    d_tt = torch.from_numpy(tt.astype(np.int32)).to(dev)
    d_owner = torch.from_numpy(owner).to(dev)
    tdata, toff = w.tenants_packed()
    d_tenants = torch.from_numpy(tdata.copy()).to(dev)
    d_tenant_off = torch.from_numpy(toff.astype(np.int32)).to(dev)
    total = torch.zeros(n, dtype=torch.int64, device=dev)
    parts = []
    for rank in range(W):
        eng = B.Engine(device=0).rebuild(sorted(shard.shard_keys(keys, key_tenant, tn, hot, W, rank)))
        part = shard.DevicePartition(eng, d_owner, n, len(data), dev)
        sel, pd, po, ptt, m = part(d_tt, d_data, d_off, rank)
        parts.append(sel.cpu().numpy().copy())
        d_row = torch.zeros(m + 1, dtype=torch.int32, device=dev)
        d_ids = torch.zeros(64 * n, dtype=torch.int32, device=dev)
        d_total = torch.zeros(1, dtype=torch.int64, device=dev)
        eng.match_batch_device(d_tenants.data_ptr(), d_tenant_off.data_ptr(), len(tn), ptt.data_ptr(), pd.data_ptr(), po.data_ptr(), m, d_row.data_ptr(),
                               d_ids.data_ptr(), d_ids.numel(), d_total.data_ptr())
        eng.finish()
        counts = (d_row[1:] - d_row[:-1]).to(torch.int64)
        total.index_add_(0, sel.long(), counts)
        # the part is exactly the topics of the rank's tenants and of the split tenant, in batch order, bytes intact
        want_sel = np.nonzero((owner[tt] == rank) | (owner[tt] < 0))[0]
        assert np.array_equal(parts[-1], want_sel)
        po_h, pd_h = po.cpu().numpy(), pd.cpu().numpy()
        raw = data.tobytes()
        for j in (0, 1, m // 2, m - 1):
            g = int(want_sel[j])
            assert pd_h[po_h[j]:po_h[j + 1]].tobytes() == raw[off[g]:off[g + 1]]
        assert np.array_equal(ptt.cpu().numpy(), tt[want_sel].astype(np.int32))
        eng.close()
    assert len(np.intersect1d(parts[0], parts[1])) == int((owner[tt] < 0).sum())  # only the split tenant's publishes go to both
    whole = B.Engine(device=0).rebuild(keys)
    row, ids = whole.match_batch(tn, tt, packed_topics=(data, off))
    assert np.array_equal(total.cpu().numpy(), np.diff(row.astype(np.int64)))
    kv = O.KV(keys)
    topics = unpack(data, off)
    for i in range(0, n, 997):
        assert int(total[i]) == len(kv.match_bruteforce(tn[int(tt[i])], [topics[i]]).per_topic()[0])
    whole.close()
