"""GPU parity tests of the retain direction (wildcard filters against the retained-topic index) vs the oracle's
restatement of TopicLevelTrie.lookup + RetainMatcher.  Golden tables: RST/index/RetainTopicIndexTest.java:42-76."""
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

from tests.test_oracle_golden import INDEX_ROWS, RETAIN_TOPICS, TOPIC_INDEX_EXTRA_ROWS

TOPICS = RETAIN_TOPICS
# RST/index/RetainTopicIndexTest.java:42-76,113-117 (filter -> matched topics), as ported for the oracle
TABLE = dict(INDEX_ROWS + TOPIC_INDEX_EXTRA_ROWS + [("+/b", ["/b", "a/b"]), ("nope/#", []), ("a/b/c/d", [])])


@pytest.fixture(scope="module")
def eng():
    e = B.Engine(device=0)
    yield e
    e.close()


def test_golden_table(eng):
    eng.retain_rebuild(["tenantA", "tenantB"], [0] * len(TOPICS) + [1], TOPICS + ["a"])
    order = U.retain_order(["tenantA", "tenantB"], [0] * len(TOPICS) + [1], TOPICS + ["a"])  # independent of the engine
    assert [eng.retain_topic(i) for i in range(len(order))] == order and eng.retain_find_all()[0] == len(order)
    ids = {pair: i for i, pair in enumerate(order)}
    assert len(ids) == len(TOPICS) + 1
    lt = O.LevelTrie(1)
    for (tenant, topic), i in ids.items():
        lt.add(tenant, topic, i)
    for f, exp in TABLE.items():
        got = eng.retain_match("tenantA", f)
        assert sorted(eng.retain_topic(i)[1] for i in got) == sorted(exp), f
        assert got == sorted(lt.match("tenantA", f)), f
        assert got == sorted(got)
    assert [eng.retain_topic(i) for i in eng.retain_match("tenantB", "#")] == [("tenantB", "a")]
    assert eng.retain_match("ghost", "#") == []


@pytest.mark.parametrize("seed", [1, 2])
def test_random_parity(eng, seed):
    rnd = random.Random(seed)
    tenants = ["tA", "tB"]
    topics = sorted({(rnd.randrange(2), U.rand_topic(rnd, 5)) for _ in range(5000)})
    eng.retain_rebuild(tenants, [t for t, _ in topics], [p for _, p in topics])
    order = U.retain_order(tenants, [t for t, _ in topics], [p for _, p in topics])  # ids = ranks of an independent sort
    lt = O.LevelTrie(1)
    n = len(order)
    assert eng.retain_find_all()[0] == n
    for i, (tn, tp) in enumerate(order):
        assert eng.retain_topic(i) == (tn, tp)
        lt.add(tn, tp, i)
    filters = [U.rand_filter(rnd, 6) for _ in range(3000)] + ["#", "+", "+/#", "+/+", "/", "", "$sys/#", "$sys/+", "a/+/#"]
    ft = [rnd.randrange(3) for _ in filters]
    tnames = tenants + ["ghost"]
    row, ids = eng.retain_match_batch(tnames, ft, filters)
    got = U.csr_rows(row, ids)
    for i, f in enumerate(filters):
        exp = sorted(lt.match(tnames[ft[i]], f))
        assert got[i] == exp, (f, tnames[ft[i]])
    # independent semantic check on a sample: retained topic matches iff the MQTT rule says so
    for i in range(0, len(filters), 11):
        if filters[i].count("#") > 1 or ("#" in filters[i][:-1]):
            continue
        exp = [k for k in range(n) if eng.retain_topic(k)[0] == tnames[ft[i]] and S.matches(eng.retain_topic(k)[1], filters[i])]
        assert got[i] == exp, filters[i]


def test_generated_workload_parity(eng):
    """config 4 shape at an oracle-friendly size: literal retained topics, wildcard query filters."""
    w = B.Workload(0xB1F20004, 4, 1, 0)
    data, off, tt = w.retain(0xB1F20004, 60000, filters=False)
    tn = w.tenants()
    eng.retain_rebuild(tn, tt, packed_topics=(data, off))
    lt = O.LevelTrie(1)
    seen = set()
    i = 0
    while True:
        try:
            tenant, topic = eng.retain_topic(i)
        except B.BmqError:
            break
        assert (tenant, topic) not in seen
        seen.add((tenant, topic))
        lt.add(tenant, topic, i)
        i += 1
    assert i == len({(int(t), p) for t, p in zip(tt, unpack(data, off))})
    fdata, foff, ft = w.retain(0xB1F20004 + 1, 20000, filters=True)
    row, ids = eng.retain_match_batch(tn, ft, packed_filters=(fdata, foff))
    res, _ = lt.match_batch(tn, ft, (fdata, foff), threads=U.host_threads())
    assert U.csr_rows(row, ids) == [sorted(r) for r in res.per_topic()]
    assert (np.diff(row.astype(np.int64)) >= 0).all()


def test_match_limited_vs_retain_store_coproc_match(eng):
    """RetainStoreCoProc.match(limit, now) (RS/RetainStoreCoProc.java:167-190): the first `limit` matches that have NOT expired.
    Engine rows == the oracle restatement (O.retain_store_match over the oracle's TopicLevelTrie, ids from an independent sort,
    expiry instants from O.retain_expire_at) for a mix of limits, at several `now`; both the range-select path (limits <= 64)
    and the full-CSR path (a limit > 64 in the batch) are exercised."""
    w = B.Workload(0xB1F20004, 3, 1, 0)
    data, off, tt = w.retain(77, 30000, filters=False)
    tn = w.tenants()
    topics = [t.decode() for t in unpack(data, off)]
    rnd = random.Random(3)
    base_ms = 1_700_000_000_000
    ts = [((base_ms + rnd.randrange(0, 100_000)) << 16) | rnd.randrange(1 << 16) for _ in topics]
    ex = [rnd.choice([0, 1, 30, 60, 3600, 0x7FFFFFFF]) for _ in topics]
    eng.retain_rebuild(tn, tt, topics, timestamps=ts, expiry=ex)
    order = U.retain_order(tn, tt, topics)
    stamp = {}
    for t, tp, a, b in zip(tt, topics, ts, ex):  # the LAST add of a duplicated topic wins
        stamp[(tn[int(t)], tp)] = (a, b)
    expire = [O.retain_expire_at(*stamp[p]) for p in order]
    lt = O.LevelTrie(1)
    for i, (tenant, topic) in enumerate(order):
        lt.add(tenant, topic, i)
    for i in rnd.sample(range(len(order)), 200):
        assert eng.retain_topic(i) == order[i] and eng.retain_topic_info(i) == (stamp[order[i]][0], stamp[order[i]][1], expire[i])
    fdata, foff, ft = w.retain(78, 3000, filters=True)
    filters = [f.decode() for f in unpack(fdata, foff)]
    row, ids = eng.retain_match_batch(tn, ft, filters)
    full = U.csr_rows(row, ids)
    assert any(len(r) > 10 for r in full) and any(0 < len(r) <= 10 for r in full)  # both regimes are exercised
    for now, big in ((0, False), (base_ms + 20_000, False), (base_ms + 50_000 + 45_000, True), (base_ms + 10**7, False)):
        limits = [rnd.choice([0, 1, 2, 10, 10, 10, 64] + ([100, 0xFFFFFFFF] if big else [])) for _ in filters]  # 10 = the default
        lrow, lids, counts = eng.retain_match_limited(tn, ft, filters, limits, now_ms=now)
        assert counts.tolist() == [len(r) for r in full]
        exp = [O.retain_store_match(lt, tn[int(t)], f, l, now, expire.__getitem__) for t, f, l in zip(ft, filters, limits)]
        assert U.csr_rows(lrow, lids) == exp
        if now == 0:  # nothing has expired: the prefix of the unlimited row
            assert exp == [r[:min(l, len(r))] for r, l in zip(full, limits)]
    # the GC scan (RS/RetainStoreCoProc.java:257-277): expired ids of one tenant / of all, with and without an expiry override
    now = base_ms + 60_000
    first = [i for i, (t, _) in enumerate(order) if t == tn[0]]
    # (with a tenant the reference scans index.match(tenantId, "#"), which never reaches '$' topics; without one it scans findAll())
    reach = [i for i in first if not order[i][1].startswith("$")]
    assert len(reach) < len(first)
    assert eng.retain_expired(tn[0], now) == [i for i in reach if expire[i] <= now] == [i for i in lt.match(tn[0], "#") if expire[i] <= now]
    assert eng.retain_expired(None, now) == [i for i in range(len(order)) if expire[i] <= now]
    assert eng.retain_expired(tn[0], now, 5) == [i for i in reach if O.retain_expire_at(stamp[order[i]][0], 5) <= now]
    # a single filter, unknown tenant
    lrow, lids, counts = eng.retain_match_limited(tn + ["nobody"], [3], ["#"], [5])
    assert lrow.tolist() == [0, 0] and counts.tolist() == [0]
    lrow, lids, counts = eng.retain_match_limited(tn, [0], ["#"], [7], now_ms=0)
    everything = eng.retain_match(tn[0], "#")
    assert counts[0] == len(everything) > 7 and lids.tolist() == everything[:7]
    # add() of a topic that is there replaces its stamp (RS/RetainStoreCoProc.java:246-249); remove() forgets it -- ids are stable
    # handles: nobody else's id moves, the removed topic's id goes dead (and comes back with the topic)
    victim = order[first[0]][1]
    eng.retain_apply(tn[0], [(0, victim, (base_ms << 16), 1)])
    assert eng.retain_topic_info(first[0])[2] == base_ms + 1000
    eng.retain_apply(tn[0], [(1, victim)])
    assert eng.retain_find_all()[0] == len(order) - 1 and eng.retain_topic(first[1]) == order[first[1]]
    assert first[0] not in eng.retain_match(tn[0], "#") and first[0] not in eng.retain_live_ids(tn[0])
    with pytest.raises(B.BmqError):
        eng.retain_topic_info(first[0])
    lrow, lids, counts = eng.retain_match_limited(tn, [0], ["#"], [7], now_ms=0)
    assert counts[0] == len(everything) - (0 if victim.startswith("$") else 1) and first[0] not in lids.tolist()
    assert eng.retain_apply_batch(tn, [0], [(0, victim)]).tolist() == [first[0]]  # retained again: the same id
    assert eng.retain_topic_info(first[0])[2] == 0xFFFFFFFFFFFFFFFF and eng.retain_find_all()[0] == len(order)


def test_edge_shapes(eng):
    deep = "/".join(["a"] * 40)
    eng.retain_rebuild(["t"], [0, 0, 0, 0], [deep, "a", "/", "x" * 5000 + "/y"])
    assert [eng.retain_topic(i)[1] for i in eng.retain_match("t", "/".join(["a"] * 39) + "/+")] == [deep]
    assert [eng.retain_topic(i)[1] for i in eng.retain_match("t", "/".join(["+"] * 40))] == [deep]
    assert [eng.retain_topic(i)[1] for i in eng.retain_match("t", "x" * 5000 + "/#")] == ["x" * 5000 + "/y"]
    assert eng.retain_match("t", "x" * 4999 + "/#") == []
    assert len(eng.retain_match("t", "#")) == 4
    assert eng.retain_match("t", "/".join(["+"] * 65)) == []  # (round 3: BMQ_E_RANGE for the whole batch -- the kernel's 64-level limit)
    assert len(eng.retain_match("t", "#")) == 4


def test_filters_deeper_than_64_levels_inside_a_normal_batch(eng):
    """TopicLevelTrie.lookup has no depth limit (UTIL/index/TopicLevelTrie.java:190-249).  k_retain_walk keeps a filter's per-level arrays in
    LDS (64 levels); deeper filters are listed and answered by a second launch of the same walk with those arrays in global memory
    (k_retain_walk_deep, in the pipeline only while batches hold such filters).  65-, 100- and 300-level filters -- literal, '+', trailing
    '#', matching and not -- inside a batch of ordinary ones, against the oracle; before and after an apply (overlay + dead ids); the deep
    pass switches itself off again after 32 batches without such filters."""
    rnd = random.Random(21)
    deep_topics = ["/".join(["a"] * n) for n in (64, 65, 66, 100, 299, 300)] + ["/".join(["a"] * 99 + ["b"]), "/".join(["a"] * 70 + ["", "c"])]
    shallow = sorted({U.rand_topic(rnd, 5, ["a", "b", "c", "", "$sys"]) for _ in range(400)})
    topics = deep_topics + shallow
    eng.retain_rebuild(["t"], [0] * len(topics), topics)
    lt = O.LevelTrie(1)
    ids = {}
    for i in eng.retain_live_ids("t"):
        ids[eng.retain_topic(i)[1]] = i
    for tp, i in ids.items():
        lt.add("t", tp, i)
    filters = [U.rand_filter(rnd, 5, ["a", "b", "c", ""]) for _ in range(300)] + [
        "/".join(["a"] * 65), "/".join(["+"] * 65), "/".join(["a"] * 64 + ["+"]), "/".join(["a"] * 64 + ["#"]), "/".join(["+"] * 64 + ["#"]),
        "/".join(["a"] * 100), "/".join(["a"] * 99 + ["+"]), "/".join(["+"] * 99 + ["b"]), "/".join(["a"] * 300), "/".join(["+"] * 300),
        "/".join(["a"] * 298 + ["#"]), "/".join(["a"] * 70 + ["", "+"]), "/".join(["b"] * 80), "/".join(["a"] * 301), "/".join(["a"] * 64)]
    rnd.shuffle(filters)

    def check():
        row, got = eng.retain_match_batch(["t"], [0] * len(filters), filters)
        exp = [sorted(lt.match("t", f)) for f in filters]
        assert U.csr_rows(row, got) == exp
        assert sum(len(e) for e, f in zip(exp, filters) if f.count("/") >= 64) >= 12  # the deep filters really match something

    check()
    # churn: remove two deep topics, add one deep and one shallow (overlay), then the same batch again
    eng.retain_apply("t", [(1, deep_topics[1]), (1, deep_topics[3])])
    lt.remove("t", deep_topics[1], ids[deep_topics[1]])
    lt.remove("t", deep_topics[3], ids[deep_topics[3]])
    new = ["/".join(["a"] * 65 + ["z"]), "a/zz"]
    out = eng.retain_apply_batch(["t"], [0, 0], [(0, new[0]), (0, new[1])])
    for tp, i in zip(new, out.tolist()):
        lt.add("t", tp, i)
    filters.append("/".join(["a"] * 65 + ["+"]))
    check()
    for _ in range(40):  # ordinary batches: the deep pass leaves the pipeline again, the next deep filter brings it back
        assert eng.retain_match("t", "a/#") == sorted(lt.match("t", "a/#"))
    check()


def test_apply_is_per_tenant(eng):
    """add/remove of one tenant at a time -- tenants appear, disappear and come back, one grows by 500 topics --: after every step the
    retained set and every match equal the oracle's TopicLevelTrie fed with the engine's (stable) ids."""
    rnd = random.Random(8)
    state = {"tA": {"a/b", "a/c", "x"}, "tB": {"a/b"}, "tD": {"$sys/1", "q/r/s"}}
    items = [(t, p) for t, ps in state.items() for p in ps]
    tn = sorted(state)
    eng.retain_rebuild(tn, [tn.index(t) for t, _ in items], [p for _, p in items])

    def check():
        lt = O.LevelTrie(1)
        n = sum(len(v) for v in state.values())
        live = eng.retain_live_ids()
        assert len(live) == n == eng.retain_find_all()[0]
        for i, (tenant, topic) in zip(live, eng.retain_topics(live)):
            assert topic in state[tenant]
            lt.add(tenant, topic, i)
        names = sorted(state) + ["ghost"]
        filters = ["#", "+/#", "a/+", "a/#", "+/+/+", "$sys/#", "new/+/3", "q/r/+"]
        ft = [i % len(names) for i in range(len(filters) * len(names))]
        fl = [filters[i // len(names)] for i in range(len(filters) * len(names))]
        row, ids = eng.retain_match_batch(names, ft, fl)
        got = U.csr_rows(row, ids)
        for i in range(len(fl)):
            assert got[i] == sorted(lt.match(names[ft[i]], fl[i])), (names[ft[i]], fl[i])

    check()
    steps = [("tA", [(0, "a/d"), (1, "x")]), ("tC", [(0, "new/1/3"), (0, "new/2/3")]),         # a tenant appears
             ("tB", [(1, "a/b")]),                                                             # a tenant disappears
             ("tA", [(0, "g/%d/%d" % (i, i % 7)) for i in range(500)]),                        # outgrows its segment
             ("tD", [(1, "$sys/1"), (0, "$sys/2"), (0, "a")]), ("tB", [(0, "a/z")])]           # ... and returns
    for tenant, ops in steps:
        eng.retain_apply(tenant, ops)
        for o, tp in ops:
            s_ = state.setdefault(tenant, set())
            (s_.discard if o else s_.add)(tp)
        for k in [k for k, v in state.items() if not v]:
            del state[k]
        check()


def test_apply_add_remove(eng):
    eng.retain_rebuild(["t"], [0, 0, 0], ["a/b", "a/c", "x"])
    assert [eng.retain_topic(i)[1] for i in eng.retain_match("t", "a/+")] == ["a/b", "a/c"]
    eng.retain_apply("t", [(1, "a/b"), (0, "a/d"), (0, "a/c"), (1, "zzz")])
    assert [eng.retain_topic(i)[1] for i in eng.retain_match("t", "a/+")] == ["a/c", "a/d"]
    assert sorted(eng.retain_topic(i)[1] for i in eng.retain_match("t", "#")) == ["a/c", "a/d", "x"]
    assert eng.retain_match("t", "#") == sorted(eng.retain_match("t", "#"))


def test_full_size_config4_properties(eng):
    """configs[3]: 1M retained topics, 100k wildcard filters.  CSR well-formed, rows strictly ascending, every id in range,
    idempotent; EVERY row bit-exact vs the oracle's TopicLevelTrie restatement."""
    w = B.Workload(0xB1F20004, 1, 1, 0)
    data, off, tt = w.retain(0xB1F20004, 1_000_000, filters=False)
    tn = w.tenants()
    eng.retain_rebuild(tn, tt, packed_topics=(data, off))
    fdata, foff, ft = w.retain(0xB1F20004 + 1, 100_000, filters=True)
    row, ids = eng.retain_match_batch(tn, ft, packed_filters=(fdata, foff))
    assert row[0] == 0 and row[-1] == len(ids) and (np.diff(row.astype(np.int64)) >= 0).all()
    d = np.diff(ids.astype(np.int64))
    starts = row[1:-1][row[1:-1] < len(ids)]
    d[(starts - 1)[starts > 0]] = 1  # ignore row boundaries
    assert (d > 0).all()
    row2, ids2 = eng.retain_match_batch(tn, ft, packed_filters=(fdata, foff))
    assert (row == row2).all() and (ids == ids2).all()
    topics_raw = data.tobytes()
    order = sorted({tuple(topics_raw[off[i]:off[i + 1]].split(b"/")) for i in range(1_000_000)})  # independent of the engine
    n_index = len(order)
    assert eng.retain_find_all()[0] == n_index
    lt = O.LevelTrie(1)
    for i, lv in enumerate(order):
        lt.add(tn[0], b"/".join(lv), i)
    for i in random.Random(5).sample(range(n_index), 5000):  # engine ids == ranks of the independent sort
        assert eng.retain_topic(i) == (tn[0], b"/".join(order[i]).decode())
    assert ids.max() < n_index
    # EVERY one of the 100 000 rows against the oracle's TopicLevelTrie restatement (UTIL/index/TopicLevelTrie.java:190-249 +
    # RS/index/RetainTopicIndex.java:36-124) on all host cores: whole-CSR comparison (the oracle's rows come out ascending: std::set)
    res, _ = lt.match_batch(tn, np.zeros(100_000, dtype=np.uint32), (fdata, foff), threads=U.host_threads())
    assert np.array_equal(res.row_ptr.astype(np.int64), row.astype(np.int64))
    assert np.array_equal(res.routes, ids)
    U.parity_report("c4: 1M retained topics, 100k wildcard filters (all rows compared)", rows_compared=100_000, rows_differing_from_reference_restatement=0,
                    ids=int(len(ids)))
