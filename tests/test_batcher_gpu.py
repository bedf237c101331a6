"""GPU tests of the batching front (include/bmq.h, bmq_batcher_*): many threads issue the production
call -- ITenantRouteMatcher.matchAll(singleton(topic)) per cache miss, DW/cache/TenantRouteCache.java:180-193 -- and
every caller must get exactly the rows a direct batch (and the oracle) gives for its topics."""
import threading

import numpy as np
import pytest

import bifromq_amd as B
from bifromq_amd import _lib
from bifromq_amd.workload import unpack
from oracle import oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


def _row_hash(ids):
    a = np.ascontiguousarray(ids, dtype=np.uint32)
    return int(_lib.gen().bmqgen_row_hash(a.ctypes.data, len(a)))


@pytest.fixture(scope="module")
def setup():
    w = B.Workload(21, 6, 1500, 1)
    keys = w.keys()
    eng = B.Engine(device=0).rebuild(keys)
    tn = w.tenants()
    data, off, tt = w.topics(9, 6000)
    topics = [t.decode() for t in unpack(data, off)]
    exp = U.semantic_rows(O.KV(keys), tn, tt, topics)
    yield eng, tn, tt, (data, off), topics, exp
    eng.close()


def test_single_caller_equals_oracle(setup):
    eng, tn, tt, _, topics, exp = setup
    b = eng.batcher()
    for i in (0, 1, 2, 57, 4000):
        rows, epoch = b.match_all(tn[tt[i]], [topics[i]])
        assert rows == [exp[i]] and epoch == eng.info().epoch
    sel = [i for i in range(len(topics)) if tt[i] == 2][:50]  # matchAll(Set<String>) of one tenant
    rows, _ = b.match_all(tn[2], [topics[i] for i in sel])
    assert rows == [exp[i] for i in sel]
    assert b.match_all(tn[0], [])[0] == []
    assert b.match_all("no-such-tenant", ["a/b"])[0] == [[]]  # every topic is a key, also with 0 routes
    st = b.stats()
    assert st.n_batches == st.n_requests and st.n_topics == 5 + len(sel) + 1  # one caller: every request runs alone
    b.close()


def test_many_native_threads_singleton_calls(setup):
    eng, tn, tt, packed, topics, exp = setup
    b = eng.batcher()
    cnt, hsh, sec = b.drive_singletons(tn, tt, packed, n_threads=64)
    assert cnt.tolist() == [len(r) for r in exp]
    assert hsh.tolist() == [_row_hash(r) for r in exp]
    st = b.stats()
    assert st.n_requests == len(topics) and st.n_topics == len(topics)
    assert st.n_batches < st.n_requests and st.max_batch_topics > 1  # calls really were collected into shared launches
    assert st.max_batch_topics <= 64  # a blocked caller has one request in flight
    b.close()


def test_async_submit_native_threads(setup):
    """bmq_batcher_submit: nobody blocks on the GPU, a dispatcher thread matches whatever has been collected; every
    callback must deliver exactly the oracle's row, and batches are far larger than the number of submitting threads."""
    eng, tn, tt, packed, topics, exp = setup
    b = eng.batcher()
    cnt, hsh, sec = b.drive_singletons(tn, tt, packed, n_threads=4, asynchronous=True)
    assert cnt.tolist() == [len(r) for r in exp]
    assert hsh.tolist() == [_row_hash(r) for r in exp]
    st = b.stats()
    assert st.n_requests == len(topics) and st.n_batches < len(topics) // 8 and st.max_batch_topics > 4
    b.close()


def test_async_submit_python_callbacks_and_drain_on_close(setup):
    eng, tn, tt, _, topics, exp = setup
    b = eng.batcher(max_batch_topics=16)  # small batches: submitters feel the back-pressure
    got = {}
    done = threading.Event()
    n = 300

    def on_done_for(i):
        def f(status, ids, epoch):
            got[i] = (status, ids, epoch)
            if len(got) == n:
                done.set()
        return f

    for i in range(n):
        b.submit(tn[tt[i]], topics[i], on_done_for(i))
    b.close()  # matches and calls back everything submitted before it
    assert done.is_set() and len(got) == n
    ep = eng.info().epoch
    assert all(got[i] == (0, exp[i], ep) for i in range(n))
    with pytest.raises(Exception):
        b.submit(tn[0], "a", lambda *a: None)


def test_python_threads_while_routes_change(setup):
    """Callers keep matching while the apply thread mutates routes (DistWorkerCoProc.java:188-209): rows always belong to
    ONE epoch -- the one reported with them -- because batch and epoch are read under one hold of the engine lock."""
    eng, tn, tt, _, topics, exp = setup
    b = eng.batcher(max_batch_topics=8)
    extra = [B.route_key_from_mqtt(tn[0], "zz/%d" % i, O.receiver_url(0, "x%d" % i, "d")) for i in range(24)]  # match nothing above
    errors, epochs = [], set()

    def caller(k):
        try:
            for i in range(k, 600, 6):
                rows, epoch = b.match_all(tn[tt[i]], [topics[i]])
                assert rows[0] == exp[i]  # ids are stable handles: mutations elsewhere do not move them
                epochs.add(epoch)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    def mutator():
        try:
            for i in range(6):
                eng.apply([(0, k) for k in extra[i * 4:(i + 1) * 4]])
                eng.apply([(1, k) for k in extra[i * 4:(i + 1) * 4]])
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    th = [threading.Thread(target=caller, args=(k,)) for k in range(6)] + [threading.Thread(target=mutator)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(epochs) > 1 and b.stats().max_batch_topics <= 8
    rows, epoch = b.match_all(tn[tt[5]], [topics[5]])
    assert rows == [exp[5]] and epoch == eng.info().epoch
    b.close()


def test_small_output_buffer_reports_needed(setup):
    eng, tn, tt, _, topics, exp = setup
    b = eng.batcher()
    i = int(np.argmax([len(r) for r in exp]))
    assert len(exp[i]) > 1
    t = tn[tt[i]].encode() if isinstance(tn[tt[i]], str) else tn[tt[i]]
    pdata, poff = B.pack([topics[i]])
    row = np.zeros(2, dtype=np.uint32)
    ids = np.zeros(1, dtype=np.uint32)
    import ctypes as C
    need, epoch = C.c_uint64(), C.c_uint64()
    rc = _lib.lib().bmq_batcher_match_all(b.h, t, len(t), pdata.ctypes.data, poff.ctypes.data, 1, row.ctypes.data, ids.ctypes.data, 1,
                                          C.byref(need), C.byref(epoch))
    assert rc == -3 and need.value == len(exp[i]) and row.tolist() == [0, len(exp[i])]
    b.close()


def test_identical_requests_of_one_launch_share_a_row(setup):
    """In-batch dedup: the same (tenant, topic) asked for by several callers at the same moment is matched once
    per launch; every caller still gets the oracle's row.  Driven through the asynchronous side, where one batch collects many
    requests: 4000 requests over 40 distinct (tenant, topic) pairs."""
    eng, tn, tt, packed, topics, exp = setup
    import threading
    b = eng.batcher()
    pick = list(range(0, 400, 10))
    got, done = {}, threading.Semaphore(0)
    N = 4000

    def submit(k):
        i = pick[k % len(pick)]

        def cb(status, ids, epoch):
            got[k] = (status, ids)
            done.release()
        b.submit(tn[tt[i]], topics[i], cb)

    for k in range(N):
        submit(k)
    for _ in range(N):
        assert done.acquire(timeout=60)
    for k in range(N):
        i = pick[k % len(pick)]
        assert got[k] == (0, exp[i]), k
    st = b.stats()
    assert st.n_requests == N and st.n_topics == N and st.n_deduped > N // 2  # most requests rode on another request's row
    # blocking side: one caller asking for the same topic three times in one call
    rows, _ = b.match_all(tn[tt[0]], [topics[0], topics[5] if tt[5] == tt[0] else topics[0], topics[0]])
    assert rows[0] == exp[0] and rows[2] == exp[0]
    assert b.stats().n_deduped > st.n_deduped
    b.close()


# ---- the route cache in front of the batching front (bmq_route_cache_*) ------------------------------------
def test_route_cache_get_hit_invalidate_against_oracle(setup):
    """ISubscriptionCache.get / isCached / refresh on the real engine: a miss is a GPU match through the batching front, a hit never
    reaches the engine, a route mutation drops exactly the cached topics its filter matches, and what a get returns afterwards is the
    oracle's row on the updated key set."""
    eng0, tn, tt, _, topics, exp = setup
    w = B.Workload(21, 6, 1500, 1)
    keys = w.keys()
    eng = B.Engine(device=0).rebuild(keys)  # own engine: this test mutates the index
    b = eng.batcher()
    c = B.RouteCache(b)
    idx = list(range(0, 3000, 7))
    for i in idx:
        ids, ep = c.get(tn[tt[i]], topics[i], now_ms=5)
        assert ids == exp[i] and ep == eng.info().epoch
    n_batches = b.stats().n_batches
    for i in idx:
        assert c.get(tn[tt[i]], topics[i], now_ms=6)[0] == exp[i]
    st = c.stats()
    assert b.stats().n_batches == n_batches  # second round: hits only
    distinct = len({(int(tt[i]), topics[i]) for i in idx})
    assert st.misses == distinct and st.hits == 2 * len(idx) - distinct and st.entries == distinct
    # mutate: a '#' subscription under tenant 0's most common first level, and the unsubscribe of one existing route
    t0 = tn[0]
    lv0 = topics[[i for i in idx if tt[i] == 0][0]].split("/")[0]
    new_key = B.route_key(t0, lv0 + "/#", 1, "0\0cacheTest\0d1")
    gone = keys[[i for i, k in enumerate(keys) if B.decode_route_key(k)[1] == t0][0]]
    assert c.is_cached(t0, lv0 + "/#")
    c.apply([(0, new_key), (1, gone)])
    keys2 = sorted((set(keys) | {new_key}) - {gone})
    assert not c.is_cached(t0, lv0 + "/#")  # every cached topic under lv0 was dropped
    assert c.stats().invalidations >= 1
    kv2 = O.KV(keys2)
    sel = [i for i in idx if tt[i] == 0]
    want = U.semantic_rows(kv2, tn, tt[sel], [topics[i] for i in sel])
    for j, i in enumerate(sel):
        ids, ep = c.get(t0, topics[i], now_ms=7)
        got = sorted(keys2.index(k) for k in eng.route_keys(np.array(ids, dtype=np.uint32))) if ids else []
        assert got == want[j], topics[i]
        # reloaded (its topic is under lv0: matched by the new filter) or still the entry of the first round
        assert ep == (eng.info().epoch if topics[i].split("/")[0] == lv0 and not topics[i].startswith("$") else ep) and ep <= eng.info().epoch
    # other tenants' entries survived the mutation
    other = [i for i in idx if tt[i] != 0][:50]
    nb = b.stats().n_batches
    for i in other:
        assert c.get(tn[tt[i]], topics[i], now_ms=8)[0] == exp[i]
    assert b.stats().n_batches == nb
    # rebuild through the cache: a new generation, nothing old is served
    c.rebuild(keys2)
    assert c.stats().entries == 0
    for j, i in enumerate(sel[:40]):
        assert c.get(t0, topics[i], now_ms=9)[0] == want[j]  # ids are ranks again
    c.close()
    b.close()
    eng.close()


def test_route_cache_native_threads_two_passes(setup):
    """64 native threads through bmq_route_cache_get, twice over the batch: pass one loads (misses collected into shared launches by the
    batching front), pass two is served from the host; both passes give exactly the oracle's rows."""
    eng, tn, tt, packed, topics, exp = setup
    b = eng.batcher()
    c = B.RouteCache(b, max_routes_per_tenant=10_000_000)
    cnt, hsh, sec = c.drive(tn, tt, packed, n_threads=64, passes=2)
    assert cnt.tolist() == [len(r) for r in exp]
    assert hsh.tolist() == [_row_hash(r) for r in exp]
    st = c.stats()
    distinct = len({(int(tt[i]), topics[i]) for i in range(len(topics))})
    assert st.entries == distinct and st.hits >= len(topics)  # the whole second pass hit
    assert st.misses >= distinct  # concurrent first requests of one topic may both load
    c.close()
    b.close()


def test_route_cache_get_async_futures(setup):
    """bmq_route_cache_get_async: a miss completes from the batcher's dispatcher thread (and is cached on the way), a hit completes
    inline before the call returns; every completion carries exactly the oracle's row."""
    eng, tn, tt, _, topics, exp = setup
    b = eng.batcher()
    c = B.RouteCache(b)
    idx = list(range(0, 2000, 3))
    got, lock, done = {}, threading.Lock(), threading.Event()

    def on_done_for(i):
        def f(status, ids, epoch):
            with lock:
                got.setdefault(i, []).append((status, ids, epoch, threading.get_ident()))
                if sum(len(v) for v in got.values()) == len(idx):
                    done.set()
        return f
    for i in idx:
        c.get_async(tn[tt[i]], topics[i], on_done_for(i), now_ms=1)
    assert done.wait(30)
    me = threading.get_ident()
    for i in idx:
        (status, ids, epoch, _), = got[i]
        assert status == 0 and ids == exp[i] and epoch == eng.info().epoch
    assert any(tid != me for v in got.values() for (_, _, _, tid) in v)  # misses completed on the dispatcher thread
    # everything is cached now: the second round completes inline, on this thread, without a launch
    nb = b.stats().n_batches
    got.clear()
    done.clear()
    for i in idx:
        c.get_async(tn[tt[i]], topics[i], on_done_for(i), now_ms=2)
        assert got[i][-1][3] == me and got[i][-1][1] == exp[i]  # already called back when get_async returns
    assert b.stats().n_batches == nb and c.stats().hits >= len(idx)
    c.close()
    b.close()


def test_route_cache_get_batch_one_launch_for_all_misses(setup):
    """bmq_route_cache_get_batch: a whole BatchDistRequest (all tenants) in one call -- every miss of the request, identical ones once,
    travels in ONE launch; the second call is served from the cache; rows equal the oracle either way, also after a mutation."""
    eng0, tn, tt, _, topics, exp = setup
    w = B.Workload(21, 6, 1500, 1)
    keys = w.keys()
    eng = B.Engine(device=0).rebuild(keys)
    b = eng.batcher()
    c = B.RouteCache(b, max_routes_per_tenant=10_000_000)
    sel = list(range(0, 6000, 2))
    row, ids, hit = c.get_batch(tn, tt[sel], [topics[i] for i in sel], now_ms=1)
    assert U.csr_rows(row, ids) == [exp[i] for i in sel]
    assert b.stats().n_batches == 1 and not hit.any()  # one launch carried everything
    distinct = len({(int(tt[i]), topics[i]) for i in sel})
    assert b.stats().n_topics == distinct and c.stats().entries == distinct  # identical (tenant, topic) pairs matched once
    row2, ids2, hit2 = c.get_batch(tn, tt[sel], [topics[i] for i in sel], now_ms=2)
    assert hit2.all() and b.stats().n_batches == 1 and (row2 == row).all() and (ids2 == ids).all()
    # a '#' subscription under one first level of tenant 0: those rows are re-matched (one more launch), the rest still hit
    t0 = tn[0]
    lv0 = topics[[i for i in sel if tt[i] == 0][0]].split("/")[0]
    new_key = B.route_key(t0, lv0 + "/#", 1, "0\0batchTest\0d2")
    c.apply([(0, new_key)])
    keys2 = sorted(set(keys) | {new_key})
    row3, ids3, hit3 = c.get_batch(tn, tt[sel], [topics[i] for i in sel], now_ms=3)
    assert b.stats().n_batches == 2 and hit3.any() and not hit3.all()
    want = U.semantic_rows(O.KV(keys2), tn, tt[sel], [topics[i] for i in sel])
    got = U.rows_as_ranks(eng, row3, ids3, keys2)
    assert got == want
    for j, i in enumerate(sel):
        affected = tt[i] == 0 and topics[i].split("/")[0] == lv0 and not topics[i].startswith("$")
        assert bool(hit3[j]) == (not affected), topics[i]
    c.close()
    b.close()
    eng.close()


def test_persistent_matcher_serves_the_small_generations(setup):
    """Round 6: generations of at most 64 topics are not launched -- resident waves (k_poll, bmq_poll_kernel.h) find them in a request ring
    and answer in place.  Every caller still gets exactly the oracle's rows; the poller's counters say it was the one that answered; a
    mutation stops it (the index never changes under it) and the next generation starts it again; deep topics are handed back to a launch."""
    eng, tn, tt, packed, topics, exp = setup
    eng.poller_control(eng.POLLER_ENABLE)
    b = eng.batcher()
    s0 = eng.poller_stats()
    assert s0.enabled
    for i in (0, 1, 2, 57, 4000):
        rows, epoch = b.match_all(tn[tt[i]], [topics[i]])
        assert rows == [exp[i]] and epoch == eng.info().epoch
    s1 = eng.poller_stats()
    assert s1.n_served - s0.n_served >= 5 and s1.n_starts >= 1 and s1.n_timeouts == 0
    cnt, hsh, sec = b.drive_singletons(tn, tt, packed, n_threads=32)
    assert cnt.tolist() == [len(r) for r in exp]
    assert hsh.tolist() == [_row_hash(r) for r in exp]
    s2 = eng.poller_stats()
    assert s2.n_served - s1.n_served > 100 and s2.n_timeouts == 0  # the generations of 32 blocked callers fit a wave
    # a topic of more than 16 levels: the wave hands the generation back, its leader launches it (k_walk_slow is a launch's business)
    deep = "/".join(["x"] * 40)
    rows, _ = b.match_all(tn[0], [deep])
    assert rows == U.semantic_rows(O.KV(eng.route_keys(np.arange(int(eng.info().next_route_id), dtype=np.uint32))), tn, [0], [deep])
    assert eng.poller_stats().n_fallback > s2.n_fallback
    # a mutation stops the poller; afterwards the new route is seen (by a new launch of it)
    starts = eng.poller_stats().n_starts
    key = B.route_key(tn[1], "poller/+/probe", 1, "0\0px\0d1")
    eng.apply([(0, key)])
    assert not eng.poller_stats().running
    rows, epoch = b.match_all(tn[1], ["poller/x/probe"])
    rid = rows[0]
    assert len(rid) == 1 and eng.route_key(rid[0]) == key and epoch == eng.info().epoch
    assert eng.poller_stats().n_starts > starts
    eng.apply([(1, key)])
    assert b.match_all(tn[1], ["poller/x/probe"])[0] == [[]]
    # switched off: the same answers through launches
    eng.poller_control(eng.POLLER_DISABLE)
    served = eng.poller_stats().n_served
    rows, _ = b.match_all(tn[tt[7]], [topics[7]])
    assert rows == [exp[7]] and eng.poller_stats().n_served == served
    eng.poller_control(eng.POLLER_ENABLE)
    b.close()


def test_a_wedged_poller_costs_a_time_out_not_an_answer():
    """The test hook makes the resident waves see the doorbells and not answer: the first generation waits out its time-out (250 ms), the
    poller is told to leave and is off for good on this engine, and the generation -- like every later one -- is launched: the caller
    sees its rows, bmq_poller_stats.n_timeouts says what happened."""
    w = B.Workload(22, 2, 300, 1)
    keys = w.keys()
    eng = B.Engine(device=0).rebuild(keys)
    try:
        tn = w.tenants()
        data, off, tt = w.topics(5, 40)
        topics = [t.decode() for t in unpack(data, off)]
        exp = U.semantic_rows(O.KV(keys), tn, tt, topics)
        b = eng.batcher()
        assert b.match_all(tn[tt[0]], [topics[0]])[0] == [exp[0]]
        assert eng.poller_stats().n_served >= 1
        eng.poller_control(eng.POLLER_TEST_IGNORE_DOORBELLS)
        assert b.match_all(tn[tt[1]], [topics[1]])[0] == [exp[1]]  # (after the time-out, through a launch)
        st = eng.poller_stats()
        assert st.n_timeouts == 1 and not st.enabled and not st.running
        for i in range(2, 10):
            assert b.match_all(tn[tt[i]], [topics[i]])[0] == [exp[i]]
        assert eng.poller_stats().n_timeouts == 1
        b.close()
    finally:
        eng.close()
