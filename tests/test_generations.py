"""bifromq_amd/generations.py: compaction beside the serving index (two engine handles, ops logged and replayed, swap).  Host-only engines
here (device = -1: the index and its builder run on the host executor, nothing is matched): what is checked is the bookkeeping -- the key set
of every generation, the log, the swap, the lifetime of a retired generation.  Matching through a pinned engine is the ordinary path of the
-m gpu tests."""
import threading

import numpy as np
import pytest

import bifromq_amd as B
from bifromq_amd.generations import GenerationalEngine


def _key(i, t=None):
    return B.route_key("tenant%d" % (i % 5 if t is None else t), ["a/%d/+", "b/%d/#", "%d/x", "+/%d"][i % 4] % i, 1, "0\0inbox%d\0d%d" % (i, i % 3))


def _live(eng):
    n = int(eng.info().next_route_id)
    return sorted(k for k in eng.route_keys(np.arange(n, dtype=np.uint32)) if k)


def test_compaction_beside_the_serving_generation_with_mutations_at_every_step():
    g = GenerationalEngine(device=-1)
    model = set(_key(i) for i in range(400))
    g.rebuild(sorted(model))
    serial = [1000]

    def churn(n_add, n_del):
        def f():
            dels = sorted(model)[:n_del]
            adds = [_key(serial[0] + j) for j in range(n_add)]
            serial[0] += n_add
            # a delete of a key, its re-insertion and a second delete in ONE stream: the replay must keep the order
            ops = [(1, k) for k in dels] + [(0, k) for k in adds] + [(0, dels[0]), (1, dels[0])]
            g.apply(ops)
            model.difference_update(dels)
            model.update(adds)
        return f

    # garbage first: deletes + re-adds leave dead ids behind
    churn(50, 120)()
    with g.pin() as (eng, gen):
        info = eng.info()
        assert gen == 0 and info.n_routes == len(model) and info.next_route_id > info.n_routes
    g.hooks = {"after_snapshot": churn(7, 5), "after_export": churn(11, 13), "after_build": churn(3, 2), "after_replay_round": churn(2, 1)}
    rounds = []
    old = {}
    with g.pin() as (eng_a, gen_a):  # a caller that is still inside generation 0 while the swap happens
        ids_a = np.arange(10, dtype=np.uint32)
        old["keys"] = eng_a.route_keys(ids_a)
        t = threading.Thread(target=lambda: rounds.append(g.compact_online()))
        t.start()
        t.join(120)
        assert not t.is_alive() and rounds
        assert eng_a.route_keys(ids_a) == old["keys"]      # the retired generation still answers for its own ids
        assert g.generation == 1
    r = rounds[0]
    assert r["generation"] == 1 and r["ops_replayed"] > 0 and r["replay_rounds"] >= 1
    assert r["replay_rounds"] <= GenerationalEngine.MAX_REPLAY_ROUNDS and r["ops_replayed_under_lock"] <= r["ops_replayed"]
    with g.pin() as (eng, gen):
        info = eng.info()
        assert gen == 1 and _live(eng) == sorted(model)
        assert info.n_routes == len(model)
        # the new generation carries at most the garbage of the replayed ops, not the old generation's
        assert info.next_route_id - info.n_routes <= r["ops_replayed"]
    assert eng_a.h is None                                   # generation 0 was closed when its last pin went
    # and again, without mutations meanwhile: dense ids, nothing replayed
    g.hooks = {}
    r2 = g.compact_online()
    with g.pin() as (eng, gen):
        info = eng.info()
        assert gen == 2 and r2["ops_replayed"] == 0 and info.next_route_id == info.n_routes == len(model) and _live(eng) == sorted(model)
    g.close()


def test_a_mutator_that_never_pauses_cannot_hold_the_swap_off():
    g = GenerationalEngine(device=-1)
    base = [_key(i) for i in range(300)]
    g.rebuild(sorted(base))
    added = []
    g.hooks = {"after_replay_round": lambda: (g.apply([(0, _key(5000 + len(added)))]), added.append(1))}
    g.hooks["after_build"] = g.hooks["after_replay_round"]
    r = g.compact_online()
    assert r["replay_rounds"] == GenerationalEngine.MAX_REPLAY_ROUNDS and r["ops_replayed_under_lock"] == 1   # the last op was replayed under the lock
    with g.pin() as (eng, gen):
        assert _live(eng) == sorted(base + [_key(5000 + j) for j in range(len(added))])
    g.close()


def test_a_failed_build_leaves_the_serving_generation_alone():
    g = GenerationalEngine(device=-1)
    g.rebuild(sorted(_key(i) for i in range(50)))

    def boom():
        raise RuntimeError("injected")

    g.hooks = {"after_export": boom}
    with pytest.raises(RuntimeError, match="injected"):
        g.compact_online()
    assert g.generation == 0
    g.apply([(0, _key(777))])                                 # not logged any more, still served
    with g.pin() as (eng, gen):
        assert gen == 0 and eng.info().n_routes == 51
    g.hooks = {}
    assert g.compact_online()["generation"] == 1               # and a later compaction goes through
    g.close()


def test_a_handle_that_also_holds_retained_topics_is_refused():
    g = GenerationalEngine(device=-1)
    g.rebuild(sorted(_key(i) for i in range(10)))
    with g.pin() as (eng, _):
        eng.retain_rebuild(["tenant0"], [0, 0], [b"a/b", b"a/c"])
    with pytest.raises(NotImplementedError):
        g.compact_online()
    assert g.generation == 0
    g.apply([(0, _key(99))])          # nothing was left half-done: not logging, still serving
    with g.pin() as (eng, _):
        assert eng.info().n_routes == 11
    g.close()


@pytest.mark.gpu
def test_matching_through_the_generations_on_the_gpu():
    """Two engine handles on one GPU: rows matched through generation 0 (garbage inside), through generation 1 after the swap -- with mutations
    landing while it was built -- equal the semantic oracle key for key; a caller still inside generation 0 during the swap gets its rows."""
    from oracle import oracle as O
    from tests import util as U

    w = B.Workload(0xB1F20041, 8, 300, 1)
    keys = w.keys()
    tn = w.tenants()
    data, off, tt = w.topics(5, 1500)
    topics = [bytes(data[off[i]:off[i + 1]]) for i in range(len(tt))]
    g = GenerationalEngine(device=0)
    g.rebuild(keys)
    model = set(keys)
    rng = np.random.default_rng(5)

    def churn(n):
        def f():
            ks = sorted(model)
            dels = [ks[int(i)] for i in rng.choice(len(ks), size=n, replace=False)]
            adds = [B.route_key(tn[int(rng.integers(0, len(tn)))], "gen/%d/+" % int(rng.integers(0, 1 << 30)), 1, "0\0g%d\0d" % j) for j in range(n)]
            g.apply([(1, k) for k in dels] + [(0, k) for k in adds])
            model.difference_update(dels)
            model.update(adds)
        return f

    def rows_as_keys(eng):
        row, ids = eng.match_batch(tn, tt, topics)
        ks = eng.route_keys(ids)
        return [sorted(ks[row[i]:row[i + 1]]) for i in range(len(tt))]

    def expected():
        kv = O.KV(sorted(model))
        ks = sorted(model)
        return [sorted(ks[x] for x in r) for r in U.semantic_rows(kv, tn, tt, topics)]

    churn(200)()
    with g.pin() as (eng, gen):
        got0 = rows_as_keys(eng)
    assert got0 == expected()
    g.hooks = {"after_export": churn(50), "after_build": churn(30)}
    with g.pin() as (eng_a, gen_a):
        t = threading.Thread(target=g.compact_online)
        t.start()
        t.join(300)
        assert not t.is_alive() and g.generation == 1
        assert gen_a == 0 and eng_a.h is not None and len(rows_as_keys(eng_a)) == len(tt)   # generation 0 still serves whoever is inside it
    with g.pin() as (eng, gen):
        info = eng.info()
        assert gen == 1 and info.n_routes == len(model)
        got1 = rows_as_keys(eng)
    assert got1 == expected()
    g.close()


def test_compaction_entry_points_refuse_what_they_cannot_do():
    """_poll / _swap without a compaction running are state errors, _abort is idempotent, _begin needs an index."""
    eng = B.Engine(device=-1)
    try:
        with pytest.raises(B.BmqError) as ei:
            eng.compact_begin()
        assert ei.value.code == -7
        eng.rebuild(sorted(_key(i) for i in range(20)))
        for f in (eng.compact_poll, eng.compact_swap):
            with pytest.raises(B.BmqError) as ei:
                f()
            assert ei.value.code == -7
        eng.compact_abort().compact_abort()
    finally:
        eng.close()


def test_a_refused_batch_during_a_compaction_is_not_replayed():
    """a mutation batch entered the compaction's log as soon as apply_begin had returned -- a batch later refused for a malformed
    key (or failing asynchronously) stayed in it and bmq_compact_swap's replay failed for ever.  The log now takes a batch only once its
    outcome is known: refused batches, blocking and async, leave no trace, the valid ones around them are replayed."""
    eng = B.Engine(device=-1)
    try:
        model = set(_key(i) for i in range(300))
        eng.rebuild(sorted(model))
        eng.apply([(1, k) for k in sorted(model)[:40]])  # garbage for the compaction to drop
        model.difference_update(sorted(model)[:40])
        eng.compact_begin()
        good1 = [_key(1000 + j) for j in range(25)]
        eng.apply([(0, k) for k in good1])
        with pytest.raises(B.BmqError) as ei:  # refused as a whole, before anything is changed
            eng.apply([(0, _key(5000)), (0, b"\x07garbage")])
        assert ei.value.code == -1
        eng.apply_async([(0, _key(6000)), (1, b"\x00\x00")])  # the same through the asynchronous form: the error is the wait's
        with pytest.raises(B.BmqError) as ei:
            eng.apply_wait()
        assert ei.value.code == -1
        good2 = [_key(2000 + j) for j in range(10)]
        eng.apply_async([(0, k) for k in good2] + [(1, good1[0])])
        done = 0
        while done < 1000:
            done = eng.compact_poll(64)
        carried, replayed = eng.compact_swap()  # (r5: BMQ_E_STATE 'malformed route key' on every call from here on)
        model.update(good1)
        model.update(good2)
        model.discard(good1[0])
        assert carried == 260 and replayed == len(good1) + len(good2) + 1
        assert _live(eng) == sorted(model)
        info = eng.info()
        # (the replayed put + delete of good1[0] used up one id of the new generation)
        assert info.n_routes == len(model) and info.next_route_id == carried + len(good1) + len(good2)
    finally:
        eng.close()


def _compaction_inside_the_engine(device, n_tenants, per_tenant, n_topics, chunk):
    """bmq_compact_begin / _poll / _swap through ctypes: the next generation is built inside ONE handle (on the device: from keys that never
    leave it), mutation batches (blocking and async) -- and on a GPU match batches -- land between the polls, and after the swap the key set
    and the matched rows equal the model / the semantic oracle key for key, the garbage is gone and the ids are dense again (+ what was
    replayed).  bmq_rebuild / bmq_compact are refused meanwhile.  device = -1: the same code over the host executor, nothing is matched."""
    from oracle import oracle as O
    from tests import util as U

    w = B.Workload(0xB1F20051, n_tenants, per_tenant, 1)
    keys = w.keys()
    tn = w.tenants()
    data, off, tt = w.topics(5, n_topics)
    topics = [bytes(data[off[i]:off[i + 1]]) for i in range(len(tt))]
    eng = B.Engine(device=device)
    try:
        eng.rebuild(keys)
        model = set(keys)
        rng = np.random.default_rng(7)
        serial, progress = [0], [-1]

        def churn(n, async_=False):
            ks = sorted(model)
            dels = [ks[int(i)] for i in rng.choice(len(ks), size=n, replace=False)]
            adds = [B.route_key(tn[int(rng.integers(0, len(tn)))], "gen/%d/+" % (serial[0] + j), 1, "0\0g%d\0d" % j) for j in range(n)]
            adds.append(B.route_key("tenant-born-%d" % serial[0], "x/#", 1, "0\0nb\0d"))   # a tenant the next generation has no region for
            serial[0] += n
            ops = [(1, k) for k in dels] + [(0, k) for k in adds] + [(0, dels[0]), (1, dels[0])]   # order inside one stream matters
            (eng.apply_async if async_ and device >= 0 else eng.apply)(ops)   # (page-locked buffers need a GPU)
            for o, k in ops:
                history.setdefault(k, []).append(("put" if o == 0 else "del", serial[0], "async" if async_ else "blocking", "at %d permille" % progress[0]))
            model.difference_update(dels)
            model.update(adds)

        def live_keys():
            return sorted(k for k in eng.route_keys(np.arange(int(eng.info().next_route_id), dtype=np.uint32)) if k)

        history = {}   # key -> what the test did with it, for the message of a failing comparison

        def check_rows():
            live = live_keys()
            if live != sorted(model):
                extra, missing = sorted(set(live) - model), sorted(model - set(live))
                dup = len(live) - len(set(live))
                raise AssertionError("key set differs: %d extra %d missing %d stored twice; extra: %r; missing: %r" % (
                    len(extra), len(missing), dup, [(k, history.get(k)) for k in extra[:6]], [(k, history.get(k)) for k in missing[:6]]))
            if device < 0:
                return
            row, ids = eng.match_batch(tn, tt, topics)
            ks = eng.route_keys(ids)
            got = [sorted(ks[row[i]:row[i + 1]]) for i in range(len(tt))]
            srt = sorted(model)
            kv = O.KV(srt)
            assert got == [sorted(srt[x] for x in r) for r in U.semantic_rows(kv, tn, tt, topics)]

        churn(len(keys) // 7)             # garbage first: a seventh of the routes deleted, as many new ones
        info0 = eng.info()
        assert info0.next_route_id > info0.n_routes == len(model)
        eng.compact_begin()
        with pytest.raises(B.BmqError) as ei:
            eng.compact_begin()
        assert ei.value.code == -7
        for f in (eng.compact, lambda: eng.rebuild(keys)):
            with pytest.raises(B.BmqError) as ei:
                f()
            assert ei.value.code == -7
        with pytest.raises(B.BmqError) as ei:
            eng.compact_swap()            # not complete yet
        assert ei.value.code == -7
        polls, done = 0, 0
        while done < 1000:
            done = progress[0] = eng.compact_poll(chunk)
            polls += 1
            if polls % 3 == 1:
                churn(len(keys) // 200, async_=(polls % 2 == 0))   # some hit keys already carried over, some keys not yet, some are new
            if polls == 4:
                check_rows()              # the serving generation serves, mutations included
            assert polls < 200
        assert polls >= 8
        carried, replayed = eng.compact_swap()
        info1 = eng.info()
        check_rows()
        assert info1.generation == info0.generation + 1 and info1.n_routes == len(model)
        assert replayed > 0 and carried >= info0.n_routes - replayed
        assert info1.next_route_id - info1.n_routes <= replayed        # at most the replayed ops' garbage, not the old generation's
        check_rows()
        # the generation after the swap is an ordinary index: mutations, another compaction without mutations meanwhile -> dense ids
        churn(len(keys) // 100)
        eng.compact_begin()
        while eng.compact_poll(1 << 20) < 1000:
            pass
        carried2, replayed2 = eng.compact_swap()
        info2 = eng.info()
        assert replayed2 == 0 and carried2 == len(model) == info2.n_routes == info2.next_route_id and info2.garbage_bytes == 0
        assert info2.device_bytes <= info1.device_bytes or device < 0
        check_rows()
        # abort drops the half-built generation; the serving one is untouched
        eng.compact_begin()
        eng.compact_poll(1000)
        eng.compact_abort()
        assert eng.info().generation == info2.generation
        check_rows()
    finally:
        eng.close()


def test_compaction_inside_the_engine_over_the_host_executor():
    _compaction_inside_the_engine(-1, 6, 400, 10, 250)


@pytest.mark.gpu
def test_compaction_inside_the_engine_between_batches_and_mutations():
    _compaction_inside_the_engine(0, 12, 2500, 3000, 3000)
