"""Host-visible result formats of the asynchronous match (include/bmq.h BMQ_FMT_*): fan-out counts only (what
DistWorkerCoProc.batchDist replies with, DW/DistWorkerCoProc.java:535-538), matched id ranges, pairs grouped by DelivererKey
(DW/DeliverExecutorGroup.java:112-241).  Every format is compared with the ORACLE's rows of the same batch (the semantic oracle over every
key of the row's tenant; the grouped pairs with oracle.fanout_groups), not with the engine's own id CSR."""
import numpy as np
import pytest

import bifromq_amd as B
from bifromq_amd.engine import pinned
from oracle import oracle as O
from bifromq_amd.workload import unpack
from tests import util as U
from tests.test_fanout import _check as check_groups_vs_oracle

pytestmark = pytest.mark.gpu


def _oracle_csr(keys_sorted, tn, tt, packed, id_of_rank=None):
    """(row_ptr, ids) the semantic oracle expects for the batch: ranks of the matching keys of each row's tenant, mapped to engine ids
    (after a rebuild the id IS the rank; after churn `id_of_rank` maps) and ascending per row"""
    kv = O.KV(keys_sorted)
    res, _ = kv.match_semantic_batch(tn, np.asarray(tt, dtype=np.uint32), packed, threads=U.host_threads())
    rp = res.row_ptr.astype(np.int64)
    ids = res.routes.astype(np.int64)
    if id_of_rank is not None:
        ids = U.csr_sorted(rp, np.asarray(id_of_rank, dtype=np.int64)[ids])
    return rp.astype(np.uint32), ids.astype(np.uint32)


@pytest.fixture(scope="module")
def eng():
    e = B.Engine(device=0)
    yield e
    e.close()


def _normal(tenant, tf, broker, recv, deliverer):
    return B.route_key_from_mqtt(tenant, tf, O.receiver_url(broker, recv, deliverer))


def _batch(w, seed, n):
    data, off, tt = w.topics(seed, n)
    pd, po, pt = pinned(len(data) + 16, np.uint8), pinned(len(off), np.uint32), pinned(len(tt), np.uint32)
    pd[:len(data)], po[:], pt[:] = data, off, tt
    return (data, off, tt), (pd, po, pt)


def _check_ranges(eng, p_t, p_to, n_tenants, pinned_batch, n, erow, eids, expect_overlap=None):
    pd, po, pt = pinned_batch
    t = eng.match_submit_fmt(p_t, p_to, n_tenants, pt, pd, po, n, eng.FMT_RANGES)
    rptr, row = pinned(n + 1, np.uint32), pinned(n + 1, np.uint32)
    ranges, side = pinned((max(len(eids), 16), 2), np.uint32), pinned(max(len(eids), 16), np.uint32)
    info = eng.match_wait_ranges(t, rptr, ranges, side, row)
    assert info.n_ids == len(eids) and (row == erow).all()
    assert rptr[0] == 0 and rptr[n] == info.n_ranges and (np.diff(rptr.astype(np.int64)) >= 0).all()
    rows = eng.expand_ranges(rptr, ranges, side, n)
    for i in range(n):
        assert np.array_equal(rows[i], eids[erow[i]:erow[i + 1]]), i
    if expect_overlap is not None:
        assert (info.n_overlapping_rows > 0) == expect_overlap
    return info


def test_counts_ranges_grouped_equal_the_id_csr(eng):
    w = B.Workload(0xF0A7, 12, 2500, 1)
    keys = w.keys()
    eng.rebuild(keys)
    tn = w.tenants()
    tdata, toff = w.tenants_packed()
    p_t, p_to = pinned(len(tdata) + 16, np.uint8), pinned(len(toff), np.uint32)
    p_t[:len(tdata)], p_to[:] = tdata, toff
    n = 30000
    (data, off, tt), pb = _batch(w, 7, n)
    pd, po, pt = pb
    erow, eids = _oracle_csr(sorted(keys), tn, tt, (data, off))  # what every format below must carry: the ORACLE's rows
    grow, gids = eng.match_batch(tn, tt, packed_topics=(data, off))
    assert np.array_equal(grow, erow) and np.array_equal(gids, eids)
    # ---- COUNTS: the row pointers, nothing else; the wrong wait is refused and leaves the ticket in flight
    t = eng.match_submit_fmt(p_t, p_to, len(tn), pt, pd, po, n, eng.FMT_COUNTS)
    row = pinned(n + 1, np.uint32)
    with pytest.raises(B.BmqError):
        eng.match_wait(t, row, pinned(16, np.uint32))
    assert eng.match_wait_counts(t, row) == len(eids)
    assert (row == erow).all()  # every fan-out count equals the oracle's
    # ---- RANGES: expanding gives the id rows, in order; fewer ranges than ids; nothing indirect after a rebuild
    info = _check_ranges(eng, p_t, p_to, len(tn), pb, n, erow, eids, expect_overlap=False)
    assert info.n_side_ids == 0 and 0 < info.n_ranges < len(eids)
    # too small a range buffer: NOSPACE with the sizes, the ticket is released, a second try fits
    t = eng.match_submit_fmt(p_t, p_to, len(tn), pt, pd, po, n, eng.FMT_RANGES)
    with pytest.raises(B.BmqError) as ei:
        eng.match_wait_ranges(t, pinned(n + 1, np.uint32), pinned((4, 2), np.uint32), pinned(4, np.uint32))
    assert ei.value.code == -3 and ei.value.info.n_ranges == info.n_ranges
    # ---- GROUPED: the oracle's DeliverExecutorGroup.submit -> BatchDeliveryCall.add restatement over the oracle's rows
    t = eng.match_submit_fmt(p_t, p_to, len(tn), pt, pd, po, n, eng.FMT_GROUPED)
    ot, orr = pinned(len(eids) + 8, np.uint32), pinned(len(eids) + 8, np.uint32)
    goff, grep = pinned(4096, np.uint32), pinned(4095, np.uint32)
    total, ng, special = eng.match_wait_grouped(t, ot, orr, goff, grep)
    assert total == len(eids)
    ks = sorted(keys)
    check_groups_vs_oracle(eng, U.csr_rows(erow, eids), lambda i: ks[i], (ot[:total], orr[:total], goff[:ng + 1], grep[:ng], special))
    # two formats in flight at once, waited for in the other order
    t0 = eng.match_submit_fmt(p_t, p_to, len(tn), pt, pd, po, n, eng.FMT_COUNTS)
    t1 = eng.match_submit_fmt(p_t, p_to, len(tn), pt, pd, po, n, eng.FMT_IDS)
    ids = pinned(len(eids) + 8, np.uint32)
    assert eng.match_wait(t1, row, ids) == len(eids) and (ids[:len(eids)] == eids).all()
    row[:] = 0
    assert eng.match_wait_counts(t0, row) == len(eids) and (row == erow).all()


def test_ranges_after_churn_carry_their_side_lists(eng):
    """Filters touched by bmq_routes_apply own id LISTS (RANGE_INDIRECT): their ids travel in the side array of the result, and rows whose
    ranges interleave are reported (the consumer orders those)."""
    w = B.Workload(0xF0A8, 6, 1500, 1)
    keys = w.keys()
    eng.rebuild(keys)
    tn = w.tenants()
    tdata, toff = w.tenants_packed()
    p_t, p_to = pinned(len(tdata) + 16, np.uint8), pinned(len(toff), np.uint32)
    p_t[:len(tdata)], p_to[:] = tdata, toff
    n = 8000
    (data, off, tt), pb = _batch(w, 3, n)
    ops = []
    for t in tn:
        for j in range(40):
            ops.append((0, _normal(t, "#", 0, "in%d" % j, "d%d" % (j % 3))))
            ops.append((0, _normal(t, "+/#", 1, "px%d" % j, "d0")))
    for k in keys[::7]:
        ops.append((1, k))
    eng.apply(ops)
    # the oracle's rows on the UPDATED key set; ranks -> engine ids
    expect_live = set(keys)
    for op, k in ops:
        (expect_live.add if op == 0 else expect_live.discard)(k)
    nid = eng.info().next_route_id  # the id <-> key mapping is the engine's key store (bmq_route_keys); WHICH keys match is the oracle's call
    id_of = {k: i for i, k in enumerate(eng.route_keys(list(range(nid)))) if k}
    assert set(id_of) == expect_live
    live = sorted(id_of)
    erow, eids = _oracle_csr(live, tn, tt, (data, off), id_of_rank=[id_of[k] for k in live])
    grow, gids = eng.match_batch(tn, tt, packed_topics=(data, off))
    assert np.array_equal(grow, erow) and np.array_equal(gids, eids)
    info = _check_ranges(eng, p_t, p_to, len(tn), pb, n, erow, eids)
    assert info.n_side_ids > 0


def test_formats_on_edge_shapes(eng):
    """no match at all, one topic, a topic of an unknown tenant"""
    eng.rebuild([_normal("t", "a/b", 0, "i", "d")])
    for topics, tenants in (([b"x/y"], [b"t"]), ([b"a/b"], [b"t"]), ([b"a/b", b"a/b"], [b"nobody", b"t"])):
        tdata = np.frombuffer(b"".join(tenants) + b"\0" * 16, dtype=np.uint8).copy()
        toff = np.cumsum([0] + [len(t) for t in tenants]).astype(np.uint32)
        pdata = np.frombuffer(b"".join(topics) + b"\0" * 16, dtype=np.uint8).copy()
        poff = np.cumsum([0] + [len(t) for t in topics]).astype(np.uint32)
        tt = np.arange(len(topics), dtype=np.uint32) % len(tenants)
        n = len(topics)
        erow, eids = eng.match_batch([t.decode() for t in tenants], tt, topics=[t.decode() for t in topics])
        t = eng.match_submit_fmt(tdata, toff, len(tenants), tt, pdata, poff, n, eng.FMT_COUNTS)
        row = np.zeros(n + 1, dtype=np.uint32)
        assert eng.match_wait_counts(t, row) == len(eids) and (row == erow).all()
        t = eng.match_submit_fmt(tdata, toff, len(tenants), tt, pdata, poff, n, eng.FMT_RANGES)
        rptr, ranges, side = np.zeros(n + 1, dtype=np.uint32), np.zeros((8, 2), dtype=np.uint32), np.zeros(8, dtype=np.uint32)
        info = eng.match_wait_ranges(t, rptr, ranges, side)
        rows = eng.expand_ranges(rptr, ranges, side, n)
        assert info.n_ids == len(eids) and all(np.array_equal(rows[i], eids[erow[i]:erow[i + 1]]) for i in range(n))
        t = eng.match_submit_fmt(tdata, toff, len(tenants), tt, pdata, poff, n, eng.FMT_GROUPED)
        ot, orr, goff, grep = (np.zeros(8, dtype=np.uint32) for _ in range(4))
        total, ng, special = eng.match_wait_grouped(t, ot, orr, goff, grep)
        assert total == len(eids) and ng == (1 if len(eids) else 0) and sorted(orr[:total].tolist()) == sorted(eids.tolist())


def test_tickets_over_page_locked_buffers_in_place(eng):
    """bmq_match_submit_dev / bmq_match_wait_dev: the kernels read the inputs and write the CSR in place in page-locked host memory (what
    the batching front launches through); several in flight; too small an id buffer is NOSPACE with the size."""
    import ctypes as C

    from bifromq_amd.engine import _ptr
    L = B._lib.lib()
    w = B.Workload(0xF0A9, 9, 2000, 1)
    eng.rebuild(w.keys())
    tn = w.tenants()
    tdata, toff = w.tenants_packed()
    p_t, p_to = pinned(len(tdata) + 32, np.uint8), pinned(len(toff), np.uint32)
    p_t[:] = 0
    p_t[:len(tdata)], p_to[:] = tdata, toff
    runs = []
    for k, n in enumerate((1, 257, 5000)):
        data, off, tt = w.topics(40 + k, n)
        pd, po, pt = pinned(len(data) + 32, np.uint8), pinned(len(off), np.uint32), pinned(len(tt), np.uint32)
        pd[:] = 0
        pd[:len(data)], po[:], pt[:] = data, off, tt
        erow, eids = eng.match_batch(tn, tt, packed_topics=(data, off))
        row, ids, tot = pinned(n + 1, np.uint32), pinned(len(eids) + 8, np.uint32), pinned(1, np.uint64)
        row[:], ids[:] = 0, 0xFFFFFFFF
        t = C.c_int()
        assert L.bmq_match_submit_dev(eng.h, _ptr(p_t), _ptr(p_to), len(tn), _ptr(pt), _ptr(pd), _ptr(po), n, _ptr(row), _ptr(ids), len(ids), _ptr(tot),
                                      C.byref(t)) == 0
        runs.append((t.value, n, erow, eids, row, ids, tot, (pd, po, pt)))
    assert sorted(r[0] for r in runs) == [0, 1, 2]
    for t, n, erow, eids, row, ids, tot, _ in reversed(runs):
        need = C.c_uint64()
        assert L.bmq_match_wait_dev(eng.h, t, C.byref(need)) == 0
        assert need.value == len(eids) == int(tot[0]) and (row == erow).all() and (ids[:len(eids)] == eids).all()
    # short id buffer
    t, n, erow, eids, row, ids, tot, (pd, po, pt) = runs[1]
    row[:] = 0
    tk, need = C.c_int(), C.c_uint64()
    assert L.bmq_match_submit_dev(eng.h, _ptr(p_t), _ptr(p_to), len(tn), _ptr(pt), _ptr(pd), _ptr(po), n, _ptr(row), _ptr(ids), 3, _ptr(tot), C.byref(tk)) == 0
    with pytest.raises(B.BmqError):
        eng.match_wait(tk.value, row, ids)  # the host-buffer wait does not take a device-buffer ticket
    assert L.bmq_match_wait_dev(eng.h, tk.value, C.byref(need)) == -3 and need.value == len(eids) and (row == erow).all()
    assert L.bmq_match_wait_dev(eng.h, tk.value, C.byref(need)) == -7  # released: BMQ_E_STATE
