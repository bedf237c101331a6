"""SURVEY.md 8f-4, second half: fan-out grouping = the segmented sort of a match CSR by DelivererKey(subBrokerId, delivererKey)
(bmq_fanout_group / bmq_fanout_group_dev, bifromq_amd/csrc/bmq_fanout_core.h) against the oracle restatement of what
DeliverExecutorGroup.submit -> DeliverExecutor.send -> BatchDeliveryCall.add do with the matched routes of a batch
(oracle/oracle.py::fanout_groups).

The CPU tests run the per-pair functions on a host-only engine (the very code the gfx950 kernels wrap) over a CSR computed by the
oracle; the GPU tests take the CSR from bmq_match_batch / bmq_match_batch_dev on the device."""
import random

import numpy as np
import pytest

import bifromq_amd as B
from oracle import oracle as O
from tests import util as U

SHARED, DEAD = 0xFFFFFFFE, 0xFFFFFFFF


def _workload(seed, n_filters=400, n_topics=300, brokers=(0, 1, 2), dkeys=9, shared=0.12):
    rnd = random.Random(seed)
    tenants = ["tenantA", "t2"]
    keys = set()
    for i in range(n_filters):
        f = U.rand_filter(rnd, 4, ["a", "b", "c", "", "dev"])
        t = rnd.choice(tenants)
        p = rnd.random()
        if p < shared / 2:
            keys.add(O.route_key_from_mqtt(t, "$share/g%d/%s" % (rnd.randint(0, 3), f)))
        elif p < shared:
            keys.add(O.route_key_from_mqtt(t, "$oshare/g%d/%s" % (rnd.randint(0, 3), f)))
        else:
            # receiver ids and deliverer keys chosen so that some differ only in the part that does NOT belong to the DelivererKey
            keys.add(O.route_key_from_mqtt(t, f, O.receiver_url(rnd.choice(brokers), "inbox%d" % rnd.randint(0, 50), "d%d" % rnd.randint(0, dkeys - 1))))
    keys = sorted(keys)
    topics = [U.rand_topic(rnd, 4, ["a", "b", "c", "", "dev"]) for _ in range(n_topics)]
    tt = np.array([rnd.randrange(len(tenants)) for _ in topics], dtype=np.uint32)
    return tenants, keys, topics, tt


def _csr(rows):
    row = np.zeros(len(rows) + 1, dtype=np.uint32)
    row[1:] = np.cumsum([len(r) for r in rows])
    ids = np.array([x for r in rows for x in r], dtype=np.uint32)
    return row, ids


def _check(eng, rows, key_of, result):
    """result of Engine.fanout_group vs the oracle: the same groups, each with the same pairs in (topic, route) order"""
    ot, orr, goff, grep, special = result
    exp, exp_shared, exp_dead = O.fanout_groups(key_of, rows)
    n_groups = len(goff) - 1
    assert goff[0] == 0 and goff[-1] == sum(len(r) for r in rows) and (np.diff(goff.astype(np.int64)) > 0).all()
    got = {}
    kinds = []
    for g in range(n_groups):
        pairs = list(zip(ot[goff[g]:goff[g + 1]].tolist(), orr[goff[g]:goff[g + 1]].tolist()))
        assert pairs == sorted(pairs)  # (topic, route id) order inside a group
        rep = int(grep[g])
        kinds.append(rep)
        if rep == SHARED:
            assert pairs == exp_shared
        elif rep == DEAD:
            assert pairs == exp_dead
        else:
            assert (any(p[1] == rep for p in pairs))  # the group is named by one of its own routes
            dk = O.deliverer_key_of(eng.route_key(rep))
            assert dk not in got
            got[dk] = pairs
    assert got == exp
    assert special == (1 if exp_shared else 0) | (2 if exp_dead else 0)
    # the special groups come last: shared, then dead
    tail = [k for k in kinds if k >= SHARED]
    assert kinds[len(kinds) - len(tail):] == tail == sorted(tail)
    return len(exp)


def test_host_engine_groups_equal_oracle():
    tenants, keys, topics, tt = _workload(1)
    eng = B.Engine(device=-1).rebuild(keys)  # ids = ranks of the sorted keys
    kv = O.KV(keys)
    rows = U.semantic_rows(kv, tenants, tt, topics)
    assert sum(len(r) for r in rows) > 500
    n = _check(eng, rows, lambda i: keys[i], eng.fanout_group(*_csr(rows)))
    assert n == 27  # 3 brokers x 9 deliverer keys: every DelivererKey of the workload is hit
    # a second call reuses the per-route cache; a smaller group table reports how many it needs
    _check(eng, rows, lambda i: keys[i], eng.fanout_group(*_csr(rows), group_cap=2))
    # empty batch / empty rows
    ot, orr, goff, grep, sp = eng.fanout_group(np.zeros(4, dtype=np.uint32), np.zeros(0, dtype=np.uint32))
    assert len(ot) == 0 and goff.tolist() == [0] and sp == 0
    eng.close()


def test_host_engine_many_deliverers_force_table_growth():
    # 3 x 700 DelivererKeys > half of the initial 1024-slot table: the table grows (x4) and every route is mapped afresh
    tenants, keys, topics, tt = _workload(2, n_filters=6000, n_topics=1500, dkeys=700, shared=0.02)
    eng = B.Engine(device=-1).rebuild(keys)
    rows = U.semantic_rows(O.KV(keys), tenants, tt, topics)
    for _ in range(3):  # first call: fill + grow; second: grown table, cached; third: steady state
        n = _check(eng, rows, lambda i: keys[i], eng.fanout_group(*_csr(rows), group_cap=4096))
    assert n > 900
    eng.close()


def test_host_engine_after_churn_and_dead_ids():
    tenants, keys, topics, tt = _workload(3)
    eng = B.Engine(device=-1).rebuild(keys)
    rows = U.semantic_rows(O.KV(keys), tenants, tt, topics)
    base = eng.fanout_group(*_csr(rows))
    # delete a fifth of the routes: a CSR computed BEFORE the apply now carries dead ids -> the trailing dead group
    rnd = random.Random(5)
    gone = set(rnd.sample(range(len(keys)), len(keys) // 5))
    eng.apply([(1, keys[i]) for i in sorted(gone)])
    res = eng.fanout_group(*_csr(rows))
    _check(eng, rows, lambda i: None if i in gone else keys[i], res)
    assert res[4] & 2
    # add routes with new deliverer keys: ids beyond the rebuild's; CSR rows that use them
    new_keys = [O.route_key_from_mqtt("tenantA", "x/%d" % i, O.receiver_url(7, "inbox%d" % i, "fresh%d" % (i % 3))) for i in range(40)]
    eng.apply([(0, k) for k in new_keys])
    first = eng.info().next_route_id - len(new_keys)
    key_of = lambda i: (None if i in gone else keys[i]) if i < len(keys) else new_keys[i - first]
    rows2 = [r + ([first + (t % 40)] if t % 3 == 0 else []) for t, r in enumerate(rows)]
    n = _check(eng, rows2, key_of, eng.fanout_group(*_csr(rows2)))
    assert n >= 3
    # a rebuild renumbers the ids: the per-route cache must not survive it
    live = sorted(k for i, k in enumerate(keys) if i not in gone) + new_keys
    live = sorted(live)
    eng.rebuild(live)
    rows3 = U.semantic_rows(O.KV(live), tenants, tt, topics)
    _check(eng, rows3, lambda i: live[i], eng.fanout_group(*_csr(rows3)))
    assert base[4] in (0, 1)
    eng.close()


def test_reference_delivery_cases():
    """The delivery side of the reference's own end-to-end tests as known answers for the grouping (oracle restatement AND engine):
    DWT/DistQoS0Test.java:95-150 (testDistCase2: three routes, two deliverers -- writer1 = (MqttBroker, "batch1") receives 2 MatchInfos,
    writer2 = (InboxService, "batch2") receives 1) and :152-193 (testDistCase3: two inboxes behind ONE deliverer key -> one delivery
    request carrying both MatchInfos for the one topic)."""
    MQTT_BROKER, INBOX_SERVICE = 0, 1  # DWT/DistWorkerTest.java:130-131
    tenant = "tenantA"
    keys = sorted([O.route_key_from_mqtt(tenant, "/你好/hello/😄", O.receiver_url(MQTT_BROKER, "inbox1", "batch1")),
                   O.route_key_from_mqtt(tenant, "/#", O.receiver_url(MQTT_BROKER, "inbox1", "batch1")),
                   O.route_key_from_mqtt(tenant, "/#", O.receiver_url(INBOX_SERVICE, "inbox2", "batch2")),
                   O.route_key_from_mqtt(tenant, "/a/b/c", O.receiver_url(MQTT_BROKER, "inbox1", "batch1")),
                   O.route_key_from_mqtt(tenant, "/a/b/c", O.receiver_url(MQTT_BROKER, "inbox2", "batch1"))])
    eng = B.Engine(device=-1).rebuild(keys)
    kv = O.KV(keys)
    # case 2: fan-out 3 (DistQoS0Test.java:103-104), 2 + 1 by deliverer
    rows = U.semantic_rows(kv, [tenant], np.zeros(1, dtype=np.uint32), ["/你好/hello/😄"])
    assert len(rows[0]) == 3
    groups, shared, dead = O.fanout_groups(lambda i: keys[i], rows)
    assert {k: len(v) for k, v in groups.items()} == {(MQTT_BROKER, "batch1"): 2, (INBOX_SERVICE, "batch2"): 1} and not shared and not dead
    assert _check(eng, rows, lambda i: keys[i], eng.fanout_group(*_csr(rows))) == 2
    # case 3: both inboxes of /a/b/c (and the MqttBroker '/#' route) sit behind (MqttBroker, "batch1"): ONE group carries them for the topic
    rows = U.semantic_rows(kv, [tenant], np.zeros(1, dtype=np.uint32), ["/a/b/c"])
    groups, shared, dead = O.fanout_groups(lambda i: keys[i], rows)
    inboxes = {O.parse_route_key(keys[r])[3].split("\0")[1] for (_, r) in groups[(MQTT_BROKER, "batch1")] if O.parse_route_key(keys[r])[2] == "/a/b/c"}
    assert inboxes == {"inbox1", "inbox2"}
    assert _check(eng, rows, lambda i: keys[i], eng.fanout_group(*_csr(rows))) == 2  # (MqttBroker, batch1) and the InboxService '/#' route
    eng.close()


def test_argument_checks():
    eng = B.Engine(device=-1)
    with pytest.raises(B.BmqError):  # no index yet
        eng.fanout_group(np.array([0, 1], dtype=np.uint32), np.array([0], dtype=np.uint32))
    eng.rebuild([O.route_key_from_mqtt("t", "a", O.receiver_url(0, "i", "d"))])
    with pytest.raises(B.BmqError):  # row_ptr must ascend
        eng.fanout_group(np.array([0, 2, 1], dtype=np.uint32), np.array([0], dtype=np.uint32))
    ot, orr, goff, grep, sp = eng.fanout_group(np.array([0, 1, 3], dtype=np.uint32), np.array([0, 0, 9], dtype=np.uint32))
    assert sp == 2 and grep.tolist() == [0, DEAD] and ot.tolist() == [0, 1, 1] and orr.tolist() == [0, 0, 9]  # id 9 was never handed out
    eng.close()


# ---- on the device --------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_groups_equal_oracle_and_host_engine():
    tenants, keys, topics, tt = _workload(11, n_filters=3000, n_topics=4000, dkeys=40)
    eng = B.Engine(device=0).rebuild(keys)
    row, ids = eng.match_batch(tenants, tt, topics)
    rows = U.csr_rows(row, ids)
    assert rows == U.semantic_rows(O.KV(keys), tenants, tt, topics)
    res = eng.fanout_group(row, ids)
    n = _check(eng, rows, lambda i: keys[i], res)
    assert n == 120
    _check(eng, rows, lambda i: keys[i], eng.fanout_group(row, ids, group_cap=7))  # NOSPACE -> retry with the reported size
    # the same batch on a host-only engine gives the same groups (as sets of pairs; group order is unspecified)
    h = B.Engine(device=-1).rebuild(keys)
    hres = h.fanout_group(row, ids)

    def as_set(r):
        return {tuple(zip(r[0][r[2][g]:r[2][g + 1]].tolist(), r[1][r[2][g]:r[2][g + 1]].tolist())) for g in range(len(r[2]) - 1)}
    assert as_set(res) == as_set(hres)
    # churn on the device index, dead ids in an old CSR
    gone = set(range(0, len(keys), 4))
    eng.apply([(1, keys[i]) for i in sorted(gone)])
    _check(eng, rows, lambda i: None if i in gone else keys[i], eng.fanout_group(row, ids))
    h.close()
    eng.close()


@pytest.mark.gpu
def test_gpu_device_resident_csr_and_table_growth():
    """bmq_fanout_group_dev on the CSR bmq_match_batch_dev left in HBM; 2100 DelivererKeys force the group table to grow twice."""
    import ctypes as C
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

    def from_dev(p, n):
        out = np.zeros(n, dtype=np.uint32)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes, 2) == 0
        return out

    tenants, keys, topics, tt = _workload(12, n_filters=20000, n_topics=5000, dkeys=700, shared=0.02)
    eng = B.Engine(device=0).rebuild(keys)
    n = len(topics)
    tdata, toff = B.pack(tenants)
    pdata, poff = B.pack(topics)
    d = [to_dev(np.ascontiguousarray(x)) for x in (tdata, toff, tt, pdata, poff)]
    cap = 16000000
    row, ids, tot = to_dev(np.zeros(n + 1, dtype=np.uint32)), to_dev(np.zeros(cap, dtype=np.uint32)), to_dev(np.zeros(1, dtype=np.uint64))
    eng.match_batch_device(d[0], d[1], len(tenants), d[2], d[3], d[4], n, row, ids, cap, tot)
    total = eng.finish()
    rows = U.csr_rows(from_dev(row, n + 1), from_dev(ids, total))
    gcap = 4096
    ot, orr, goff, grep = (to_dev(np.zeros(total, dtype=np.uint32)), to_dev(np.zeros(total, dtype=np.uint32)), to_dev(np.zeros(gcap + 1, dtype=np.uint32)),
                           to_dev(np.zeros(gcap, dtype=np.uint32)))
    for _ in range(3):
        ng, sp = eng.fanout_group_device(row, ids, n, total, ot, orr, goff, grep, gcap)
        res = (from_dev(ot, total), from_dev(orr, total), from_dev(goff, ng + 1), from_dev(grep, ng), sp)
        groups = _check(eng, rows, lambda i: keys[i], res)
    assert groups > 1500
    with pytest.raises(B.BmqError) as ex:
        eng.fanout_group_device(row, ids, n, total, ot, orr, goff, grep, 10)
    assert ex.value.code == -3 and ex.value.needed == ng
    for p in bufs:
        hip.hipFree(p)
    eng.close()
