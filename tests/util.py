"""Shared helpers for the parity tests: random topic/filter generators and oracle adapters."""
import random

import numpy as np

from oracle import oracle as O

ALPHABET = ["a", "b", "c", "", "$sys", "$x", "你好", "😄", "dev", "+x", "a b", "#a", "x" * 17, "y" * 40]


def rand_level(rnd, alphabet=ALPHABET):
    return rnd.choice(alphabet)


def rand_topic(rnd, max_levels=5, alphabet=ALPHABET):
    return "/".join(rand_level(rnd, alphabet) for _ in range(rnd.randint(1, max_levels)))


def rand_filter(rnd, max_levels=5, alphabet=ALPHABET):
    n = rnd.randint(1, max_levels)
    lv = []
    for i in range(n):
        p = rnd.random()
        if p < 0.2:
            lv.append("+")
        elif p < 0.3 and i == n - 1:
            lv.append("#")
        else:
            lv.append(rand_level(rnd, alphabet))
    return "/".join(lv)


def rand_route_key(rnd, tenant, topic_filter, rid):
    p = rnd.random()
    if p < 0.1:
        return O.route_key_from_mqtt(tenant, "$share/g%d/%s" % (rnd.randint(0, 3), topic_filter))
    if p < 0.15:
        return O.route_key_from_mqtt(tenant, "$oshare/g%d/%s" % (rnd.randint(0, 3), topic_filter))
    broker = rnd.choice([0, 1, 1, 2])
    return O.route_key_from_mqtt(tenant, topic_filter, O.receiver_url(broker, "inbox%d" % rid, "d%d" % (rid % 7)))


def oracle_rows(kv, tenants, topic_tenant, topics):
    """Per-topic sorted route ranks from the structural oracle, one matchAll per tenant (whole-batch mode)."""
    rows = [None] * len(topics)
    by_tenant = {}
    for i, ti in enumerate(topic_tenant):
        by_tenant.setdefault(int(ti), []).append(i)
    for ti, idxs in by_tenant.items():
        res = kv.match_all(tenants[ti], [topics[i] for i in idxs]).per_topic()
        for j, i in enumerate(idxs):
            rows[i] = sorted(res[j])
    return rows


def semantic_rows(kv, tenants, topic_tenant, topics):
    """Authoritative rows: the reference's own brute-force TopicMatcher (TRIET/TopicMatcher.java:39-101, restated
    in bmq_oracle.cpp) applied to every key of the tenant.  O(topics x keys of tenant)."""
    rows = [None] * len(topics)
    by_tenant = {}
    for i, ti in enumerate(topic_tenant):
        by_tenant.setdefault(int(ti), []).append(i)
    for ti, idxs in by_tenant.items():
        res = kv.match_bruteforce(tenants[ti], [topics[i] for i in idxs]).per_topic()
        for j, i in enumerate(idxs):
            rows[i] = res[j]
    return rows


def csr_rows(row_ptr, ids):
    return [ids[row_ptr[i]:row_ptr[i + 1]].tolist() for i in range(len(row_ptr) - 1)]


def quirk_ii_filters(keys):
    """(tenant, filter F) pairs for which the KV also holds a filter F + "/" + "" + ... (next level empty).
    The reference's probe-then-seek loop can skip F's keys in that situation (SURVEY.md 8c quirk ii,
    DW/cache/TenantRouteMatcher.java:127-137): keys of "F/" sort between the seek target of F and F's own keys
    with a non-zero bucket byte.  The semantic oracle is authoritative there."""
    filters = set()
    for k in keys:
        flag, tenant, mqtt, _ = O.parse_route_key(k)
        if flag != 1:
            mqtt = mqtt.split("/", 2)[2]
        filters.add((tenant, mqtt))
    out = set()
    for tenant, f in filters:
        parts = f.split("/")
        for i in range(1, len(parts)):
            if parts[i] == "":
                out.add((tenant, "/".join(parts[:i])))
    return out


def assert_rows_equal_modulo_quirk_ii(keys, tenants, topic_tenant, reference_rows, engine_rows):
    """reference_rows come from the structural restatement in the reference's production call pattern (one
    matchAll per topic); they may only differ from the engine by routes LOST to quirk (ii)."""
    quirk = None
    n_diff = 0
    for i, (ref, got) in enumerate(zip(reference_rows, engine_rows)):
        if ref == got:
            continue
        n_diff += 1
        if quirk is None:
            quirk = quirk_ii_filters(keys)
        assert set(ref) <= set(got), i
        for r in set(got) - set(ref):
            flag, tenant, mqtt, _ = O.parse_route_key(keys[r])
            if flag != 1:
                mqtt = mqtt.split("/", 2)[2]
            assert (tenant, mqtt) in quirk, (i, mqtt)
    return n_diff


def churn_case(eng, match_fn, n_tenants, per_tenant, n_ops, n_topics, sample_tenants=16, n_sample=2000, seed=0xB1F20005):
    """configs[4]: an index of n_tenants x per_tenant generated routes, then ONE batch of n_ops mutations (50 % unsubscribes of
    existing routes, 50 % subscribes of new filters, spread over all tenants) through bmq_routes_apply, then a batch of
    publishes.  Checked: route count, ids are ranks again (route_key of sampled ids), CSR well-formed and ascending, tenant
    isolation with the NEW id ranges, sampled rows of the first tenants bit-exact vs the oracle on the updated key set.
    match_fn(tenants, topic_tenant, (data, off)) -> (row_ptr, ids): the engine's batch match (GPU), or a stand-in in the CPU test
    of this helper."""
    import random

    import numpy as np

    import bifromq_amd as B
    from oracle import oracle as O

    w = B.Workload(seed, n_tenants, per_tenant, 1)
    kb, ko = w.keys_packed()
    eng.rebuild(packed=(kb, ko))
    assert eng.info().n_routes == w.n_keys
    tn = w.tenants()
    first = np.asarray(w.tenant_first(), dtype=np.int64)
    rnd = random.Random(seed & 0xFFFF)
    mv = memoryview(kb)  # no copy of the (large) key bytes

    def key_at(i):
        return bytes(mv[int(ko[i]):int(ko[i + 1])])

    del_ids = sorted(set(rnd.randrange(w.n_keys) for _ in range(n_ops // 2)))
    ops = [(1, key_at(i)) for i in del_ids]
    per_tenant_delta = np.zeros(n_tenants, dtype=np.int64)
    owner = np.searchsorted(first, np.asarray(del_ids), side="right") - 1
    np.subtract.at(per_tenant_delta, owner, 1)
    added = {}
    for q in range(n_ops - n_ops // 2):
        t = rnd.randrange(n_tenants)
        # two in three new filters match nothing published ("churn/..."), the third is "<first level>/#" and matches a lot
        f = "churn/l1_%d/+/l3_%d" % (q % 64, q % 4096) if q % 3 else "l0_%d/#" % (q % 8)
        k = B.route_key(tn[t], f, 1, "0\0c%d\0d%d" % (q, q % 64))
        if k not in added:
            added[k] = t
            per_tenant_delta[t] += 1
        ops.append((0, k))
    eng.apply(ops)
    n_new = w.n_keys - len(del_ids) + len(added)
    assert eng.info().n_routes == n_new
    counts = np.diff(first) + per_tenant_delta
    new_first = np.concatenate([[0], np.cumsum(counts)])
    assert new_first[-1] == n_new
    # the key set of the first S tenants after the batch = exactly the ids [0, new_first[S])
    S = min(sample_tenants, n_tenants)
    hi_old = int(first[S])
    deleted = set(del_ids)
    keys_s = [key_at(i) for i in range(hi_old) if i not in deleted] + [k for k, t in added.items() if t < S]
    kv = O.KV(keys_s)  # sorts
    keys_sorted = sorted(keys_s)
    assert len(keys_sorted) == new_first[S]
    for i in sorted(rnd.sample(range(len(keys_sorted)), min(200, len(keys_sorted)))):
        assert eng.route_key(i) == keys_sorted[i]
    data, off, tt = w.topics(seed + 1000, n_topics)
    row, ids = match_fn(tn, tt, (data, off))
    assert row[0] == 0 and row[-1] == len(ids) and (np.diff(row.astype(np.int64)) >= 0).all()
    if len(ids):
        d = np.diff(ids.astype(np.int64))
        starts = row[1:-1][row[1:-1] < len(ids)]
        d[(starts - 1)[starts > 0]] = 1
        assert (d > 0).all()
        own = np.repeat(tt.astype(np.int64), np.diff(row.astype(np.int64)))
        assert ((ids >= new_first[own]) & (ids < new_first[own + 1])).all()
    cand = np.nonzero(tt < S)[0]
    sample = sorted(rnd.sample(cand.tolist(), min(n_sample, len(cand))))
    traw = data.tobytes()
    topics = [traw[off[i]:off[i + 1]] for i in sample]
    stt = tt[sample]
    res, _ = kv.match_singletons(tn[:S], stt, O.pack(topics), threads=8)
    got = [ids[row[i]:row[i + 1]].tolist() for i in sample]
    assert_rows_equal_modulo_quirk_ii(keys_sorted, tn[:S], stt.tolist(), [sorted(r) for r in res.per_topic()], got)
    assert any(keys_sorted[r] in added for g in got for r in g)  # routes subscribed by the batch are matched
    return n_new
