"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, the product codec agrees
with the oracle's, the index builder (run on host threads by a host-only engine) indexes every route exactly once, and matching
refuses to run without a device."""
import ctypes as C
import os
import random
import re

import numpy as np
import pytest

import bifromq_amd as B
from bifromq_amd import _lib
from oracle import oracle as O
from tests import util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bmq.h")).read()
    declared = set(re.findall(r"\b(bmq_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"bmq_engine", "bmq_config", "bmq_stats", "bmq_index_info", "bmq_status"}
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    L = C.CDLL(_lib.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in _lib.lib().bmq_version()


def test_no_cpu_fallback():
    e = B.Engine(device=-1)
    e.rebuild([B.route_key("t", "a/b", 1, "0\0r\0d")])
    with pytest.raises(B.BmqError) as ei:
        e.match_tenant("t", ["a/b"])
    assert ei.value.code == -2  # BMQ_E_NODEVICE


def test_codec_matches_oracle():
    rnd = random.Random(7)
    for i in range(500):
        tenant = rnd.choice(["t", "tenantA", "租户", ""])
        f = U.rand_filter(rnd)
        flag = rnd.choice([1, 2, 3])
        recv = O.receiver_url(rnd.randint(0, 2), "inbox%d/😄" % i, "d%d" % i) if flag == 1 else "grp%d" % i
        k = B.route_key(tenant, f, flag, recv)
        assert k == O.route_key(tenant, f, flag, recv)
        got = B.decode_route_key(k)
        exp = O.parse_route_key(k)
        assert got == exp
    for s in ["", "a", "hello", "你好", "😄x", "0\0inbox\0d"]:
        assert B.java_string_hash(s) == O.java_hash(s)
    assert B.decode_route_key(b"\x01garbage") is None
    assert B.decode_route_key(b"") is None


def test_schema_vectors():
    # SCHEMA KVSchemaUtilTest.java:88-145 round trips via the product codec
    k = B.route_key_from_mqtt("tenantA", "/a/b/c", O.receiver_url(1, "inbox1", "deliverer1"))
    assert B.decode_route_key(k) == (1, "tenantA", "/a/b/c", "1\0inbox1\0deliverer1")
    g = B.route_key_from_mqtt("tenantA", "$share/group//a/b/c")
    assert B.decode_route_key(g) == (2, "tenantA", "$share/group//a/b/c", "group")
    og = B.route_key_from_mqtt("tenantA", "$oshare/group//a/b/c")
    assert B.decode_route_key(og) == (3, "tenantA", "$oshare/group//a/b/c", "group")


def test_builder_indexes_every_route_once():
    rnd = random.Random(11)
    tenants = ["tA", "tB", "t"]
    keys = set()
    for i in range(3000):
        keys.add(U.rand_route_key(rnd, rnd.choice(tenants), U.rand_filter(rnd, 6), i))
    keys = sorted(keys)
    e = B.Engine(device=-1).rebuild(list(reversed(keys)))  # any input order
    info = e.info()
    assert info.n_routes == len(keys) and info.n_tenants == 3
    assert [e.route_key(i) for i in range(0, len(keys), 97)] == keys[::97]
    by_filter = {}
    for rank, k in enumerate(keys):
        flag, tenant, mqtt, recv = B.decode_route_key(k)
        if flag != 1:
            mqtt = mqtt.split("/", 2)[2]
        by_filter.setdefault((tenant, mqtt), []).append(rank)
    seen = 0
    for (tenant, f), ranks in by_filter.items():
        assert e.find(tenant, f) == ranks, (tenant, f)
        seen += len(ranks)
    assert seen == len(keys)
    assert e.find("tA", "no/such/filter") == [] and e.find("nobody", "a") == []


def test_apply_put_delete_host():
    ks = [B.route_key("t", "a/%d" % i, 1, "0\0r%d\0d" % i) for i in range(50)]
    e = B.Engine(device=-1).rebuild(ks)
    e.apply([(1, ks[3]), (0, B.route_key("t", "z", 1, "0\0r\0d")), (1, ks[7]), (0, ks[7]), (1, b"\x00\x00\x01tq\x00\x00\x00\x01\x00\x00")])
    z = B.route_key("t", "z", 1, "0\0r\0d")
    exp = (set(ks) - {ks[3]}) | {z}
    info = e.info()
    assert info.n_routes == len(exp) and info.epoch == 2 and info.generation == 1 and info.next_route_id == 52
    # ids are stable handles: survivors keep their rank, the deleted ids are dead, the two adding puts got 50 and 51
    ranks = sorted(ks)
    got = e.route_keys(list(range(52)))
    assert got[:50] == [b"" if k in (ks[3], ks[7]) else k for k in ranks] and got[50:] == [z, ks[7]]
    assert e.find("t", "z") == [50] and e.find("t", "a/7") == [51] and e.find("t", "a/3") == []
    with pytest.raises(B.BmqError) as ei:
        e.route_key(ranks.index(ks[3]))
    assert ei.value.code == -1
    with pytest.raises(B.BmqError):
        e.apply([(0, b"not a key")])
    with pytest.raises(B.BmqError):  # a bad op in the middle of a batch: nothing of the batch is applied
        e.apply([(0, B.route_key("t", "new", 1, "0\0r\0d")), (0, b"\x01garbage")])
    assert e.info().n_routes == len(exp) and e.find("t", "new") == []
    e.rebuild(ks)  # a rebuild re-numbers: ranks again, new generation
    assert e.info().generation == 2 and e.route_keys(list(range(50))) == ranks


def test_workload_generator_is_deterministic_and_sorted():
    w1, w2 = B.Workload(0xB1F20002, 3, 500), B.Workload(0xB1F20002, 3, 500)
    k1 = w1.keys()
    assert k1 == w2.keys() and k1 == sorted(set(k1))
    assert all(B.decode_route_key(k) is not None for k in k1)
    d1, o1, t1 = w1.topics(9, 200)
    d2, o2, t2 = w2.topics(9, 200)
    assert d1.tobytes() == d2.tobytes() and (o1 == o2).all() and (t1 == t2).all()
    # 90 % of publishes are instantiated from a stored filter: most must have >= 1 semantic match
    kv = O.KV(k1)
    from bifromq_amd.workload import unpack
    topics = [t.decode() for t in unpack(d1, o1)]
    tn = w1.tenants()
    hits = sum(1 for t, ti in zip(topics, t1) if kv.match_bruteforce(tn[ti], [t]).per_topic()[0])
    assert hits >= 0.8 * len(topics)


def test_hash_sharded_workloads_partition_the_population():
    """bench.py --gpus N gives every rank the tenants hash(tenantId) mod N assigns to it: the shards' key sets are a
    partition of the unsharded workload's key set (a tenant's routes depend only on seed and global tenant index)."""
    from bifromq_amd import shard
    full = set(B.Workload(5, 16, 40, 1).keys())
    union, total = set(), 0
    for r in range(4):
        mine = [t for t in range(16) if shard.tenant_rank("tenant%06d" % t, 4) == r]
        w = B.Workload(5, len(mine), 40, 1, tenant_ids=mine)
        assert w.tenants() == ["tenant%06d" % t for t in mine]
        ks = w.keys()
        total += len(ks)
        union |= set(ks)
        _, _, tt = w.topics(3, 50)
        assert len(mine) == 0 or tt.max() < len(mine)
    assert union == full and total == len(full)


def test_host_index_fuzz_under_sanitizers():
    """tools/host_fuzz.cpp: the index BUILDER -- the very functions the gfx950 builder kernels run (bmq_build_core.h), driven by
    the same host control (bmq_dist_index.h) -- on host threads under ASan + UBSan and under TSan: random rebuild / apply
    sequences with minimal initial capacities (every growth path runs all the time); after every step ids must follow the ABI's
    rule (ranks after a rebuild, next unused id for an adding put, stable otherwise) and a CPU walk over the image (directory,
    regions, dictionary, exactly the probes k_walk does) must give the brute-force result of the matching rule.  Round 5: every fourth
    round also builds the NEXT GENERATION beside the index (reserve_like, import_snapshot / import_apply in chunks of random size, the
    index mutated between a chunk's snapshot and its apply, the log replayed as one merged batch): its key set must be the model's."""
    import subprocess
    csrc = os.path.join(ROOT, "bifromq_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "fuzz"], check=True, capture_output=True, timeout=600)
    exe = os.path.join(ROOT, "tools", "host_fuzz")
    for seed, rounds, threads in ((1, 40, 4), (9, 58, 1), (3, 40, 8)):
        r = subprocess.run([exe, str(seed), str(rounds), "3000", str(threads)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "host_fuzz ok" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "5", "20", "20000", "8"], capture_output=True, text=True, timeout=600, env=dict(os.environ, BMQ_FUZZ_BIG="1"))
    assert r.returncode == 0 and "host_fuzz ok" in r.stdout, r.stdout + r.stderr
    for seed in ("2", "11"):  # the lock-free inserts (dictionary slots, trie slots, Bloom words) under ThreadSanitizer
        r = subprocess.run([exe + "_tsan", seed, "25", "3000", "8"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "host_fuzz ok" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr
    # the retain direction (tools/retain_fuzz.cpp): the bulk load, and the mutation functions the gfx950 kernels k_r_locate / k_r_commit /
    # k_r_rank run (bmq_retain_core.h) on host threads with minimal capacities: id stability, stamps, dead-aware image walk + overlay
    # walk == brute force, GC scan, live-id listing; the lock-free overlay inserts under ThreadSanitizer
    exe = os.path.join(ROOT, "tools", "retain_fuzz")
    for seed, rounds, threads in ((1, 30, 4), (7, 30, 1)):
        r = subprocess.run([exe, str(seed), str(rounds), str(threads)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "retain_fuzz ok" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe + "_tsan", "3", "25", "8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "retain_fuzz ok" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr


def test_header_is_plain_c_and_usable_from_c(tmp_path):
    """include/bmq.h is what a JNI/cgo binding compiles against: it must be valid C99 (-pedantic) and the host-side entry
    points must work from a C program linked against libbmq.so (tests/c/abi_smoke.c; no GPU: host-only engine)."""
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "bifromq_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L", libdir, "-lbmq", "-Wl,-rpath," + libdir],
                   check=True, capture_output=True, timeout=120)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "abi_smoke ok" in r.stdout, r.stdout + r.stderr


def test_batching_front_under_thread_sanitizer():
    """tools/batcher_tsan.cpp: the batching front (bmq_batcher.inc) built with -fsanitize=thread against a stand-in engine whose
    match is a pure function of (tenant, topic): 24 blocking callers, 4 submitting threads, an epoch-bumping mutator, three
    max_batch settings, destroy while requests wait -- every caller must get exactly its own rows and TSan must stay silent."""
    import subprocess
    csrc = os.path.join(ROOT, "bifromq_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "tsan"], check=True, capture_output=True, timeout=600)
    exe = os.path.join(ROOT, "tools", "batcher_tsan")
    for _ in range(3):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "batcher_tsan ok" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr


def test_jni_binding_compiles_against_the_header(tmp_path):
    """integration/jni/bmq_jni.c -- the binding INTEGRATION.md describes -- compiles (C99, -Werror) against include/bmq.h and links
    against libbmq.so, and exports one Java_..._NativeMatcher_<name> symbol per native method NativeMatcher.java declares.
    (No JDK in this image: jni_min.h stands in for jni.h; the Java sources are not compiled.)"""
    import subprocess
    so = str(tmp_path / "libbmq_jni.so")
    libdir = os.path.join(ROOT, "bifromq_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "integration", "jni"), "-o", so, os.path.join(ROOT, "integration", "jni", "bmq_jni.c"), "-L", libdir,
                    "-lbmq"], check=True, capture_output=True, timeout=120)
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"Java_org_apache_bifromq_dist_worker_gpu_NativeMatcher_(\w+)", syms))
    java = open(os.path.join(ROOT, "integration", "java", "org", "apache", "bifromq", "dist", "worker", "gpu", "NativeMatcher.java")).read()
    declared = set(re.findall(r"static native \w+ (\w+)\(", java))
    assert declared and exported == declared, (sorted(declared - exported), sorted(exported - declared))


def test_churn_case_helper_on_host_only_engine():
    """tests/util.py::churn_case is the body of the full-size configs[4] GPU test; here it runs small, on a host-only engine
    (rebuild / apply / info / route_key are host code) with the match taken from the semantic oracle over the updated key set,
    so that the helper's bookkeeping (new id ranges, sub-KV of the first tenants, sampling) is verified without a GPU."""
    eng = B.Engine(device=-1)
    state = {}

    def oracle_match(tn, tt, packed):
        data, off = packed
        raw = data.tobytes()
        topics = [raw[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]
        all_ids = list(range(int(eng.info().next_route_id)))
        live = [(k, i) for i, k in zip(all_ids, eng.route_keys(all_ids)) if k]  # (key, engine id) of every live route
        live.sort()
        rows = U.semantic_rows(O.KV([k for k, _ in live]), tn, tt, topics)  # ranks in the live key list ...
        rows = [sorted(live[r][1] for r in rr) for rr in rows]              # ... as engine ids, ascending
        row = np.zeros(len(rows) + 1, dtype=np.uint32)
        row[1:] = np.cumsum([len(r) for r in rows])
        state["rows"] = len(rows)
        return row, np.array([x for r in rows for x in r], dtype=np.uint32)

    n = U.churn_case(eng, oracle_match, n_tenants=12, per_tenant=400, n_ops=1200, n_topics=1500, sample_tenants=5, n_sample=300)
    assert state["rows"] == 1500 and n == eng.info().n_routes
    # ... and the whole-CSR mode the full-size GPU test uses (every row of the first tenants, ids mapped to ranks vectorised)
    n = U.churn_case(eng, oracle_match, n_tenants=12, per_tenant=400, n_ops=1200, n_topics=1500, sample_tenants=9, n_sample=None)
    assert n == eng.info().n_routes
    eng.close()


def test_retain_key_schema_vectors_and_parity():
    """the retain store's key schema.  Golden: KVSchemaUtilTest.java:45-79 (retainKeyPrefix of 13 filters, written
    there in terms of tenantBeginKey / level count / LevelHash) and LevelHashTest.java:30-40 (sizes); parity: the product codec
    against the oracle's independent restatement on random topics / filters incl. non-BMP characters."""
    from bifromq_amd.engine import retain_filter_route, retain_message_key
    ns = O.retain_tenant_begin_key("tenantA")
    lb = lambda v: v.to_bytes(2, "big")
    H = O.retain_level_hash
    table = {"#": ns + lb(0), "/#": ns + lb(1) + H([""]), "+": ns + lb(1), "+/#": ns + lb(1), "a/#": ns + lb(1) + H(["a"]),
             "/a": ns + lb(2) + H(["", "a"]), "a/+": ns + lb(2) + H(["a"]), "a/b": ns + lb(2) + H(["a", "b"]), "/a/#": ns + lb(2) + H(["", "a"]),
             "/a/+": ns + lb(3) + H(["", "a"]), "/a/+/+": ns + lb(4) + H(["", "a"]), "/+/b/": ns + lb(4) + H([""]), "/+/b/+/": ns + lb(5) + H([""])}
    for f, exp in table.items():
        assert O.retain_key_prefix_of_filter("tenantA", f) == exp, f  # the oracle reproduces the reference's own table
        key, lh, levels, wild, multi = retain_filter_route("tenantA", f)
        if wild:
            assert key == exp and lh == H(O.retain_filter_prefix(f.split("/"))), f
        else:
            assert key == O.retain_message_key("tenantA", f) and key.startswith(exp), f
        assert multi == f.endswith("#") and levels == len(f.split("/")) - (1 if multi else 0)
    assert len(H([])) == 0 and len(H([""])) == 1 and len(H(["a", "b", "c"])) == 3  # LevelHashTest
    rnd = random.Random(5)
    alpha = ["a", "b", "", "$sys", "你好", "😄", "x" * 30, "0", " "]
    for _ in range(500):
        tenant = rnd.choice(["t", "tenantA", "租户"])
        topic = "/".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 6)))
        assert retain_message_key(tenant, topic) == O.retain_message_key(tenant, topic)
        f = U.rand_filter(rnd, 6, alpha)
        key, lh, levels, wild, multi = retain_filter_route(tenant, f)
        if wild:
            assert key == O.retain_key_prefix_of_filter(tenant, f) and lh == H(O.retain_filter_prefix(f.split("/")))
        else:
            assert key == O.retain_message_key(tenant, f)


def test_compact_renumbers_and_frees_garbage_host():
    """bmq_compact on a host-only engine: after churn (deleted routes, a tenant that outgrew its region twice) the index is
    re-built from its own live keys: ids are ranks again, a new generation, no garbage, exact lookups unchanged."""
    ks = [B.route_key("t", "a/%d" % i, 1, "0\0r%d\0d" % i) for i in range(200)]
    e = B.Engine(device=-1).rebuild(ks)
    e.apply([(1, k) for k in ks[::3]] + [(0, B.route_key("t", "grow/%d/+/x" % i, 1, "0\0g%d\0d" % i)) for i in range(3000)])
    before = e.info()
    assert before.garbage_bytes > 0 and before.next_route_id == 3200
    live = sorted(k for k in e.route_keys(list(range(3200))) if k)
    assert len(live) == before.n_routes == 200 - 67 + 3000
    e.compact()
    after = e.info()
    assert after.generation == before.generation + 1 and after.n_routes == len(live) and after.next_route_id == len(live)
    assert after.garbage_bytes == 0 and after.n_nodes < before.n_nodes  # the nodes of the 67 deleted filters are gone
    assert e.route_keys(list(range(len(live)))) == live
    assert e.find("t", "grow/7/+/x") == [live.index(B.route_key("t", "grow/7/+/x", 1, "0\0g7\0d"))] and e.find("t", "a/0") == []


def test_route_cache_under_sanitizers():
    """tools/cache_fuzz.cpp: the route cache (bmq_cache.cpp: ISubscriptionCache / TenantRouteCache / TopicIndex on the engine's side of
    the boundary) over a stand-in engine, under ASan + UBSan and under TSan: TopicIndex against the reference's golden table
    (DWT/TopicIndexTest.java:41-73) and the matching rule, hit / invalidate / evict / expire / rebuild behaviour, and getter threads
    against a mutator -- no load overtaken by a mutation may end up cached."""
    import subprocess
    csrc = os.path.join(ROOT, "bifromq_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "cachefuzz"], check=True, capture_output=True, timeout=600)
    exe = os.path.join(ROOT, "tools", "cache_fuzz")
    r = subprocess.run([exe, "1", "6", "1200"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cache_fuzz ok" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe + "_tsan", "2", "6", "1200"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cache_fuzz ok" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr


def test_route_detail_and_receiver_cache_cases():
    """SCHEMA/cache/RouteDetailCacheTest.java:46-131 (a route key parses back to tenant, receiverUrl / group, type and the ORIGINAL MQTT topic
    filter, for normal, $share and $oshare routes; an unknown flag byte is refused) and ReceiverCacheTest.java:38-52 (receiverUrl ->
    subBrokerId, receiverId, delivererKey) -- for the product codec (bmq_route_key_decode) and the oracle's parser, on the UUID-shaped
    inputs the reference tests use."""
    import uuid
    for _ in range(20):
        tenant = "tenant-%s" % uuid.uuid4()
        broker = uuid.uuid4().int & 0x7FFFFFFF
        rid, dk = "inbox-%s" % uuid.uuid4(), "deliverer-%s" % uuid.uuid4()
        url = O.receiver_url(broker, rid, dk)
        # normal route
        f = "/home/%s" % uuid.uuid4()
        k = O.route_key_from_mqtt(tenant, f, url)
        assert k == B.route_key_from_mqtt(tenant, f, url)
        assert O.parse_route_key(k) == (1, tenant, f, url) == B.decode_route_key(k)
        assert O.deliverer_key_of(k) == (broker, dk)  # ReceiverCache.get(url): parts[0] as int, parts[2]
        # unordered / ordered share: the group is the receiver part, the MQTT filter comes back with its prefix
        for prefix, flag in (("$share", 2), ("$oshare", 3)):
            orig = "%s/group-%s/s/%s" % (prefix, uuid.uuid4(), uuid.uuid4())
            gk = O.route_key_from_mqtt(tenant, orig)
            assert gk == B.route_key_from_mqtt(tenant, orig)
            flag_got, tn, mqtt, group = O.parse_route_key(gk)
            assert (flag_got, tn, mqtt) == (flag, tenant, orig) and group == orig.split("/")[1]
            assert B.decode_route_key(gk) == (flag, tenant, orig, group)
            assert O.deliverer_key_of(gk) is None
        # equal content -> equal parse; different tenants -> different keys (RouteDetailCacheTest.java:99-127)
        assert O.parse_route_key(bytes(bytearray(k))) == O.parse_route_key(k)
        assert O.route_key_from_mqtt("tenant-" + str(uuid.uuid4()), f, url) != k
        # unsupportedFlagThrows (:129-141): a flag byte other than 1 / 2 / 3
        bad = bytearray(k)
        rlen = int.from_bytes(k[-2:], "big")
        bad[len(k) - 2 - rlen - 1] = 9
        assert O.parse_route_key(bytes(bad)) is None and B.decode_route_key(bytes(bad)) is None


def test_retain_mutation_abi_on_host_only_engine():
    """bmq_retain_apply_batch / info / live_ids / topics / expired / compact through the C ABI on a host-only engine: the same
    bmq_retain_core.h functions the gfx950 kernels run, on host threads (matching itself needs the device)."""
    e = B.Engine(device=-1)
    e.retain_rebuild(["t", "u"], [0, 0, 0, 1], ["a/b", "a/c", "$s/x", "q"], timestamps=[1 << 16, 2 << 16, 3 << 16, 4 << 16], expiry=[1, 2, 3, 4])
    assert e.retain_find_all()[0] == 4 and e.retain_live_ids() == [0, 1, 2, 3] and e.retain_live_ids("t") == [0, 1, 2]
    ids = e.retain_apply_batch(["t", "v"], [0, 0, 1, 0, 0], [(0, "a/d", 5 << 16, 9), (1, "a/b"), (0, "x/y"), (1, "nope"), (0, "$s/z")])
    assert ids.tolist() == [4, 1, 5, 0xFFFFFFFF, 6]  # new topics: the next unused ids; a removed topic reports its id; an absent one none
    assert e.retain_live_ids() == [0, 2, 3, 4, 5, 6] and e.retain_find_all()[0] == 6
    assert e.retain_topics([0, 2, 3, 4, 5, 6, 1]) == [("t", "$s/x"), ("t", "a/c"), ("u", "q"), ("t", "a/d"), ("v", "x/y"), ("t", "$s/z"), ("t", "a/b")]
    assert e.retain_topic_info(4) == (5 << 16, 9, 9005) and e.retain_topic_info(5) == (0, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF)
    with pytest.raises(B.BmqError):
        e.retain_topic_info(1)  # removed
    # the GC scan: one tenant = what match(tenant, "#") reaches (no '$' topics), all tenants = findAll()
    assert e.retain_expired("t", 10 ** 9) == [2, 4] and e.retain_expired(None, 10 ** 9) == [0, 2, 3, 4]
    info = e.retain_info()
    assert (info.n_topics, info.id_bound, info.loaded_topics, info.loaded_removed, info.added_ids, info.generation) == (6, 7, 4, 1, 3, 1)
    # ops on one topic inside a batch take effect in order; a re-added topic gets its id back
    ids = e.retain_apply_batch(["t"], None, [(0, "a/b"), (1, "a/d"), (0, "a/d"), (1, "a/d")])
    assert ids.tolist() == [1, 0xFFFFFFFF, 0xFFFFFFFF, 4] and e.retain_live_ids() == [0, 1, 2, 3, 5, 6]
    with pytest.raises(B.BmqError) as ei:
        e.retain_apply_batch(["t"], [0, 3], [(0, "ok"), (0, "bad-tenant-index")])
    assert ei.value.code == -1 and e.retain_find_all()[0] == 6  # nothing was changed
    e.retain_compact()
    live = e.retain_live_ids()
    assert live == list(range(6)) and e.retain_topics(live) == [("t", "$s/x"), ("t", "$s/z"), ("t", "a/b"), ("t", "a/c"), ("u", "q"), ("v", "x/y")]
    info = e.retain_info()
    assert (info.generation, info.added_ids, info.loaded_removed, info.n_tenants) == (2, 0, 0, 3)
    assert e.retain_topic_info(2)[0] == 0 and e.retain_topic_info(3) == (2 << 16, 2, 2002)  # stamps travel through the compaction
    # a fresh engine: the first add creates the index
    e2 = B.Engine(device=-1)
    assert e2.retain_apply_batch(["z"], None, [(0, "k/l")]).tolist() == [0] and e2.retain_topics([0]) == [("z", "k/l")]
    e2.close()
    e.close()


def test_expand_ranges_orders_rows_whose_ranges_interleave():
    """The consumer side of BMQ_FMT_RANGES (bifromq_amd.Engine.expand_ranges): direct ranges, side lists, empty ranges, and a row whose ranges
    overlap (only possible after churn) comes out ascending -- what the id CSR of the same batch holds."""
    import bifromq_amd as B
    S = B.Engine.RANGE_SIDE
    side = np.array([7, 9, 40, 41, 5], dtype=np.uint32)
    ranges = np.array([[10, 3], [0, 2 | S], [20, 0], [30, 2],      # row 0: 10..12, side[0:2] = 7, 9 -> overlap: ordered
                       [2, 2 | S], [50, 1],                          # row 1: 40, 41, 50
                       [4, 1 | S]], dtype=np.uint32)                 # row 3 (row 2 is empty): 5
    rptr = np.array([0, 4, 6, 6, 7], dtype=np.uint32)
    rows = B.Engine.expand_ranges(rptr, ranges, side, 4)
    assert [r.tolist() for r in rows] == [[7, 9, 10, 11, 12, 30, 31], [40, 41, 50], [], [5]]


def test_walk_geometry_caps_are_a_selector():
    """bmq_config.wave_queue_cap / wave_pair_cap select one of the walk kernel's two compiled LDS geometries: 0 (default) or 128 (smallest
    lists); a value that would silently run the default is refused."""
    B.Engine(device=-1, wave_queue_cap=128, wave_pair_cap=128).close()
    B.Engine(device=-1, wave_queue_cap=128).close()
    for bad in (dict(wave_queue_cap=192), dict(wave_pair_cap=160), dict(wave_queue_cap=4096, wave_pair_cap=4096), dict(wave_queue_cap=64)):
        with pytest.raises(B.BmqError) as ei:
            B.Engine(device=-1, **bad)
        assert ei.value.code == -1  # BMQ_E_INVAL


def test_region_slack_sizes_the_trie_regions():
    """bmq_config.region_slack: a tenant's region holds nodes x (1 + slack / 4) two-slot buckets -- 0 = the default (6: load factor 0.2), 1 = the layout of
    rounds 2-5 (0.4); more than 64 is refused.  Only the table size depends on it (the walk's results are compared on the GPU: tests/test_dist_gpu.py)."""
    keys = sorted(B.route_key("t%d" % (i % 3), "a/%d/+/x%d" % (i % 50, i), 1, "0\0r%d\0d" % i) for i in range(3000))
    slots = {}
    for slack in (0, 1, 6, 12):
        e = B.Engine(device=-1, region_slack=slack).rebuild(keys)
        info = e.info()
        slots[slack] = (int(info.trie_slots), int(info.n_nodes), int(info.n_routes))
        e.close()
    assert slots[0] == slots[6] and slots[1][1:] == slots[6][1:] == slots[12][1:]
    n_nodes = slots[6][1]
    assert slots[1][0] < slots[6][0] < slots[12][0]
    assert 2 * 1.2 * n_nodes <= slots[1][0] <= 2 * 1.3 * n_nodes + 64   # nodes x 1.25 buckets of two slots (+ a few buckets per tenant)
    assert 2 * 2.4 * n_nodes <= slots[6][0] <= 2 * 2.6 * n_nodes + 64   # nodes x 2.5
    with pytest.raises(B.BmqError) as ei:
        B.Engine(device=-1, region_slack=65)
    assert ei.value.code == -1  # BMQ_E_INVAL


def test_ops_of_one_batch_apply_in_order_also_when_the_filter_is_new():
    """tests/test_dist_gpu.py's test of the same name over the host executor (its par() runs the ops of a large batch on several threads)."""
    import bifromq_amd as B
    from oracle import oracle as O

    def normal(tenant, tf, recv):
        return B.route_key_from_mqtt(tenant, tf, O.receiver_url(0, recv, "d"))
    eng = B.Engine(device=-1)
    try:
        base = [normal("t1", "a/%d/+" % i, "r%d" % i) for i in range(200)]
        eng.rebuild(base)
        fresh = [normal("t%d" % (i % 3), "fresh/%d/x%d/#" % (i, i % 7), "f%d" % i) for i in range(20000)]
        stay = [normal("t1", "stay/%d" % i, "s%d" % i) for i in range(500)]
        ops = [(0, k) for k in fresh] + [(1, base[3]), (0, base[3]), (1, base[3])] + [(0, k) for k in stay] + [(1, k) for k in fresh] + \
              [(0, stay[5]), (1, stay[5]), (0, stay[5])]
        eng.apply(ops)
        live = sorted(set(base) - {base[3]} | set(stay))
        info = eng.info()
        assert info.n_routes == len(live)
        assert sorted(k for k in eng.route_keys(np.arange(int(info.next_route_id), dtype=np.uint32)) if k) == live
    finally:
        eng.close()


def test_ctypes_mirrors_have_the_layout_of_the_header(tmp_path):
    """bifromq_amd/_lib.py restates the structures of include/bmq.h as ctypes structures: same size, every field at the same offset (a C program
    prints what the compiler makes of the header)."""
    import ctypes as C
    import subprocess

    from bifromq_amd import _lib
    structs = {"bmq_config": _lib.Config, "bmq_stats": _lib.Stats, "bmq_index_info": _lib.IndexInfo, "bmq_batcher_config": _lib.BatcherConfig,
               "bmq_batcher_stats": _lib.BatcherStats, "bmq_poller_stats": _lib.PollerStats, "bmq_route_cache_config": _lib.RouteCacheConfig, "bmq_route_cache_stats": _lib.RouteCacheStats,
               "bmq_route_cache_tenant_stats": _lib.RouteCacheTenantStats, "bmq_retain_info": _lib.RetainInfo, "bmq_ranges_info": _lib.RangesInfo}
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "bmq.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0;', '}']
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True, text=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_retain_compaction_beside_the_serving_generation_over_the_host_executor():
    """bmq_retain_compact_begin / _build / _swap / _abort on a host-only engine (the retain index and its mutations run on the host executor;
    nothing is matched): the live set, the replay of what was added / removed between begin and swap -- in order --, ids = ranks after a
    second round, the refusals.  The rows a GPU engine serves through the same calls: tests/test_retain_churn_gpu.py."""
    import random
    rnd = random.Random(5)
    eng = B.Engine(device=-1)
    try:
        tenants = ["tA", "tB"]
        base = sorted({(rnd.randrange(2), "l%d/m%d/n%d" % (rnd.randrange(5), rnd.randrange(20), rnd.randrange(50))) for _ in range(1500)})
        eng.retain_rebuild(tenants, [t for t, _ in base], [p for _, p in base])
        live = {(tenants[t], p) for t, p in base}

        def live_now():
            out = {}
            for t in tenants:
                ids = eng.retain_live_ids(t)
                for i, k in zip(ids, eng.retain_topics(ids)):
                    out[k] = i
            return out

        def churn(n_rm, n_add, tag):
            rm = rnd.sample(sorted(live), n_rm)
            add = [(tenants[j % 2], "z%s/%d" % (tag, j)) for j in range(n_add)]
            for t in tenants:
                ops = [(1, p) for tt_, p in rm if tt_ == t] + [(0, p) for tt_, p in add if tt_ == t]
                eng.retain_apply(t, ops)
            live.difference_update(rm)
            live.update(add)
            return rm, add

        with pytest.raises(B.BmqError) as ei:
            eng.retain_compact_swap()
        assert ei.value.code == -7
        churn(200, 100, "a")
        g0 = eng.retain_info().generation
        eng.retain_compact_begin()
        with pytest.raises(B.BmqError):
            eng.retain_compact_begin()
        with pytest.raises(B.BmqError):
            eng.retain_compact_swap()  # not built yet
        rm, add = churn(50, 70, "b")
        eng.retain_compact_build()
        eng.retain_apply(rm[0][0], [(0, rm[0][1])])
        live.add(rm[0])
        eng.retain_apply(add[0][0], [(1, add[0][1])])
        live.discard(add[0])
        carried, replayed = eng.retain_compact_swap()
        assert carried == len(base) - 200 + 100 and replayed == 50 + 70 + 2
        assert set(live_now()) == live and eng.retain_info().generation == g0 + 1
        eng.retain_compact_begin().retain_compact_build()
        eng.retain_compact_swap()
        got = live_now()
        order = sorted(live, key=lambda k: (k[0].encode(), [lv.encode() for lv in k[1].split("/")]))
        assert [got[k] for k in order] == list(range(len(order)))
        info = eng.retain_info()
        assert info.loaded_removed == 0 and info.added_ids == 0 and info.n_topics == len(live)
        eng.retain_compact_begin()
        churn(3, 3, "c")
        eng.retain_compact_abort().retain_compact_abort()
        assert set(live_now()) == live
    finally:
        eng.close()
