"""dist-server side range pruning (TenantRangeLookupCache.lookup).  CPU tier: the function the GPU kernel runs
(bmq_range_core.h, compiled for the host by tests/c/range_shim.cpp) against the oracle's restatement over the structural
expansion iterator; GPU tier: bmq_range_lookup through the C ABI against the same oracle."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEVELS = ["a", "b", "c", "", "$sys", "$x", "0", "!", " ", "z", "a b", "你好", "dev", "x" * 20]
FLEVELS = LEVELS + ["+", "+", "#", "$", "~"]


def rand_topic(rnd, max_levels=5):
    return "/".join(rnd.choice(LEVELS) for _ in range(rnd.randint(1, max_levels)))


def rand_global_filter(rnd, tenant, topic, others):
    """mostly near-misses of the topic's own expansion filters, so that the seek has to work"""
    t = topic.split("/")
    p = rnd.random()
    if p < 0.5:
        lv = [rnd.choice([x, x, "+", rnd.choice(FLEVELS)]) for x in t[:rnd.randint(0, len(t))]]
        if rnd.random() < 0.4:
            lv.append("#")
        elif rnd.random() < 0.3:
            lv.append(rnd.choice(FLEVELS))
    else:
        lv = [rnd.choice(FLEVELS) for _ in range(rnd.randint(0, 5))]
    tn = tenant if rnd.random() < 0.8 else rnd.choice(others)
    return [tn] + lv


def pack_levels(lists):
    raw = [b"\0".join(x.encode() for x in lv) for lv in lists]
    off = np.zeros(len(raw) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(r) for r in raw])
    data = np.zeros(int(off[-1]) + 32, dtype=np.uint8)
    if off[-1]:
        data[:int(off[-1])] = np.frombuffer(b"".join(raw), dtype=np.uint8)
    return data, off


def make_case(rnd, n_topics, n_cand):
    tenant = rnd.choice(["tenantA", "t", "mm"])
    others = ["a", "tenantB", "zz", "s", "tenant"]
    topics = [rand_topic(rnd) for _ in range(n_topics)]
    cands = []
    for _ in range(n_cand):
        p = rnd.random()
        if p < 0.1:
            cands.append(None)
        elif p < 0.2:
            cands.append(())
        else:
            base = rnd.choice(topics)
            a, b = rand_global_filter(rnd, tenant, base, others), rand_global_filter(rnd, tenant, base, others)
            if "\0".join(a) > "\0".join(b):
                a, b = b, a
            cands.append((a, b))
    return tenant, topics, cands


def cand_arrays(cands):
    kind = np.array([0 if c is None else (1 if len(c) == 0 else 2) for c in cands], dtype=np.uint8)
    fd, fo = pack_levels([c[0] if c else [] for c in cands])
    ld, lo = pack_levels([c[1] if c else [] for c in cands])
    return kind, fd, fo, ld, lo


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("range") / "librange_shim.so")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-fPIC",
                    "-o", so, os.path.join(ROOT, "tests", "c", "range_shim.cpp")], check=True, capture_output=True, timeout=300)
    return so


def _run_shim(so, seed, rounds):
    code = r'''
import ctypes as C, sys, random, numpy as np
sys.path.insert(0, %r)
from tests.test_range import make_case, cand_arrays
from oracle import oracle as O
L = C.CDLL(%r)
rnd = random.Random(%d)
bad = 0
for r in range(%d):
    tenant, topics, cands = make_case(rnd, 12, rnd.randint(1, 9))
    kind, fd, fo, ld, lo = cand_arrays(cands)
    raw = [t.encode() for t in topics]
    toff = np.zeros(len(raw) + 1, dtype=np.uint32); toff[1:] = np.cumsum([len(x) for x in raw])
    tdata = np.frombuffer(b"".join(raw) + b"\0" * 32, dtype=np.uint8).copy()
    keep = np.zeros(len(topics) * len(cands), dtype=np.uint8)
    tb = tenant.encode()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.range_lookup_host(tb, len(tb), p(tdata), p(toff), len(topics), p(kind), p(fd), p(fo), p(ld), p(lo), len(cands), p(keep))
    for i, tp in enumerate(topics):
        exp = O.range_lookup(tenant, tp, cands)
        got = [c for c in range(len(cands)) if keep[i * len(cands) + c]]
        if got != exp:
            bad += 1
            if bad < 6: print("MISMATCH", repr(tenant), repr(tp), cands, "exp", exp, "got", got)
print("range_shim ok" if not bad else "range_shim FAILED %%d" %% bad)
''' % (ROOT, so, seed, rounds)
    env = dict(os.environ)
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0"
    return subprocess.run(["python3", "-c", code], capture_output=True, text=True, timeout=900, env=env)


# bifromq-dist/bifromq-dist-server/src/test/java/org/apache/bifromq/dist/server/scheduler/TenantRangeLookupCacheTest.java:106-330:
# (topic, candidates in boundary order, indices of the candidates lookup() returns).  None = no Fact, () = Fact without first / last.
_T = "tenantA"
REFERENCE_LOOKUP_CASES = [
    ("singleCandidateFullFactCovers :114-125", "m/n", [([_T, "a"], [_T, "z"])], [0]),
    ("singleCandidateFullFactNotCover :127-137", "a/b", [([_T, "m"], [_T, "z"])], []),
    ("singleCandidateMissingFirstOrLastIsEmptyRange :139-157", "a", [()], []),
    ("singleCandidateNoFactIsIncluded :159-169", "topic", [None], [0]),
    ("multiCandidatesTwoCoveringAndOneNot :171-195", "n/1", [([_T, "a"], [_T, "z"]), ([_T, "n"], [_T, "s"]), ([_T, "t"], [_T, "z"])], [0, 1]),
    ("multiCandidatesMixWithNoFact :197-220", "z/1", [None, ([_T, "x"], [_T, "z"]), ([_T, "z"], [_T, "zz"])], [0, 2]),
    ("multiCandidatesLastLessThanTopicThenFollowingCovers :222-238", "n/1", [([_T, "a"], [_T, "m"]), ([_T, "n"], [_T, "z"])], [1]),
    ("earlyStopTopicLessThanFirstOfFirstCandidate :240-255", "a", [([_T, "b"], [_T, "c"]), None], []),
    ("earlyStopAfterIncludingNoFactFirst :257-273", "a", [None, ([_T, "b"], [_T, "c"])], [0]),
    ("includeWhenEqualToFirstOrLast :275-288 (first)", "b", [([_T, "b"], [_T, "z"])], [0]),
    ("includeWhenEqualToFirstOrLast :275-288 (last)", "b", [([_T, "a"], [_T, "b"])], [0]),
    ("multiLevelTopicAndOrder :290-309", "a/b/c", [([_T, "a", "b"], [_T, "a", "z"]), ([_T, "b"], [_T, "c"])], [0]),
    ("cacheFunctionalConsistency :311-330 (a)", "n/1", [([_T, "a"], [_T, "m"]), ([_T, "n"], [_T, "z"])], [1]),
    ("cacheFunctionalConsistency :311-330 (b)", "n/1", [([_T, "x"], [_T, "z"])], []),
]


def test_reference_lookup_cases_oracle_and_core(shim):
    """The reference's own TenantRangeLookupCacheTest as known answers: the oracle restatement (structural expansion iterator) and the
    function the GPU kernel runs (bmq_range_core.h, host build under ASan) both return exactly the ranges the test expects."""
    for name, topic, cands, want in REFERENCE_LOOKUP_CASES:
        assert O.range_lookup(_T, topic, cands) == want, name
    code = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r)
from tests.test_range import REFERENCE_LOOKUP_CASES, cand_arrays
L = C.CDLL(%r)
bad = 0
for name, topic, cands, want in REFERENCE_LOOKUP_CASES:
    kind, fd, fo, ld, lo = cand_arrays(cands)
    tb, tp = b"tenantA", topic.encode()
    toff = np.array([0, len(tp)], dtype=np.uint32)
    tdata = np.frombuffer(tp + b"\0" * 32, dtype=np.uint8).copy()
    keep = np.zeros(len(cands), dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.range_lookup_host(tb, len(tb), p(tdata), p(toff), 1, p(kind), p(fd), p(fo), p(ld), p(lo), len(cands), p(keep))
    got = [c for c in range(len(cands)) if keep[c]]
    if got != want:
        bad += 1
        print("MISMATCH", name, got, want)
print("reference cases ok" if not bad else "reference cases FAILED")
''' % (ROOT, shim)
    env = dict(os.environ)
    env["LD_PRELOAD"] = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env["ASAN_OPTIONS"] = "detect_leaks=0"
    r = subprocess.run(["python3", "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "reference cases ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_range_core_matches_the_expansion_iterator(shim, seed):
    r = _run_shim(shim, seed, 250)
    assert r.returncode == 0 and "range_shim ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_range_lookup_on_gpu_matches_the_oracle(seed):
    """bmq_range_lookup (k_range_lookup, one lane per topic) through the C ABI against O.range_lookup."""
    import bifromq_amd as B
    from bifromq_amd import _lib
    from bifromq_amd.engine import pack
    eng = B.Engine(device=0)
    rnd = random.Random(seed)
    for _ in range(30):
        tenant, topics, cands = make_case(rnd, 200, rnd.randint(1, 12))
        kind, fd, fo, ld, lo = cand_arrays(cands)
        tdata, toff = pack(topics)
        keep = np.zeros(len(topics) * len(cands), dtype=np.uint8)
        tb = tenant.encode()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = _lib.lib().bmq_range_lookup(eng.h, tb, len(tb), p(tdata), p(toff), len(topics), p(kind), p(fd), p(fo), p(ld), p(lo), len(cands), p(keep))
        assert rc == 0
        for i, tp in enumerate(topics):
            assert [c for c in range(len(cands)) if keep[i * len(cands) + c]] == O.range_lookup(tenant, tp, cands), (tenant, tp, cands)
    # a 100-level topic: decided all the same (<= RL_MAX_LEVELS is exact, deeper is kept conservatively)
    deep = "/".join(["a"] * 60)
    cands = [(["t", "a"], ["t", "a", "a"]), (["t", "b"], ["t", "c"]), (["u"], ["u", "x"])]
    kind, fd, fo, ld, lo = cand_arrays(cands)
    tdata, toff = pack([deep])
    keep = np.zeros(3, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert _lib.lib().bmq_range_lookup(eng.h, b"t", 1, p(tdata), p(toff), 1, p(kind), p(fd), p(fo), p(ld), p(lo), 3, p(keep)) == 0
    assert [c for c in range(3) if keep[c]] == O.range_lookup("t", deep, cands)
    eng.close()

