"""bring-up probe of the persistent matcher: python tools/poll_probe.py <scenario> [hook] (hooks: experiment builds, BMQ_LIB=build/variants/libbmq_exp.so)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import bifromq_amd as B
from bifromq_amd.workload import unpack
sc = sys.argv[1] if len(sys.argv) > 1 else "single"
hook = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = B.Workload(21, 6, 1500, 1)
eng = B.Engine(device=0).rebuild(w.keys())
tn = w.tenants()
data, off, tt = w.topics(9, 6000)
topics = [t.decode() for t in unpack(data, off)]
if hook:
    eng.poller_control(100 + hook)
b = eng.batcher()
if sc == "single":
    for i in range(5):
        rows, epoch = b.match_all(tn[tt[i]], [topics[i]])
        print(i, len(rows[0]), flush=True)
elif sc == "set50":
    sel = [i for i in range(len(topics)) if tt[i] == 2][:50]
    rows, _ = b.match_all(tn[2], [topics[i] for i in sel])
    print("set50", sum(len(r) for r in rows), flush=True)
elif sc == "unknown":
    print(b.match_all("no-such-tenant", ["a/b"]), flush=True)
elif sc == "wedge":
    import time
    print(b.match_all(tn[tt[0]], [topics[0]])[0] is not None, eng.poller_stats().n_served, flush=True)
    eng.poller_control(eng.POLLER_TEST_IGNORE_DOORBELLS)
    t0 = time.time()
    r = b.match_all(tn[tt[1]], [topics[1]])
    st = eng.poller_stats()
    print("wedge call took %.3f s" % (time.time() - t0), "timeouts", st.n_timeouts, "unserved", st.n_unserved, "fallback", st.n_fallback, "enabled", st.enabled, "running", st.running, flush=True)
elif sc.startswith("threads"):
    n = int(sc[7:])
    cnt, hsh, sec = b.drive_singletons(tn, tt, (data, off), n_threads=n)
    print("threads", n, int(cnt.sum()), 6000 / sec, "calls/s", flush=True)
st = eng.poller_stats()
print(sc, "hook", hook, "served", st.n_served, "fallback", st.n_fallback, "unserved", st.n_unserved, "timeouts", st.n_timeouts, "bad_input", st.n_bad_input, "starts", st.n_starts, flush=True)
b.close()
eng.close()
print("done", flush=True)
