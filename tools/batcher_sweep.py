"""Experiment driver (not product): blocking single-topic calls/s through the batching front for a few settings of its knobs."""
import os, subprocess, sys, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import bifromq_amd as B
    threads = int(sys.argv[2])
    w = B.Workload(0xB1F2, 200, 5000, 1)
    eng = B.Engine(device=0).rebuild(w.keys())
    data, off, tt = w.topics(9, 200000)
    bt = eng.batcher()
    bt.drive_singletons(w.tenants(), tt[:20000], (data, off[:20001]), threads)
    cnt, hsh, sec = bt.drive_singletons(w.tenants(), tt, (data, off), threads)
    st = bt.stats()
    print(json.dumps({"threads": threads, "calls_per_s": len(tt) / sec, "topics_per_launch": st.n_topics / max(st.n_batches, 1)}))
    bt.close(); eng.close()
else:
    # python tools/batcher_sweep.py [threads,...] [inflight,...] [fanout,...]   (BMQ_LIB = a -DBMQ_EXPERIMENTS=1 build: the two knobs are environment switches there only)
    arg = lambda i, d: tuple(int(x) for x in sys.argv[i].split(",")) if len(sys.argv) > i else d
    for threads in arg(1, (64, 256)):
        for infl in arg(2, (1, 2, 3)):
            for fan in arg(3, (2, 4, 8)):
                env = dict(os.environ, BMQ_BATCHER_INFLIGHT=str(infl), BMQ_BATCHER_FANOUT=str(fan), PYTHONPATH=".")
                r = subprocess.run([sys.executable, __file__, "child", str(threads)], env=env, capture_output=True, text=True, timeout=120)
                print("inflight", infl, "fanout", fan, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
