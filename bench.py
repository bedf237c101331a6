#!/usr/bin/env python3
"""bench.py -- publish-topic matches/s of the MI355X engine on BASELINE.json's workload.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic publishes: tokenise + trie walk + CSR expand of
--topics (default 1M) publish topics against this rank's shard of the filter index, inputs resident in HBM, results
left in HBM (plus, for N > 1, the one exchange step of SURVEY.md 8e: an all-gather of every rank's CSR).
Workload (config.workload):
  c3 (default): 1000 tenants x 10k routes = 10M route keys (the index size BASELINE.json's metric is quoted at; it fits
      one GPU), Zipf(1.0) tenant popularity, 1M publishes per batch.  N > 1: tenants are partitioned across ranks by
      hash(tenantId) mod N (bifromq_amd/shard.py), every rank matches its own 1M-publish batch per step -> weak scaling
      in publishes, the 10M-route index is split N ways.
  c2: 1 tenant x 1M routes, 1M publishes (configs[1]).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


np = None


def main():
    global np
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "small", "c4"])
    ap.add_argument("--topics", type=int, default=1_000_000, help="publishes per batch per rank")
    ap.add_argument("--batches", type=int, default=4, help="distinct pre-generated batches cycled through")
    ap.add_argument("--ungrouped", action="store_true",
                    help="publishes in random tenant order instead of one DistPack per tenant (BatchDistRequest shape)")
    ap.add_argument("--churn", type=int, default=0,
                    help="configs[4]: apply this many route mutations (50%% subscribe / 50%% unsubscribe) between batches")
    ap.add_argument("--exchange-selftest", action="store_true",
                    help="run the N>1 code paths (exchange step, node-wide batch with partition) also at world size 1")
    ap.add_argument("--exchange", default="fanout", choices=["fanout", "ids", "none"],
                    help="N>1 exchange step: per-topic fan-out counts (what the reference sends upstream, DistWorkerCoProc.java:535-538) "
                         "or the complete CSR as an all-gatherv of route ids")
    ap.add_argument("--exchange-impl", default="lib", choices=["torch", "lib"],
                    help="who issues the collectives: libbmq itself (bmq_exchange_*: RCCL loaded by the library, on the engine's exchange "
                         "stream) or torch.distributed (RCCL through PyTorch)")
    ap.add_argument("--csr-exchange-steps", type=int, default=5,
                    help="N>1: extra steps, outside the timed region, with the OTHER exchange form (the all-gatherv of the complete CSR that "
                         "north_star names when --exchange is fanout, and vice versa): reported as exchange.csr_ms / exchange.fanout_ms")
    ap.add_argument("--split-policy", default="fanout-hinter", choices=["fanout-hinter", "publish-share"],
                    help="node-wide leg: which tenants are split by filter over all ranks.  fanout-hinter: bifromq_amd.shard.FanoutSplitHinter "
                         "(DW/hinter/FanoutSplitHinter.java restated: a prefix whose route count reaches --split-threshold) fed with the bulk load; "
                         "publish-share: tenants whose share of the batch exceeds half a rank's fair share (no analogue in the reference: its "
                         "hinters look at mutations only)")
    ap.add_argument("--split-threshold", type=int, default=100_000, help="FanoutSplitHinterFactory.java:33: splitThreshold default")
    ap.add_argument("--node-batch-steps", type=int, default=10,
                    help="N>1: steps of the extra node-wide measurement (one shared Zipf batch, device-side partition, hot tenants "
                         "split by filter, fan-out all-reduce); 0 = skip")
    ap.add_argument("--dedup", action="store_true", help="de-duplicate every batch on the device first (bmq_config.dedup_min_topics = 1; default: never)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ordered-only", action="store_true",
                    help="after the step loop only the ordered-batch leg (extra.ordered_batch): what tools/ordered_collect.py profiles under rocprofv3")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: skip the compact extra legs (configs[4] churn on this index, the batching front at 64 threads, C2 and C4 as "
                         "child runs) that the default run appends under `extra`")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-visible (PCIe-inclusive) measurement")
    ap.add_argument("--compact-chunk", type=int, default=8192, help="compaction leg: route ids handed to the next generation's builder per bmq_compact_poll")
    ap.add_argument("--compact-duty", type=float, default=0.5, help="compaction leg: share of its time the compacting thread spends inside bmq_compact_poll")
    ap.add_argument("--no-churn", action="store_true", help="--workload c4: skip the add / remove leg (A/B runs of the walk kernel)")
    ap.add_argument("--batcher-threads", type=int, default=-1,
                    help="also measure the batching front (bmq_batcher_*, SURVEY 8f-1): N native threads issue single-topic calls")
    ap.add_argument("--batcher-topics", type=int, default=200_000)
    ap.add_argument("--cpu-sample-tenants", type=int, default=1_000_000, help="the CPU baseline / parity leg takes the publishes of the first N tenants (default: all of them, the whole batch)")
    ap.add_argument("--cpu-sample-topics", type=int, default=1_000_000)
    args = ap.parse_args()
    if args.ordered_only:
        args.no_host_path = args.no_cpu_baseline = True
        args.batcher_threads = 0

    import numpy
    import torch

    import bifromq_amd as B

    np = numpy
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dist = None
    # BMQ_BENCH_ONE_GPU=1 (self-test of the N > 1 code path on a 1-GPU box, tools/selftest_two_ranks.sh): every rank works on GPU 0 and the
    # collectives go through gloo (two ranks on one GPU cannot share an RCCL communicator) -- the numbers mean nothing, the code path is the point
    one_gpu = os.environ.get("BMQ_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        args.exchange_impl = "torch"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.exchange_selftest:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.workload == "c4":
        return bench_retain(args, rank, world, local_rank, dev, dist)

    # ---- workload: this rank's shard ----------------------------------------------------------------------------------
    if args.workload == "c3":
        total_tenants, per_tenant, mode, seed = 1000, 10_000, 1, 0xB1F20003
        name = "C3: 1000 tenants x 10k routes (10M route keys, +/# mix), Zipf publishes, %d-publish batches" % args.topics
    elif args.workload == "c2":
        total_tenants, per_tenant, mode, seed = 1, 1_000_000, 1, 0xB1F20002
        name = "C2: 1 tenant x 1M routes (+/# mix), %d-publish batches" % args.topics
    else:
        total_tenants, per_tenant, mode, seed = 8, 5_000, 1, 0xB1F20009
        name = "small: 8 tenants x 5k routes"
    t0 = time.time()
    if total_tenants >= world and world > 1:
        # the index shards by hash(tenantId) mod N (bifromq_amd/shard.py); publishes follow their tenant
        from bifromq_amd import shard
        mine = [t for t in range(total_tenants) if shard.tenant_rank("tenant%06d" % t, world) == rank]
        w = B.Workload(seed, len(mine), per_tenant, mode, tenant_ids=mine)
    else:  # one GPU, or a single-tenant config on several GPUs: replicas (documented in DESIGN.md)
        w = B.Workload(seed, total_tenants, per_tenant, mode)
    t_gen = time.time() - t0
    eng = B.Engine(device=local_rank, kernel_timing=True,  # HIP events around k_walk / k_expand: the roofline needs the kernel time
                   dedup_min_topics=1 if args.dedup else 0)
    kb, ko = w.keys_packed()
    # bmq_rebuild, three times each way (a fresh box's first call pays for page faults of the staging copy and the first builder launches;
    # the docs quoted a best run, the driver saw 3-5x more): min and median are reported
    t_builds = []
    for _ in range(3 if world == 1 else 1):
        t0 = time.time()
        eng.rebuild_raw(kb.ctypes.data, ko.ctypes.data, w.n_keys)
        t_builds.append(time.time() - t0)
    t_build = t_builds[0]
    # the same load from page-locked buffers (what a JNI caller hands over as direct buffers from bmq_host_alloc): the upload is
    # DMA at PCIe speed instead of a staged copy of pageable memory
    t_builds_pinned = []
    if world == 1 and not args.no_host_path:
        pk, po_ = torch.from_numpy(kb).pin_memory(), torch.from_numpy(ko.astype(np.int32)).pin_memory()
        for _ in range(3):
            t0 = time.time()
            eng.rebuild_raw(pk.data_ptr(), po_.data_ptr(), w.n_keys)
            t_builds_pinned.append(time.time() - t0)
        del pk, po_
    info = eng.info()

    tdata, toff = w.tenants_packed()
    d_tenants = torch.from_numpy(tdata.copy()).to(dev)
    d_tenant_off = torch.from_numpy(toff.astype(np.int32)).to(dev)
    n_tenants = w.n_tenants
    n = args.topics
    batches = []
    for b in range(args.batches):
        data, off, tt = w.topics(seed + 1000 * (rank + 1) + b, n, grouped=not args.ungrouped)
        batches.append((torch.from_numpy(data).to(dev), torch.from_numpy(off.astype(np.int32)).to(dev),
                        torch.from_numpy(tt.astype(np.int32)).to(dev), (data, off, tt) if b == 0 else None))
    cap = 16 * n
    # Result buffers are double-buffered: the exchange of batch i (RCCL, on its own stream) overlaps the match of batch
    # i + 1 (engine stream), so a step costs max(match, exchange) instead of their sum; every exchange still completes
    # inside the timed region (barrier() synchronises the device).
    NBUF = 2 if dist is not None else 1
    d_row = [torch.zeros(n + 1, dtype=torch.int32, device=dev) for _ in range(NBUF)]
    d_ids = [torch.zeros(cap, dtype=torch.int32, device=dev) for _ in range(NBUF)]
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    ex_stream = torch.cuda.Stream(device=dev) if dist is not None else None
    ex_done = [None] * NBUF
    use_lib_ex = dist is not None and args.exchange_impl == "lib" and args.exchange != "none"
    if use_lib_ex:  # the library's own RCCL communicator: rank 0 makes the id, torch.distributed only carries its 128 bytes
        import ctypes as C
        uid = C.create_string_buffer(128)
        have_id = rank != 0 or B._lib.lib().bmq_comm_unique_id(uid) == 0
        box = [uid.raw if have_id else b""]  # (no id on rank 0, e.g. no librccl: every rank learns it and takes the torch path)
        dist.broadcast_object_list(box, src=0)
    if use_lib_ex and len(box[0]) != 128:
        use_lib_ex = False
    if use_lib_ex:
        # ncclCommInitRank is collective: it runs in a helper thread so that a rank that cannot reach the others ends the run with a
        # message instead of hanging it; a failure on ANY rank sends ALL ranks to the torch.distributed path (agreed by an all-reduce)
        import threading
        init_rc = [None]

        def _init():
            init_rc[0] = B._lib.lib().bmq_comm_init(eng.h, world, rank, box[0])

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(180)
        if th.is_alive():
            sys.stderr.write("bench.py: bmq_comm_init (ncclCommInitRank of the library's communicator) did not return within 180 s on rank %d\n" % rank)
            sys.stderr.flush()
            os._exit(3)
        ok = torch.tensor([1 if init_rc[0] == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if rank == 0:
                sys.stderr.write("bench.py: the in-library RCCL communicator could not be set up (rc %s: %s) -- exchange through torch.distributed instead\n"
                                 % (init_rc[0], B._lib.lib().bmq_last_error(eng.h)))
            use_lib_ex = False
    if use_lib_ex:
        d_counts_all = [torch.zeros(world * n, dtype=torch.int32, device=dev) for _ in range(NBUF)]
        d_rows_all = [torch.zeros(world * (n + 1), dtype=torch.int32, device=dev) for _ in range(NBUF)]
        d_ids_all = [torch.zeros(world * cap, dtype=torch.int32, device=dev) for _ in range(NBUF)]
        h_totals = np.zeros(world, dtype=np.uint64)
    torch.cuda.synchronize()

    ex_mode = [args.exchange]  # the exchange form of the steps being run (the extra pass below switches it)

    def step(i):
        nonlocal cap
        bt = batches[i % len(batches)]
        k = i % NBUF
        if ex_done[k] is not None:  # the exchange that still reads this buffer pair (issued two steps ago)
            if use_lib_ex:
                B._lib.lib().bmq_exchange_wait(eng.h)
            else:
                ex_done[k].synchronize()
            ex_done[k] = None
        while True:
            eng.match_batch_device(d_tenants.data_ptr(), d_tenant_off.data_ptr(), n_tenants, bt[2].data_ptr(),
                                   bt[0].data_ptr(), bt[1].data_ptr(), n, d_row[k].data_ptr(), d_ids[k].data_ptr(),
                                   d_ids[k].numel(), d_total.data_ptr())
            try:
                total = eng.finish()  # stream sync + counters
                break
            except B.BmqError as ex:
                if ex.code != -3:
                    raise
                cap = int(d_total.item()) * 2  # only during warm-up in practice
                d_ids[k] = torch.zeros(cap, dtype=torch.int32, device=dev)
        if dist is not None and ex_mode[0] != "none":  # the one exchange step over RCCL/xGMI, on its own stream
            from bifromq_amd import shard
            if d_ids[k].numel() < total:
                raise RuntimeError("id buffer smaller than the batch result")
            if use_lib_ex:  # asynchronous on the engine's exchange stream, behind this batch, overlapping the next one
                L = B._lib.lib()
                if ex_mode[0] == "fanout":
                    rc = L.bmq_exchange_fanout(eng.h, d_row[k].data_ptr(), n, d_counts_all[k].data_ptr())
                else:
                    if d_ids_all[k].numel() < world * d_ids[k].numel():  # the result buffer grew during warm-up
                        d_ids_all[k] = torch.zeros(world * d_ids[k].numel(), dtype=torch.int32, device=dev)
                    rc = L.bmq_exchange_csr(eng.h, d_row[k].data_ptr(), d_ids[k].data_ptr(), n, total, d_rows_all[k].data_ptr(),
                                            d_ids_all[k].data_ptr(), d_ids_all[k].numel(), h_totals.ctypes.data)
                if rc:
                    raise RuntimeError("bmq_exchange failed: %d %s" % (rc, L.bmq_last_error(eng.h)))
                ex_done[k] = True
                return total
            with torch.cuda.stream(ex_stream):  # results are complete (finish() synchronised the engine stream)
                if ex_mode[0] == "fanout":  # 4 B per topic: every rank learns every topic's fan-out
                    shard.exchange_counts_weak(dist, d_row[k], world)
                else:  # all-gatherv of the complete CSR: exact sizes, one grouped broadcast per rank
                    shard.exchange_csr_v(dist, d_row[k], d_ids[k], total, world)
                ex_done[k] = torch.cuda.Event()
                ex_done[k].record(ex_stream)
        return total

    churn_ms = []
    churn_batches = []
    if args.churn:
        # configs[4]: per step one batch of N route mutations -- N/2 unsubscribes of existing routes (each route at most once over
        # the run) + N/2 subscribes of new filters -- generated BEFORE the timed region into pinned host buffers (what a JNI caller
        # hands over as direct ByteBuffers); the timed region holds the bmq_routes_apply calls themselves.
        from bifromq_amd.engine import pack
        rng = np.random.default_rng(1234 + rank)
        kb_h, ko_h = w.keys_packed()
        mv = memoryview(kb_h)
        n_steps_total = args.warmup + args.steps
        perm = rng.permutation(w.n_keys)[:n_steps_total * (args.churn // 2)]
        tenants_l = w.tenants()
        q = 0
        for b in range(n_steps_total):
            dels = perm[b * (args.churn // 2):(b + 1) * (args.churn // 2)]
            keys_b = [bytes(mv[int(ko_h[i]):int(ko_h[i + 1])]) for i in dels]
            ops_b = [1] * len(keys_b)
            for _ in range(args.churn - args.churn // 2):
                q += 1
                t = tenants_l[int(rng.integers(0, len(tenants_l)))]
                keys_b.append(B.route_key(t, "churn/l1_%d/+/l3_%d" % (q % 64, q % 4096), 1, "0\0c%d\0d%d" % (q, q % 64)))
                ops_b.append(0)
            order = rng.permutation(len(keys_b))  # subscribes and unsubscribes interleaved, as they arrive
            data, off = pack([keys_b[i] for i in order])
            opb = np.array([ops_b[i] for i in order], dtype=np.uint8)
            churn_batches.append(tuple(torch.from_numpy(x).pin_memory() for x in (data, off, opb)))

    def churn(i):
        """configs[4]: apply the i-th pre-generated mutation batch (bmq_routes_apply: builder kernels on the engine stream)."""
        if not args.churn:
            return
        data, off, opb = churn_batches[i]
        t0c = time.perf_counter()
        rc = B._lib.lib().bmq_routes_apply(eng.h, data.data_ptr(), off.data_ptr(), opb.data_ptr(), len(opb))
        churn_ms.append((time.perf_counter() - t0c) * 1e3)
        if rc:
            raise RuntimeError("bmq_routes_apply failed: %d %s" % (rc, B._lib.lib().bmq_last_error(eng.h)))

    def barrier():
        if use_lib_ex:
            B._lib.lib().bmq_exchange_wait(eng.h)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Python's cyclic garbage collector is kept out of every timed region of this script.  The workload generator leaves millions of small
    # objects behind (10 M route keys, 1 M topics as lists of bytes); a generation-2 collection that walks them takes 100-250 ms, and one of
    # them inside a 5-step leg is how the driver's round-5 line came to read 18.99 ms per pipelined C5 step (the builder's: 0.79) and how
    # profiles/r06's first full line read p99 1.60 ms / 2.59 G topics/s for a loop whose kernels take 0.28 ms: what exists now is frozen
    # (never scanned again), what the legs allocate later is collected by reference counting (no cycles are built).
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    for i in range(args.warmup):
        churn(i)
        step(i)
    churn_ms.clear()
    barrier()
    lat = []
    walk_ms, expand_ms, total_ms = [], [], []
    alg_bytes, walk_own, expand_own = [], [], []
    n_match = n_visit = n_slow = 0
    t_start = time.perf_counter()
    for i in range(args.steps):
        churn(args.warmup + i)  # inside the timed region when --churn is given
        ts = time.perf_counter()
        step(i)
        lat.append((time.perf_counter() - ts) * 1e3)
        st = eng.stats()  # HIP-event times recorded on the engine's stream for this batch
        walk_ms.append(st.ms_walk)
        expand_ms.append(st.ms_expand)
        total_ms.append(st.ms_total)
        # ALGORITHMIC bytes: len(topic) + 8 + 32 * N_visit + 4 * N_match, summed over the batch
        alg_bytes.append(st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit + 4 * st.n_match)
        walk_own.append(st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit)
        expand_own.append(4 * st.n_match)
        n_match += st.n_match
        n_visit += st.n_visit
        n_slow += st.n_slow_topics
    barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # the OTHER exchange form, a few steps outside the timed region: north_star's "all-gatherv of (topic -> matched-route) pairs" next to
    # the fan-out counts the reference really sends upstream
    other_ms = None
    if dist is not None and args.exchange != "none" and args.csr_exchange_steps > 0:
        ex_mode[0] = "ids" if args.exchange == "fanout" else "fanout"
        step(0)
        barrier()
        t_o = time.perf_counter()
        for i in range(args.csr_exchange_steps):
            step(i)
        barrier()
        t_other = time.perf_counter() - t_o
        tmax = torch.tensor([t_other], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        other_ms = float(tmax.item()) / args.csr_exchange_steps * 1e3
        ex_mode[0] = args.exchange
    node = None
    if dist is not None and args.node_batch_steps > 0 and args.workload == "c3":
        node = node_batch(args, rank, world, local_rank, dev, dist, total_tenants, per_tenant, mode, seed)
    # The engine's production default records no events between its kernels (they cost ~4 us each, ~16 us per batch): the same
    # K steps once more without them.  `value` above keeps the events on, because the roofline wants the kernel time of exactly the
    # timed steps; this figure is what a caller that does not ask for kernel times gets.
    eng.set_kernel_timing(False)
    for i in range(min(args.warmup, 2)):
        step(i)
    barrier()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed_plain = time.perf_counter() - t1
    if dist is not None:
        tmax = torch.tensor([elapsed_plain], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed_plain = float(tmax.item())
    eng.set_kernel_timing(True)
    # Per rank, next to the whole-job value: with the index sharded N ways every rank's k_walk runs over 1 / N of the
    # 10 M route keys -- it hits its L2 more often, and a curve that reads super-linear explains itself only with these beside it.
    per_rank = None
    if dist is not None:
        mine = torch.tensor([float(np.mean(walk_ms)), float(np.mean(expand_ms)), float(np.mean(total_ms)), float(info.device_bytes), float(info.n_routes),
                             float(info.n_nodes), float(n_visit) / max(1, args.steps), float(n_match) / max(1, args.steps), float(np.mean(lat))],
                            dtype=torch.float64, device=dev)
        allr = torch.zeros(world, mine.numel(), dtype=torch.float64, device=dev)  # (a sum over rows of which every rank fills its own: any backend does it)
        allr[rank] = mine
        dist.all_reduce(allr)
        per_rank = [{"rank": r, "kernel_ms": {"k_walk": float(v[0]), "k_expand": float(v[1]), "all_kernels": float(v[2])}, "index_bytes_this_rank": int(v[3]),
                     "route_keys_this_rank": int(v[4]), "trie_nodes_this_rank": int(v[5]), "n_visit_per_batch": float(v[6]), "n_match_per_batch": float(v[7]),
                     "ms_per_step_this_rank": float(v[8])} for r, v in enumerate(x.tolist() for x in allr)]
    if rank != 0:
        dist.destroy_process_group()
        return

    steps = args.steps
    value = world * n * steps / elapsed
    k_walk_ms = float(np.mean(walk_ms))
    k_exp_ms = float(np.mean(expand_ms))
    dom_name, dom_ms = ("k_walk", k_walk_ms) if k_walk_ms >= k_exp_ms else ("k_expand", k_exp_ms)
    # ONE convention in every leg of this line: `frac` = the dominant kernel's OWN algorithmic bytes / its time,
    # `frac_batch` = the whole batch's SURVEY 8d bytes / the same time.  Own bytes: k_walk reads the topics and walks the trie
    # (len + 8 + 32 * N_visit), k_expand writes the ids (4 * N_match).
    own = {"k_walk": float(np.mean(walk_own)), "k_expand": float(np.mean(expand_own))}
    achieved = own[dom_name] / (max(dom_ms, 1e-9) * 1e-3) / 1e9      # GB/s
    achieved_batch = float(np.mean(alg_bytes)) / (max(dom_ms, 1e-9) * 1e-3) / 1e9
    out = {
        "metric": "publish-topic matches/sec (whole node)",
        "value": value,
        "unit": "topics/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": name, "total_route_keys": total_tenants * per_tenant, "route_keys_this_rank": int(info.n_routes),
                   "tenants_this_rank": int(info.n_tenants), "trie_nodes_this_rank": int(info.n_nodes),
                   "index_bytes_this_rank": int(info.device_bytes), "publishes_per_batch_per_rank": n,
                   "batch_order": "random" if args.ungrouped else "grouped by tenant (one DistPack per tenant)",
                   "in_batch_dedup": ("on: identical (tenant, topic) rows of a batch are walked once (k_dedup / k_fill), every row keeps its row; N_visit and "
                                      "the algorithmic bytes count every row" if args.dedup else "off (the engine's default)"),
                   "parallelism": "tenant-shard x%d" % world if world > 1 else "single GPU",
                   "exchange_impl": None if dist is None else ("libbmq (bmq_exchange_*, RCCL loaded by the library)" if use_lib_ex else "torch.distributed (nccl backend = RCCL)"),
                   "exchange": ("none" if dist is None or args.exchange == "none" else
                                "RCCL all-gather of per-topic fan-out counts (4 B/topic; the CSR stays on the GPU that matched: the reference "
                                "replies fan-out per topic, DistWorkerCoProc.java:535-538), overlapped with the next batch's match"
                                if args.exchange == "fanout" else
                                "RCCL all-gatherv of the complete CSR (row_ptr + route ids, exact sizes, grouped per-rank broadcasts), "
                                "overlapped with the next batch's match")},
        "exchange": None if dist is None or args.exchange == "none" else {
            ("fanout_ms" if args.exchange == "fanout" else "csr_ms"): elapsed / steps * 1e3,
            ("csr_ms" if args.exchange == "fanout" else "fanout_ms"): other_ms,
            "note": "ms per step (match + exchange, max over ranks) with the fan-out exchange (all-gather of 4 B per topic) and with the "
                    "all-gatherv of the complete CSR (totals, row pointers, then one exact-size broadcast per rank in one group); the first "
                    "is the timed region of `value`, the second a few extra steps"},
        "value_without_kernel_timing": world * n * steps / elapsed_plain if not args.churn else None,
        "ms_per_step_without_kernel_timing": elapsed_plain / steps * 1e3 if not args.churn else None,
        "p99_batch_ms": float(np.percentile(lat, 99)),
        "p50_batch_ms": float(np.percentile(lat, 50)),
        "routes_per_topic": n_match / (n * steps),
        "visits_per_topic": n_visit / (n * steps),
        "slow_path_topics_per_batch": n_slow / steps,
        "churn": {"ops_per_batch": args.churn, "apply_ms_mean": float(np.mean(churn_ms)) if churn_ms else None,
                  "apply_ms_p99": float(np.percentile(churn_ms, 99)) if churn_ms else None,
                  "note": "bmq_routes_apply before every match batch, inside the timed region: ops uploaded from pinned host memory, "
                          "parsed and applied by the builder kernels on the engine stream (prepare, locate, sort, group); "
                          "time of the C-ABI call alone (returns when the device has applied the batch)"},
        "kernel_ms": {"k_walk": k_walk_ms, "k_expand": float(np.mean(expand_ms)), "all_kernels": float(np.mean(total_ms))},
        "host_s": {"generate": t_gen, "rebuild": float(np.median(t_builds)), "rebuild_min": float(np.min(t_builds)), "rebuild_first_call": t_build,
                   "rebuild_from_pinned_keys": float(np.median(t_builds_pinned)) if t_builds_pinned else None,
                   "rebuild_from_pinned_keys_min": float(np.min(t_builds_pinned)) if t_builds_pinned else None,
                   "note": "bmq_rebuild = upload of the route keys + the GPU builder kernels (bulk load); wall time of the C-ABI call, median / min "
                           "of 3 calls each (pageable, then page-locked keys); rebuild_first_call = the very first one on the fresh process"},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": None,
                     "own_algorithmic_bytes_per_launch": own[dom_name],
                     "own_bytes_rule": "k_walk: len(topic) + 8 + 32 * N_visit per topic; k_expand: 4 * N_match",
                     # the whole batch's SURVEY 8d bytes (len + 8 + 32 * N_visit + 4 * N_match) against the dominant kernel's time (what
                     # `frac` meant up to round 3), against ALL kernels of a batch, and against the whole step (host sync included)
                     "frac_batch": achieved_batch / 8000.0,
                     "frac_pipeline": float(np.mean(alg_bytes)) / (float(np.mean(total_ms)) * 1e-3) / 8e12,
                     "frac_step": float(np.mean(alg_bytes)) / (elapsed / steps) / 8e12,
                     "algorithmic_bytes_per_launch": float(np.mean(alg_bytes)),
                     "kernel_own_frac": {k: own[k] / (max(v, 1e-9) * 1e-3) / 8e12 for k, v in (("k_walk", k_walk_ms), ("k_expand", k_exp_ms))}},
    }
    if args.topics == 1_000_000 and not args.ungrouped and not args.dedup and not args.churn:
        attach_traffic(out, args.workload, world)
    else:  # the counter passes ran the default batch: their bytes per launch say nothing about another batch size / order
        out["roofline"]["traffic_source"] = "profiles/traffic_*.json was measured with the default batch (1 M grouped publishes, no churn): not reported for this run"

    if node is not None:
        out["node_batch"] = node
    if per_rank is not None:
        out["per_rank"] = per_rank  # (every rank's own kernel times and shard size: what a super-linear weak-scaling curve has to be read with)
    if world == 1 and not args.no_host_path:
        out["fanout_group"] = fanout_group_leg(eng, d_row[0], d_ids[0], n, dev, torch, np)
    if world == 1 and not args.no_host_path:
        ids_per_batch = n_match / steps
        if ids_per_batch * 4 <= 1 << 30:
            out["host_visible"] = host_visible(args, eng, w, batches, n, seed, rank, int(ids_per_batch * 1.25) + 4096)
        else:
            out["host_visible"] = {"skipped": "a batch returns %.1f GB of route ids: the host-visible rate is PCIe bandwidth / that" % (ids_per_batch * 4 / 1e9)}
    if args.batcher_threads < 0:  # default: the batching front at 64 threads is part of the N = 1 line (unless the extras are off)
        args.batcher_threads = 64 if (world == 1 and not args.no_extras and not args.no_host_path and args.workload == "c3") else 0
    if args.batcher_threads and world == 1:  # the production call pattern (one topic per call, many threads) through the collector
        hdata, hoff, htt = batches[0][3]
        m = min(args.batcher_topics, n)
        sub = (hdata, hoff[:m + 1].copy())
        # the same calls twice: generations launched one by one (rounds 2-5), then through the persistent matcher (round 6: resident waves
        # poll a request ring, bmq_poll_kernel.h) -- the default; `calls_per_s` is the default's
        eng.poller_control(eng.POLLER_DISABLE)
        bt = eng.batcher()
        cnt0, hsh0, sec0 = bt.drive_singletons(w.tenants(), htt[:m], sub, args.batcher_threads)
        bs0 = bt.stats()
        bt.close()
        eng.poller_control(eng.POLLER_ENABLE)
        p0 = eng.poller_stats()
        bt = eng.batcher()
        cnt, hsh, sec = bt.drive_singletons(w.tenants(), htt[:m], sub, args.batcher_threads)
        bs = bt.stats()
        p1 = eng.poller_stats()
        out["batching_front"] = {"threads": args.batcher_threads, "host_cpus_granted": effective_cpus(), "single_topic_calls": m, "calls_per_s": m / sec,
                                 "launches": int(bs.n_batches), "mean_topics_per_launch": bs.n_topics / max(1, bs.n_batches),
                                 "max_topics_per_launch": int(bs.max_batch_topics), "ids_returned": int(cnt.sum()),
                                 "persistent_matcher": {"generations_served": int(p1.n_served - p0.n_served), "handed_back_or_unserved": int(p1.n_fallback - p0.n_fallback),
                                                        "k_poll_launches": int(p1.n_starts - p0.n_starts), "timeouts": int(p1.n_timeouts)},
                                 "with_a_launch_per_generation": {"calls_per_s": m / sec0, "launches": int(bs0.n_batches),
                                                                  "mean_topics_per_launch": bs0.n_topics / max(1, bs0.n_batches),
                                                                  "same_rows": bool((cnt0 == cnt).all() and (hsh0 == hsh).all())},
                                 "note": "bmq_batcher_match_all, blocking callers: a generation holds at most one topic per thread"}
        bt.close()
        bt = eng.batcher()  # asynchronous side: 4 submitting threads, nobody blocks on the GPU
        cnt2, hsh2, sec2 = bt.drive_singletons(w.tenants(), htt[:m], sub, 4, asynchronous=True)
        bs = bt.stats()
        out["batching_front"]["async_submit"] = {"threads": 4, "calls_per_s": m / sec2, "launches": int(bs.n_batches),
                                                 "mean_topics_per_launch": bs.n_topics / max(1, bs.n_batches),
                                                 "max_topics_per_launch": int(bs.max_batch_topics),
                                                 "rows_equal_blocking_path": bool((cnt2 == cnt).all() and (hsh2 == hsh).all())}
        bt.close()
        # the route cache in front of it (bmq_route_cache_*: TenantRouteCache + TopicIndex on the engine's side): the same calls, twice over
        # the same topics -- pass 1 loads through the batching front, pass 2 is answered on the host
        try:
            bt = eng.batcher()
            rc_ = B.RouteCache(bt, max_routes_per_tenant=1 << 40)
            cnt3, hsh3, secs = rc_.drive(w.tenants(), htt[:m], sub, args.batcher_threads, passes=2)
            cs = rc_.stats()
            out["batching_front"]["route_cache"] = {
                "threads": args.batcher_threads, "calls_per_s_first_pass": m / secs[0], "calls_per_s_second_pass": m / secs[1],
                "hits": int(cs.hits), "misses": int(cs.misses), "entries": int(cs.entries), "cached_routes": int(cs.cached_routes),
                "rows_equal_blocking_path": bool((cnt3 == cnt).all() and (hsh3 == hsh).all()),
                "note": "bmq_route_cache_get: ISubscriptionCache.get per (tenant, topic); Zipf repeats inside pass 1 already hit"}
            rc_.close()
            bt.close()
            bt = eng.batcher()  # the future-shaped call from 4 threads: nobody blocks, misses ride the asynchronous side of the front
            rc_ = B.RouteCache(bt, max_routes_per_tenant=1 << 40)
            cnt4, hsh4, secs4 = rc_.drive(w.tenants(), htt[:m], sub, 4, passes=2, asynchronous=True)
            out["batching_front"]["route_cache"]["get_async"] = {
                "threads": 4, "calls_per_s_first_pass": m / secs4[0], "calls_per_s_second_pass": m / secs4[1],
                "launches": int(bt.stats().n_batches), "rows_equal_blocking_path": bool((cnt4 == cnt).all() and (hsh4 == hsh).all())}
            rc_.close()
            bt.close()
        except Exception as ex:  # noqa: BLE001
            out["batching_front"]["route_cache"] = {"error": repr(ex)}
    if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N=1 only (rank 0's host cores)
        step(0)  # batch 0 once more: its CSR (device) is compared with the restatement's rows inside the CPU leg
        torch.cuda.synchronize()
        out["cpu_baseline"] = cpu_baseline(args, w, batches[0][3], n, (d_row[0].cpu().numpy().view(np.uint32), d_ids[0].cpu().numpy().view(np.uint32)))
    if world == 1 and not args.no_extras and not args.churn and args.workload == "c3":
        def fetch_csr(i):  # batch i once more on the index as it is now -> (host batch, engine CSR): the C5 leg's parity check
            step(i)
            torch.cuda.synchronize()
            k = i % NBUF
            return batches[i % len(batches)][3], (d_row[k].cpu().numpy().view(np.uint32), d_ids[k].cpu().numpy().view(np.uint32))
        import ctypes as C_

        def submit_dev(i):  # the same batch as a TICKET (bmq_match_submit_dev): the engine is not taken, an apply may be queued behind it
            bt = batches[i % len(batches)]
            k = i % NBUF
            t = C_.c_int()
            rc = B._lib.lib().bmq_match_submit_dev(eng.h, d_tenants.data_ptr(), d_tenant_off.data_ptr(), n_tenants, bt[2].data_ptr(), bt[0].data_ptr(),
                                                   bt[1].data_ptr(), n, d_row[k].data_ptr(), d_ids[k].data_ptr(), d_ids[k].numel(), d_total.data_ptr(), C_.byref(t))
            if rc:
                raise RuntimeError("bmq_match_submit_dev failed: %d" % rc)
            return t.value

        def wait_dev(t):
            rc = B._lib.lib().bmq_match_wait_dev(eng.h, t, None)
            if rc:
                raise RuntimeError("bmq_match_wait_dev failed: %d" % rc)
        try:  # (first: the index is still the one the headline ran on)
            ordered = ordered_batch_leg(eng, w, batches[0][3], n, local_rank, torch, np)
        except Exception as ex:  # noqa: BLE001
            ordered = {"error": repr(ex)}
        out["extra"] = {} if args.ordered_only else extra_legs(args, eng, w, step, torch, np, fetch_csr if not args.no_cpu_baseline else None, (submit_dev, wait_dev))
        out["extra"]["ordered_batch"] = ordered
    eng.close()  # deterministic teardown of everything this script owns, in order, before the line goes out
    if dist is not None:
        dist.destroy_process_group()
    emit_json(out)


def ordered_batch_leg(eng, w, host_batch, n, device, torch, np, steps=5, warm=2):  # noqa: C901
    """The batch as BatchDistRequest carries it: "sorted by tenantId and topic", every topic once (DistWorkerCoProc.proto:75-83,
    BatchDistServerCall.java:138-152).  The headline's batch 0 -- 1 M Zipf publishes, repeats included -- (a) as generated, (b) ordered by
    (tenant, topic) with its repeats, through the same engine; (c) the same rows through an engine with bmq_config.dedup_sorted: equal rows are
    neighbours, the run heads go into a dense batch, the walk runs on that (bmq_dedup_adj_kernels.h); (d) the distinct rows only, the
    caller having done what the reference's batcher does.  Rates are PUBLISHES per second: (c) and (d) answer all n publishes."""
    import bifromq_amd as B
    dev = torch.device("cuda:%d" % device)
    hdata, hoff, htt = host_batch
    t_host = time.perf_counter()
    mv = memoryview(np.ascontiguousarray(hdata))
    rows = [bytes(mv[int(hoff[i]):int(hoff[i + 1])]) for i in range(n)]
    tt_l = htt.tolist()
    order = sorted(range(n), key=lambda i: (tt_l[i], rows[i]))
    srows, stt = [rows[i] for i in order], np.asarray([tt_l[i] for i in order], dtype=np.uint32)
    head = np.ones(n, dtype=bool)
    head[1:] = [stt[i] != stt[i - 1] or srows[i] != srows[i - 1] for i in range(1, n)]
    hidx = np.flatnonzero(head)
    host_s = time.perf_counter() - t_host

    def upload(rs, tts):
        data = np.frombuffer(b"".join(rs) + b"\0" * 64, dtype=np.uint8)
        off = np.zeros(len(rs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(r) for r in rs])
        return (torch.from_numpy(data.copy()).to(dev), torch.from_numpy(off.astype(np.int32)).to(dev), torch.from_numpy(tts.astype(np.int32)).to(dev), len(rs))
    tdata, toff = w.tenants_packed()
    d_tenants, d_tenant_off = torch.from_numpy(tdata.copy()).to(dev), torch.from_numpy(toff.astype(np.int32)).to(dev)
    cap = 24 * n
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)

    def run(engine, bt, k, bufs=None):  # k steps of one shape -> (ms per step, stats of the last one, row pointers, ids)
        nonlocal cap
        d_data, d_off, d_tt, m = bt
        d_row, d_ids = bufs if bufs else (torch.zeros(m + 1, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev))
        ms, st, total = [], None, 0
        for _ in range(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while True:
                engine.match_batch_device(d_tenants.data_ptr(), d_tenant_off.data_ptr(), w.n_tenants, d_tt.data_ptr(), d_data.data_ptr(), d_off.data_ptr(), m,
                                          d_row.data_ptr(), d_ids.data_ptr(), d_ids.numel(), d_total.data_ptr())
                try:
                    total = engine.finish()
                    break
                except B.BmqError as ex:  # the id buffer was too small: the total is known now (a warm-up step's business)
                    if ex.code != -3:
                        raise
                    cap = int(d_total.item()) + int(d_total.item()) // 8 + 4096
                    d_ids = torch.zeros(cap, dtype=torch.int32, device=dev)
            ms.append((time.perf_counter() - t0) * 1e3)
            st = engine.stats()
        return ms, st, d_row, d_ids, total

    b_gen = (torch.from_numpy(np.ascontiguousarray(hdata)).to(dev), torch.from_numpy(hoff.astype(np.int32)).to(dev), torch.from_numpy(htt.astype(np.int32)).to(dev), n)
    b_ord = upload(srows, stt)
    b_dis = upload([srows[i] for i in hidx], stt[hidx])
    eng2 = B.Engine(device=device, kernel_timing=True, dedup_min_topics=1, dedup_sorted=True)
    kb, ko = w.keys_packed()
    eng2.rebuild_raw(kb.ctypes.data, ko.ctypes.data, w.n_keys)
    shapes = (("as_generated", eng, b_gen), ("ordered_with_repeats", eng, b_ord), ("ordered_dedup_sorted", eng2, b_ord), ("ordered_distinct", eng, b_dis))
    # every shape warmed up first (buffers grow, re-runs happen), then `steps` launches of each, shape after shape: in a kernel trace of
    # `bench.py --ordered-only` the LAST 4 x steps dispatches of k_walk are these, in this order (tools/ordered_collect.py)
    bufs = {}
    for name, engine, bt in shapes:
        _, _, d_row, d_ids, _ = run(engine, bt, warm)
        bufs[name] = (d_row, d_ids)
    out = {"workload": "the headline's batch 0 (%d Zipf publishes, %d distinct (tenant, topic) pairs) ordered by (tenant, topic) as BatchDistRequest is" % (n, len(hidx)),
           "host_sort_s": host_s, "steps": steps}
    res = {}
    for name, engine, bt in shapes:
        ms, st, d_row, d_ids, total = run(engine, bt, steps, bufs[name])
        res[name] = (d_row, d_ids[:total])
        # ids per wave of k_expand (64 consecutive rows): an ordered batch puts a hot prefix's rows -- the same big filters' subscribers -- side by side
        rp = d_row.long()
        edge = torch.arange(0, bt[3] + 64, 64, device=dev).clamp(max=bt[3])
        per_wave = (rp[edge[1:]] - rp[edge[:-1]]).float()
        row_len = (rp[1:] - rp[:-1]).float()
        wave_ids = {"mean": float(per_wave.mean()), "p50": float(per_wave.median()), "p99": float(per_wave.quantile(0.99)), "max": float(per_wave.max()),
                    "waves_over_16k_ids": int((per_wave > 16384).sum()), "share_of_ids_in_them": float(per_wave[per_wave > 16384].sum() / max(1.0, float(per_wave.sum()))),
                    "row_max": float(row_len.max())}
        out[name] = {"rows": bt[3], "ms_per_step": float(np.mean(ms)), "publishes_per_s": n / (float(np.mean(ms)) * 1e-3),
                     "kernel_ms": {"k_walk": st.ms_walk, "k_expand (+ k_fill_adj)": st.ms_expand, "dedup kernels": max(0.0, st.ms_total - st.ms_walk - st.ms_expand),
                                   "all_kernels": st.ms_total},
                     "n_walked": int(st.n_walked), "n_visit": int(st.n_visit), "n_match": int(st.n_match), "ids_per_expand_wave": wave_ids,
                     # every shape with its own roofline: the walk's own algorithmic bytes (len + 8 + 32 N_visit per walked row) over its time
                     "roofline": {"bound": "hbm", "kernel": "k_walk", "peak": 8000.0, "unit": "GB/s",
                                  "achieved": (st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit) / (max(st.ms_walk, 1e-9) * 1e-3) / 1e9,
                                  "frac": (st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit) / (max(st.ms_walk, 1e-9) * 1e-3) / 8e12,
                                  "frac_batch": (st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit + 4 * st.n_match) / (max(st.ms_walk, 1e-9) * 1e-3) / 8e12}}
    eng2.close()
    (row_b, ids_b), (row_c, ids_c), (row_d, ids_d) = res["ordered_with_repeats"], res["ordered_dedup_sorted"], res["ordered_distinct"]
    out["ordered_dedup_sorted"]["rows_equal_undeduplicated_engine"] = bool(torch.equal(row_b, row_c) and torch.equal(ids_b, ids_c))
    # the distinct batch's rows are the heads' rows of the ordered batch: same lengths, same ids
    hi = torch.from_numpy(hidx).to(dev)
    len_b = (row_b[1:] - row_b[:-1])[hi]
    same_len = bool(torch.equal(len_b, row_d[1:] - row_d[:-1]))
    same_ids = False
    if same_len:
        starts = row_b[:-1][hi].long()
        pos = torch.repeat_interleave(starts - row_d[:-1].long(), len_b.long()) + torch.arange(ids_d.numel(), device=dev)
        same_ids = bool(torch.equal(ids_b[pos], ids_d))
    out["ordered_distinct"]["rows_equal_heads_of_ordered_batch"] = same_len and same_ids
    return out


def extra_legs(args, eng, w, step, torch, np, fetch_csr=None, tickets=None):
    """Compact legs the default N = 1 run appends, so that the driver's own command exercises them: configs[4] (100 k route mutations
    before every batch) on THIS index, and configs[1] / configs[3] as child runs of this script (3 steps each)."""
    import subprocess

    import bifromq_amd as B
    from bifromq_amd.engine import pack

    extra = {}
    t_all = time.perf_counter()
    try:  # configs[0]: the reference's own CPU-runnable case, engine and restatement side by side
        extra["c1"] = c1_leg(args, torch, np, torch.device("cuda", torch.cuda.current_device()))
    except Exception as ex:  # noqa: BLE001
        extra["c1"] = {"error": repr(ex)}
    for wl in ("c2", "c4"):  # child runs: their own index, their own roofline
        try:
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                                "--no-host-path", "--no-extras"], capture_output=True, text=True, timeout=240)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            keep = ("metric", "value", "unit", "ms_per_step", "kernel_ms", "roofline", "routes_per_topic", "topics_per_filter", "churn", "limited", "compaction")
            extra[wl] = {k: d[k] for k in keep if k in d}
            extra[wl]["workload"] = d["config"]["workload"]
            extra[wl]["wall_s"] = time.perf_counter() - t0
        except Exception as ex:  # noqa: BLE001
            extra[wl] = {"error": repr(ex)}
    # configs[4] on the C3 index of this run: 5 steps, each = bmq_routes_apply(100 k ops from pinned memory) + the 1 M-publish batch
    try:
        # 20 pipelined + 5 blocking steps.  (Round 5's line had 5 + 5 and ONE mean for the pipelined shape: the driver's run read 18.99 ms per
        # step where the builder's read 0.8 and the line could not tell a one-off stall from a steady state.  Now every pipelined step is
        # clocked -- the three calls separately -- and the line carries the list.)
        n_ops, n_pipe_steps, n_block_steps = 100_000, 20, 5
        n_steps = n_pipe_steps + n_block_steps
        rng = np.random.default_rng(4321)
        kb_h, ko_h = w.keys_packed()
        mv = memoryview(kb_h)
        perm = rng.permutation(w.n_keys)[:(n_steps + 1) * (n_ops // 2)]
        tenants_l = w.tenants()
        cb = []
        q = 0
        for b in range(n_steps + 1):
            dels = perm[b * (n_ops // 2):(b + 1) * (n_ops // 2)]
            keys_b = [bytes(mv[int(ko_h[i]):int(ko_h[i + 1])]) for i in dels]
            ops_b = [1] * len(keys_b)
            for _ in range(n_ops - n_ops // 2):
                q += 1
                t = tenants_l[int(rng.integers(0, len(tenants_l)))]
                keys_b.append(B.route_key(t, "churn/l1_%d/+/l3_%d" % (q % 64, q % 4096), 1, "0\0c%d\0d%d" % (q, q % 64)))
                ops_b.append(0)
            order = rng.permutation(len(keys_b))
            data, off = pack([keys_b[i] for i in order])
            cb.append(tuple(torch.from_numpy(x).pin_memory() for x in (data, off, np.array([ops_b[i] for i in order], dtype=np.uint8))))
        lib = B._lib.lib()

        def apply(i):
            data, off, opb = cb[i]
            t0c = time.perf_counter()
            rc = lib.bmq_routes_apply(eng.h, data.data_ptr(), off.data_ptr(), opb.data_ptr(), len(opb))
            if rc:
                raise RuntimeError("bmq_routes_apply failed: %d" % rc)
            return (time.perf_counter() - t0c) * 1e3

        # Two shapes of the same work.  "blocking": bmq_routes_apply, then the batch (rounds 1-4).  "pipelined" (the first n_pipe batches of the
        # mutation stream): the batch is handed over as a ticket, the NEXT mutation batch with bmq_routes_apply_async right behind it -- its
        # upload runs beside the match kernels, its builder kernels behind them, nobody waits in between -- and the ticket is waited for.
        n_pipe = n_pipe_steps if tickets is not None else 0
        apply(0)
        step(0)
        torch.cuda.synchronize()
        pipe = None
        if n_pipe:
            submit_dev, wait_dev = tickets

            def pipe_step(i, clk=None):
                t_a = time.perf_counter()
                t = submit_dev(i)
                t_b = time.perf_counter()
                data, off, opb = cb[i + 1]
                rc = lib.bmq_routes_apply_async(eng.h, data.data_ptr(), off.data_ptr(), opb.data_ptr(), len(opb))
                if rc:
                    raise RuntimeError("bmq_routes_apply_async failed: %d" % rc)
                t_c = time.perf_counter()
                wait_dev(t)
                t_d = time.perf_counter()
                if clk is not None:
                    clk.append(((t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (t_d - t_c) * 1e3))

            for i in range(3):  # (every ticket slot once: their scratch buffers are allocated on first use)
                wait_dev(submit_dev(i))
            # ... and two untimed steps of the pipelined shape itself: the first bmq_routes_apply_async of an engine sets up its upload
            # stream's staging and the page-locked counter block (include/bmq.h says what that first call costs)
            warm_clk = []
            for i in range(2):
                pipe_step(i, warm_clk)
            rc = lib.bmq_routes_apply_wait(eng.h)
            if rc:
                raise RuntimeError("bmq_routes_apply_wait failed: %d" % rc)
            torch.cuda.synchronize()
            clk = []
            t0 = time.perf_counter()
            for i in range(2, n_pipe):
                pipe_step(i, clk)
            rc = lib.bmq_routes_apply_wait(eng.h)
            if rc:
                raise RuntimeError("bmq_routes_apply_wait failed: %d" % rc)
            torch.cuda.synchronize()
            el_p = time.perf_counter() - t0
            n_timed = n_pipe - 2
            tot = [sum(c) for c in clk]
            pipe = {"value": args.topics * n_timed / el_p, "unit": "topics/s", "steps": n_timed, "ms_per_step": el_p / n_timed * 1e3,
                    "ms_per_step_p50": float(np.median(tot)), "ms_per_step_max": float(np.max(tot)),
                    "step_ms": [round(x, 3) for x in tot],
                    "call_ms_mean": {"submit": float(np.mean([c[0] for c in clk])), "apply_async": float(np.mean([c[1] for c in clk])),
                                     "wait": float(np.mean([c[2] for c in clk]))},
                    "warm_step_ms": [[round(x, 3) for x in c] for c in warm_clk],
                    "shape": "ticket(batch i) | bmq_routes_apply_async(mutations i + 1) | wait(ticket): the upload beside the match kernels, the builder kernels "
                             "behind them, one host wait per step; 2 untimed steps of the same shape first"}
        ams, t0 = [], time.perf_counter()
        for i in range(n_pipe, n_steps):
            ams.append(apply(i + 1))
            step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        n_steps_blocking = n_steps - n_pipe
        st = eng.stats()
        alg = st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit + 4 * st.n_match
        extra["c5"] = {"workload": "C5 = C3 + %d route mutations (50 %% unsubscribe / 50 %% subscribe) before every 1 M-publish batch" % n_ops,
                       "value": args.topics * n_steps_blocking / el, "unit": "topics/s", "steps": n_steps_blocking, "ms_per_step": el / n_steps_blocking * 1e3,
                       "pipelined": pipe,
                       "apply_ms_mean": float(np.mean(ams)), "apply_ms_max": float(np.max(ams)),
                       "kernel_ms": {"k_walk": st.ms_walk, "k_expand": st.ms_expand, "all_kernels": st.ms_total},
                       "roofline": {"bound": "hbm", "kernel": "k_walk", "achieved": (alg - 4 * st.n_match) / (st.ms_walk * 1e-3) / 1e9, "peak": 8000.0,
                                    "unit": "GB/s", "frac": (alg - 4 * st.n_match) / (st.ms_walk * 1e-3) / 8e12,
                                    "frac_batch": alg / (st.ms_walk * 1e-3) / 8e12}}
        if fetch_csr is not None:
            extra["c5"]["parity"] = c5_parity(w, cb, n_steps + 1, fetch_csr, np)
    except Exception as ex:  # noqa: BLE001
        extra["c5"] = {"error": repr(ex)}
    try:  # last: the route ids are re-numbered by it
        extra["compaction"] = compaction_leg(args, eng, step, torch, np, fetch_csr)
    except Exception as ex:  # noqa: BLE001
        extra["compaction"] = {"error": repr(ex)}
    extra["wall_s"] = time.perf_counter() - t_all
    return extra


def c1_leg(args, torch, np, dev):
    """configs[0]: 1 tenant, 10 k literal filters (no wildcards), 100 k publishes -- "the reference Java TopicTrie on host CPU" case, the
    small-index / small-batch regime where launch latency, not k_walk, is the story.  The structural restatement of matchAll on the box's
    host cores (one matchAll(singleton) per publish, and ONE matchAll(Set) for the whole batch: a single tenant's call is sequential) beside
    the engine on the same inputs: device-resident batch (kernels + finish) and one blocking host-to-host call; all 100 k rows compared."""
    import bifromq_amd as B
    from bifromq_amd.workload import MODE_LITERAL
    from oracle import oracle as O
    from tests import util as U

    w = B.Workload(0xB1F20001, 1, 10_000, MODE_LITERAL)
    n = 100_000
    data, off, tt = w.topics(0xB1F20001 + 7, n, 0, 1, 900, True)
    tn = w.tenants()
    eng = B.Engine(device=dev.index if dev.index is not None else 0, kernel_timing=True)
    try:
        kb, ko = w.keys_packed()
        eng.rebuild(packed=(kb, ko))
        tdata, toff = w.tenants_packed()
        d_t, d_to = torch.from_numpy(tdata.copy()).to(dev), torch.from_numpy(toff.astype(np.int32)).to(dev)
        d_data, d_off, d_tt = torch.from_numpy(data).to(dev), torch.from_numpy(off.astype(np.int32)).to(dev), torch.from_numpy(tt.astype(np.int32)).to(dev)
        d_row = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        d_ids = torch.zeros(16 * n, dtype=torch.int32, device=dev)
        d_total = torch.zeros(1, dtype=torch.int64, device=dev)

        def step():
            eng.match_batch_device(d_t.data_ptr(), d_to.data_ptr(), 1, d_tt.data_ptr(), d_data.data_ptr(), d_off.data_ptr(), n, d_row.data_ptr(),
                                   d_ids.data_ptr(), d_ids.numel(), d_total.data_ptr())
            return eng.finish()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ms = []
        for _ in range(20):
            t0 = time.perf_counter()
            total = step()
            ms.append((time.perf_counter() - t0) * 1e3)
        st = eng.stats()
        host_ms = []
        for _ in range(5):
            t0 = time.perf_counter()
            row, ids = eng.match_batch(tn, tt, packed_topics=(data, off))
            host_ms.append((time.perf_counter() - t0) * 1e3)
        cores = effective_cpus()
        kv = O.KV(packed=(kb, ko))
        raw = data.tobytes()
        topics = [raw[off[i]:off[i + 1]] for i in range(n)]
        packed = O.pack(topics)
        res, sec = kv.match_singletons(tn, tt, packed, threads=cores)
        t0 = time.perf_counter()
        whole = kv.match_all(tn[0], sorted(set(topics)))
        sec_whole = time.perf_counter() - t0
        parity = {"rows_compared": n}
        try:
            differ = U.assert_csr_equal_modulo_quirk_ii(lambda: w.keys(), kv.key, tn, tt, res.row_ptr.astype(np.int64), res.routes,
                                                        row.astype(np.int64), ids)
            parity["rows_differing_from_reference_restatement"] = int(len(differ))
            parity["differing_rows_equal_semantic_oracle"] = int(U.assert_differing_rows_semantic("bench c1", kv, tn, tt, packed, differ, row.astype(np.int64), ids,
                                                                                                 livelocks=res.livelocks))
        except AssertionError as ex:
            parity = {"FAILED": repr(ex)}
        med = float(np.median(ms))
        alg = st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit + 4 * st.n_match
        return {"workload": "C1: 1 tenant, 10 k literal filters, 100 k publishes (configs[0])",
                "gpu": {"value": n / (med * 1e-3), "unit": "topics/s", "ms_per_batch_p50": med, "ms_per_batch_max": float(np.max(ms)), "steps": len(ms),
                        "kernel_ms": {"k_walk": st.ms_walk, "k_expand": st.ms_expand, "all_kernels": st.ms_total},
                        "host_to_host_call_ms_p50": float(np.median(host_ms)), "routes_per_topic": total / n,
                        "roofline": {"bound": "hbm", "kernel": "k_walk", "achieved": (alg - 4 * st.n_match) / (st.ms_walk * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                     "frac": (alg - 4 * st.n_match) / (st.ms_walk * 1e-3) / 8e12,
                                     "note": "a 100 k-topic launch is 1563 waves on 1024 SIMDs: the chip is not full, latency of the launch and of "
                                             "one wave's dependent chain set the time, not bandwidth"}},
                "cpu_baseline": {"value": n / sec, "unit": "topics/s", "cores": cores, "kind": "port",
                                 "sample": "all %d publishes, one matchAll(singleton(topic)) per publish on %d threads; %.2f s" % (n, cores, sec),
                                 "whole_batch": {"value": n / sec_whole, "unit": "topics/s", "threads": 1, "seconds": sec_whole,
                                                 "note": "ONE matchAll(Set<topic>) with the batch's %d distinct topics: a tenant's call is sequential" % len(set(topics)),
                                                 "reference_livelocks_stepped_over": int(whole.livelocks)}},
                "parity": parity}
    finally:
        eng.close()


def compaction_leg(args, eng, step, torch, np, fetch_csr=None):
    """bmq_compact_begin / _poll / _swap on the index the C5 leg left behind (700 k mutations: abandoned id lists, dead ids, dead trie nodes):
    one thread matches 1 M-publish batches back to back and clocks every one, a second thread builds the next generation in chunks of
    --compact-chunk route ids (its kernels share the engine stream with the batches) -> batch latency idle / while compacting, the
    generation change itself, and the rows of batch 0 before / after it (ids re-numbered: the map old id -> new id must be ONE increasing
    function over every row)."""
    import threading

    from tests import util as U

    stamps = []  # (start of a clocked batch, its duration in ms): where the slowest batch of the compaction fell

    def clocked(k, base=0):
        out = []
        for i in range(k):
            t0 = time.perf_counter()
            step(base + i)
            out.append((time.perf_counter() - t0) * 1e3)
            stamps.append((t0, out[-1]))
        return out

    def pct(v):
        v = np.sort(np.asarray(v))
        return {"n": int(len(v)), "p50": float(v[len(v) // 2]), "p99": float(v[min(len(v) - 1, int(len(v) * 0.99))]),
                "p999": float(v[min(len(v) - 1, int(len(v) * 0.999))]), "max": float(v[-1])}

    clocked(8)
    torch.cuda.synchronize()
    idle = clocked(400)
    before = fetch_csr(0) if fetch_csr is not None else None
    info0 = eng.info()
    t_begin = time.perf_counter()
    eng.compact_begin()
    begin_ms = (time.perf_counter() - t_begin) * 1e3
    poll_ms, poll_t0, failure = [], [], []

    def compactor():
        try:
            done, t_start = 0, time.perf_counter()
            while done < 1000:
                if time.perf_counter() - t_start > 90:
                    raise RuntimeError("the compaction leg is limited to 90 s: at %d permille" % done)
                t0 = time.perf_counter()
                done = eng.compact_poll(args.compact_chunk)
                dt = time.perf_counter() - t0
                poll_ms.append(dt * 1e3)
                poll_t0.append(t0)
                time.sleep(dt * (1.0 - args.compact_duty) / max(args.compact_duty, 1e-3))
        except Exception as ex:  # noqa: BLE001
            failure.append(repr(ex))

    th = threading.Thread(target=compactor)
    t0 = time.perf_counter()
    del stamps[:]
    th.start()
    during, i = [], 0
    while th.is_alive():
        during.extend(clocked(4, i))
        i += 4
    th.join()
    build_s = time.perf_counter() - t0
    if failure:
        eng.compact_abort()
        return {"error": failure[0]}
    # the slowest batch of the compaction: when it ran, and which bmq_compact_poll was in progress then (the three slowest, for the pattern)
    slowest = []
    for ts, ms in sorted(stamps, key=lambda x: -x[1])[:3]:
        k = int(np.searchsorted(np.asarray(poll_t0), ts, side="right")) - 1
        inside = k >= 0 and ts < poll_t0[k] + poll_ms[k] * 1e-3
        slowest.append({"ms": float(ms), "s_after_the_first_poll": float(ts - poll_t0[0]) if poll_t0 else None, "poll_no": k,
                        "inside_that_poll": bool(inside), "that_poll_ms": float(poll_ms[k]) if k >= 0 else None})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    carried, replayed = eng.compact_swap()
    swap_ms = (time.perf_counter() - t0) * 1e3
    info1 = eng.info()
    after_idle = clocked(200)
    out = {"what": "bmq_compact_begin / _poll(%d ids) / _swap while 1 M-publish batches are matched back to back by another thread" % args.compact_chunk,
           "batch_ms_idle": pct(idle), "batch_ms_while_compacting": pct(during), "batch_ms_after_swap": pct(after_idle),
           "p99_ratio": pct(during)["p99"] / pct(idle)["p99"], "poll_ms": pct(poll_ms), "polls": len(poll_ms), "duty": args.compact_duty,
           "slowest_batches_while_compacting": slowest, "begin_ms": begin_ms, "build_s": build_s, "swap_ms": swap_ms, "keys_carried": int(carried), "ops_replayed": int(replayed),
           "before": {"n_routes": int(info0.n_routes), "next_route_id": int(info0.next_route_id), "garbage_bytes": int(info0.garbage_bytes),
                      "device_bytes": int(info0.device_bytes), "generation": int(info0.generation)},
           "after": {"n_routes": int(info1.n_routes), "next_route_id": int(info1.next_route_id), "garbage_bytes": int(info1.garbage_bytes),
                     "device_bytes": int(info1.device_bytes), "generation": int(info1.generation)}}
    if before is not None:
        (_, _, tt), (row0, ids0) = before
        _, (row1, ids1) = fetch_csr(0)
        n = len(tt)
        same_rows = bool((row0[:n + 1] == row1[:n + 1]).all())
        ok = same_rows
        if same_rows:
            rp = row0[:n + 1].astype(np.int64)
            a = U.csr_sorted(rp, ids0[:int(rp[n])].astype(np.int64))
            b = U.csr_sorted(rp, ids1[:int(rp[n])].astype(np.int64))
            pairs = np.unique(np.stack([a, b], axis=1), axis=0)  # sorted by old id, then new id
            ok = bool(len(pairs) == len(np.unique(a)) and (np.diff(pairs[:, 1]) > 0).all())
        out["rows_of_batch_0_equal_across_the_swap"] = {"row_sizes_equal": same_rows, "old_id_to_new_id_is_one_increasing_map": ok, "rows": int(n)}
    return out


def c5_parity(w, cb, n_applied, fetch_csr, np):
    """configs[4]: the rows of batch 0 on the index AFTER the leg's mutation batches against the oracle over the key set as it is then --
    the structural restatement row for row, every differing row against the semantic oracle (tests/util.py), the whole batch."""
    try:
        from oracle import oracle as O
        from tests import util as U
        t0 = time.perf_counter()
        kb, ko = w.keys_packed()
        ko64 = np.asarray(ko, dtype=np.int64)
        raw = kb.tobytes()
        key_id = None
        dels, adds, new_ids = [], [], {}
        next_id = w.n_keys
        for b in range(n_applied):  # the ops in the order they were applied: a put gets the next unused id
            data, off, opb = (x.numpy() for x in cb[b])
            braw = data.tobytes()
            for j, op in enumerate(opb.tolist()):
                k = braw[off[j]:off[j + 1]]
                if op == 0:
                    adds.append(k)
                    new_ids[k] = next_id
                    next_id += 1
                else:
                    dels.append(k)
        # a deleted key -> its old id: the old keys are sorted
        class _Old:
            def __len__(self):
                return w.n_keys

            def __getitem__(self, i):
                return raw[ko64[i]:ko64[i + 1]]
        import bisect
        old = _Old()
        del_ids = [bisect.bisect_left(old, k) for k in dels]
        add_keys = sorted(set(adds))
        kv, old_rank, add_rank, all_keys = U.kv_after_mutations(kb, ko, w.n_keys, del_ids, add_keys)
        new_rank = {new_ids[k]: int(r) for k, r in zip(add_keys, add_rank.tolist())}
        (data, off, tt), (row, ids) = fetch_csr(0)
        n = len(tt)
        res, sec = kv.match_singletons(w.tenants(), tt, (data, off), threads=effective_cpus())
        got = ids[:int(row[n])].astype(np.int64)
        is_old = got < w.n_keys
        mapped = np.where(is_old, old_rank[np.where(is_old, got, 0)], 0)
        if (~is_old).any():
            mapped[~is_old] = [new_rank[int(x)] for x in got[~is_old]]
        if (mapped < 0).any():
            return {"FAILED": "the engine returned the id of a deleted route"}
        got_rp = row[:n + 1].astype(np.int64)
        mapped = U.csr_sorted(got_rp, mapped)
        differ = U.assert_csr_equal_modulo_quirk_ii(all_keys, kv.key, w.tenants(), tt, res.row_ptr.astype(np.int64), res.routes.astype(np.int64), got_rp, mapped)
        n_sem = U.assert_differing_rows_semantic("bench c5 (batch 0 after %d mutation batches)" % n_applied, kv, w.tenants(), tt, (data, off), differ, got_rp, mapped,
                                                 livelocks=res.livelocks)
        return {"rows_compared": int(n), "rows_differing_from_reference_restatement": int(len(differ)), "differing_rows_equal_semantic_oracle": int(n_sem),
                "reference_livelocks": int(res.livelocks), "route_keys_after_the_mutations": int(len(kv)), "seconds": round(time.perf_counter() - t0, 1)}
    except AssertionError as ex:
        return {"FAILED": repr(ex)}


def node_batch(args, rank, world, local_rank, dev, dist, total_tenants, per_tenant, mode, seed):
    """The node-wide shape of configs[2]: ONE Zipf batch of --topics publishes over all tenants arrives (resident on every GPU),
    each rank picks its part on the device (bifromq_amd/shard.py::partition_batch), matches it, and one all-reduce of per-topic
    fan-out counts gives every rank the node-wide answer.  Tenants are owned by hash(tenantId) mod N; tenants whose share of the
    publishes exceeds half a rank's fair share are SPLIT BY FILTER: their route keys are spread over all ranks by hash(route key)
    and their publishes go to every rank (fan-outs add up).  Strong scaling: the batch is fixed, a rank matches ~1/N of it."""
    import numpy as np
    import torch

    import bifromq_amd as B
    from bifromq_amd import shard

    n = args.topics
    # A rank that fails while it sets up must not leave the others waiting in a collective: everybody reports in first.
    setup_error = None
    try:
        state = _node_batch_setup(args, rank, world, local_rank, dev, total_tenants, per_tenant, mode, seed, n)
    except Exception as ex:  # noqa: BLE001
        setup_error, state = repr(ex), None
    flag = torch.tensor([0 if setup_error is None else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        if state is not None:
            state[0].close()
        return {"skipped": "set-up failed on at least one rank" + ("" if setup_error is None else ": " + setup_error)}
    eng, tn, tt, hot, n_split_keys, d_tenants, d_tenant_off, d_owner, d_data, d_off, d_tt = state
    cap = 24 * n
    d_row = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    d_ids = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    m_sel = 0
    return _node_batch_run(args, rank, world, dev, dist, n, eng, tn, tt, hot, n_split_keys, d_tenants, d_tenant_off, d_owner, d_data, d_off, d_tt,
                           d_row, d_ids, d_total)


def _node_batch_setup(args, rank, world, local_rank, dev, total_tenants, per_tenant, mode, seed, n):
    import numpy as np
    import torch

    import bifromq_amd as B
    from bifromq_amd import shard

    full = B.Workload(seed, total_tenants, per_tenant, mode)
    tn = full.tenants()
    data, off, tt = full.topics(seed + 77, n, grouped=not args.ungrouped)  # same batch on every rank
    share = np.bincount(tt, minlength=len(tn)) / float(n)
    hot_by_share = shard.pick_hot_tenants(share, world, 0.5)
    if args.split_policy == "fanout-hinter":
        # the split the reference's own rule would make: FanoutSplitHinter over the mutation stream -- here the bulk load, one batch of
        # per-tenant route counts -- with the reference's threshold; every rank computes the same decision from the same counts
        hinter = shard.FanoutSplitHinter(world, args.split_threshold)
        counts = np.diff(np.asarray(full.tenant_first(), dtype=np.int64))
        to_split, _ = hinter.record_counts({tn[t]: int(counts[t]) for t in range(total_tenants)})
        index_of = {name: t for t, name in enumerate(tn)}
        hot = sorted(index_of[name] for name in to_split)
    else:
        hot = hot_by_share
    owner_np = shard.topic_targets(tn, hot, world)
    # this rank's index: its tenants (bulk load of the sorted keys) + its hash share of the split tenants' keys (apply)
    mine = [t for t in range(total_tenants) if owner_np[t] == rank]
    eng = B.Engine(device=local_rank)
    wm = B.Workload(seed, len(mine), per_tenant, mode, tenant_ids=mine)
    kb, ko = wm.keys_packed()
    eng.rebuild_raw(kb.ctypes.data, ko.ctypes.data, wm.n_keys)
    n_split_keys = 0
    if hot:
        wh = B.Workload(seed, len(hot), per_tenant, mode, tenant_ids=hot)
        my_hot = [k for k in wh.keys() if shard.key_rank(k, world) == rank]
        n_split_keys = len(my_hot)
        for i in range(0, len(my_hot), 50000):
            eng.apply([(0, k) for k in my_hot[i:i + 50000]])
    tdata, toff = full.tenants_packed()
    d_tenants = torch.from_numpy(tdata.copy()).to(dev)
    d_tenant_off = torch.from_numpy(toff.astype(np.int32)).to(dev)
    d_owner = torch.from_numpy(owner_np).to(dev)
    d_data = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(off.astype(np.int32)).to(dev)
    d_tt = torch.from_numpy(tt.astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    args._split_info = {"policy": args.split_policy, "threshold": args.split_threshold if args.split_policy == "fanout-hinter" else None,
                        "tenants_the_publish_share_rule_would_split": [tn[h] for h in hot_by_share]}
    return eng, tn, tt, hot, n_split_keys, d_tenants, d_tenant_off, d_owner, d_data, d_off, d_tt


def _node_batch_run(args, rank, world, dev, dist, n, eng, tn, tt, hot, n_split_keys, d_tenants, d_tenant_off, d_owner, d_data, d_off, d_tt,
                    d_row, d_ids, d_total):
    import numpy as np
    import torch

    import bifromq_amd as B
    from bifromq_amd import shard

    m_sel = 0
    part = shard.DevicePartition(eng, d_owner, n, int(d_data.numel()), dev)  # bmq_partition_batch_dev: kernels on the engine stream

    step_error = [None]

    def step():
        # a rank that fails here must not leave the others waiting in the all-reduce: it contributes nothing and reports afterwards
        try:
            return step_body()
        except Exception as ex:  # noqa: BLE001
            step_error[0] = repr(ex)
            return shard.exchange_fanout(dist, torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int64, device=dev), n)

    def step_body():
        nonlocal m_sel, d_ids
        sel, pd, po, ptt, m = part(d_tt, d_data, d_off, rank)
        m_sel = m
        if m:
            while True:
                eng.match_batch_device(d_tenants.data_ptr(), d_tenant_off.data_ptr(), len(tn), ptt.data_ptr(), pd.data_ptr(), po.data_ptr(), m,
                                       d_row.data_ptr(), d_ids.data_ptr(), d_ids.numel(), d_total.data_ptr())
                try:
                    eng.finish()
                    break
                except B.BmqError as ex:
                    if ex.code != -3:
                        raise
                    d_ids = torch.zeros(int(d_total.item()) * 2, dtype=torch.int32, device=dev)
            counts = d_row[1:m + 1] - d_row[:m]
        else:
            counts = torch.zeros(0, dtype=torch.int32, device=dev)
        return shard.exchange_fanout(dist, counts, sel, n)

    fan = step()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.node_batch_steps):
        fan = step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed, float(m_sel)], dtype=torch.float64, device=dev)
    sel_all = torch.zeros(2 * world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(sel_all, tmax)
    per_rank = sel_all.view(world, 2)[:, 1].cpu().numpy()
    elapsed = float(sel_all.view(world, 2)[:, 0].max().item())
    flag = torch.tensor([0 if step_error[0] is None else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        eng.close()
        return {"skipped": "a step failed on at least one rank" + ("" if step_error[0] is None else ": " + step_error[0])}
    # what the imbalance would be WITHOUT the split (every publish to its tenant's owner)
    plain_owner = np.array([shard.tenant_rank(t, world) for t in tn])
    plain = np.bincount(plain_owner[tt], minlength=world)
    eng.close()
    return {"value": n * args.node_batch_steps / elapsed, "unit": "topics/s", "scaling": "strong", "steps": args.node_batch_steps,
            "ms_per_step": elapsed / args.node_batch_steps * 1e3, "publishes_per_batch_node": n, "fanout_total": int(fan.sum().item()),
            "split_tenants": [tn[h] for h in hot], "split_route_keys_this_rank": n_split_keys,
            "publishes_per_rank": [int(x) for x in per_rank], "imbalance_max_over_mean": float(per_rank.max() / per_rank.mean()),
            "imbalance_without_split": float(plain.max() / plain.mean()), "split": getattr(args, "_split_info", None),
            "note": "one shared batch: device-side partition (bmq_partition_batch_dev) by hash(tenantId) mod N (hot tenants split by filter, publishes to all ranks) "
                    "-> match -> all-reduce of per-topic fan-out; includes the partition and the exchange"}


def host_visible(args, eng, w, batches, n, seed, rank, cap):
    """SURVEY 8d's latency definition: host enqueue -> results visible on host.  Inputs and outputs live in page-locked host
    memory (bmq_host_alloc: what a JNI binding hands over as direct buffers).
      * p99 / p50 latency of ONE blocking bmq_match_batch call (upload + kernels + download, nothing overlapped);
      * throughput with three batches in flight (bmq_match_submit / bmq_match_wait): the upload of batch i+1 and the download of
        batch i-1 overlap the kernels of batch i."""
    import numpy as np

    import bifromq_amd as B
    from bifromq_amd.engine import pinned, _ptr
    import ctypes as C

    tdata, toff = w.tenants_packed()
    p_t = pinned(len(tdata), np.uint8)
    p_t[:] = tdata
    p_to = pinned(len(toff), np.uint32)
    p_to[:] = toff
    hb = []
    for b in range(2):
        data, off, tt = w.topics(seed + 1000 * (rank + 1) + b, n, grouped=not args.ungrouped)
        pd, po, pt = pinned(len(data), np.uint8), pinned(len(off), np.uint32), pinned(len(tt), np.uint32)
        pd[:], po[:], pt[:] = data, off, tt
        hb.append((pd, po, pt))
    D = 3  # BMQ_MAX_TICKETS: one batch on its way up, one on the GPU, one on its way down
    rows = [pinned(n + 1, np.uint32) for _ in range(D)]
    ids = [pinned(cap, np.uint32) for _ in range(D)]
    L = B._lib.lib()
    need = C.c_uint64()
    lat = []
    for i in range(args.steps + 3):
        pd, po, pt = hb[i % 2]
        t0 = time.perf_counter()
        rc = L.bmq_match_batch(eng.h, _ptr(p_t), _ptr(p_to), w.n_tenants, _ptr(pt), _ptr(pd), _ptr(po), n, _ptr(rows[0]), _ptr(ids[0]), cap,
                               C.byref(need))
        if rc:
            raise RuntimeError("bmq_match_batch: %d" % rc)
        if i >= 3:
            lat.append((time.perf_counter() - t0) * 1e3)
    total_ids = need.value
    # pipelined
    K = args.steps
    tickets = [None] * D

    def pipelined(k):
        for i in range(k + D - 1):
            if i < k:
                pd, po, pt = hb[i % 2]
                tickets[i % D] = eng.match_submit(p_t, p_to, w.n_tenants, pt, pd, po, n)
            if i >= D - 1:
                j = (i - (D - 1)) % D
                got = eng.match_wait(tickets[j], rows[j], ids[j])
                assert got == total_ids or len(hb) > 1

    pipelined(D)  # untimed: every ticket's slot allocates its buffers on first use
    t0 = time.perf_counter()
    pipelined(K)
    sec = time.perf_counter() - t0
    in_bytes = sum(int(x.nbytes) for x in hb[0])
    # ---- the formats that fit the wire (include/bmq.h BMQ_FMT_*): the same two measurements per format
    formats = {}
    try:
        rptr = [pinned(n + 1, np.uint32) for _ in range(D)]
        ranges = [pinned((max(total_ids // 2, 1024), 2), np.uint32) for _ in range(D)]
        side = [pinned(1 << 20, np.uint32) for _ in range(D)]
        goff, grep_ = pinned(4097, np.uint32), pinned(4096, np.uint32)
        state = {}

        def wait(fmt, ticket, j):
            if fmt == eng.FMT_COUNTS:
                eng.match_wait_counts(ticket, rows[j])
                return 4 * (n + 1)
            if fmt == eng.FMT_RANGES:
                info = eng.match_wait_ranges(ticket, rptr[j], ranges[j], side[j], rows[j])
                state["ranges"] = int(info.n_ranges)
                return 8 * (n + 1) + 8 * int(info.n_ranges) + 4 * int(info.n_side_ids)
            tot, ng, _ = eng.match_wait_grouped(ticket, ids[0], ids[1], goff, grep_)  # (one pair of big buffers: no second batch in flight)
            return 8 * int(tot) + 8 * int(ng)

        for name, fmt in (("counts", eng.FMT_COUNTS), ("ranges", eng.FMT_RANGES), ("grouped", eng.FMT_GROUPED)):
            lat_f, out_bytes = [], 0
            for i in range(args.steps + 2):
                pd, po, pt = hb[i % 2]
                t1 = time.perf_counter()
                out_bytes = wait(fmt, eng.match_submit_fmt(p_t, p_to, w.n_tenants, pt, pd, po, n, fmt), 0)
                if i >= 2:
                    lat_f.append((time.perf_counter() - t1) * 1e3)
            depth = 1 if fmt == eng.FMT_GROUPED else D
            tk = [None] * D

            def run(k):
                for i in range(k + depth - 1):
                    if i < k:
                        pd, po, pt = hb[i % 2]
                        if depth == 1:
                            wait(fmt, eng.match_submit_fmt(p_t, p_to, w.n_tenants, pt, pd, po, n, fmt), 0)
                            continue
                        tk[i % D] = eng.match_submit_fmt(p_t, p_to, w.n_tenants, pt, pd, po, n, fmt)
                    if i >= depth - 1:
                        wait(fmt, tk[(i - (depth - 1)) % D], (i - (depth - 1)) % D)

            run(D)
            t1 = time.perf_counter()
            run(K)
            sec_f = time.perf_counter() - t1
            formats[name] = {"value_host_visible": n * K / sec_f, "ms_per_batch_pipelined": sec_f / K * 1e3,
                             "p50_host_visible_ms": float(np.percentile(lat_f, 50)), "p99_host_visible_ms": float(np.percentile(lat_f, 99)),
                             "bytes_out_per_batch": int(out_bytes), "in_flight": depth}
            if name == "ranges":
                formats[name]["ranges_per_batch"] = state.get("ranges")
    except Exception as ex:  # never fails the bench line
        formats["error"] = repr(ex)
    return {"value_host_visible": n * K / sec, "unit": "topics/s", "ms_per_batch_pipelined": sec / K * 1e3,
            "formats": formats,
            "p50_host_visible_ms": float(np.percentile(lat, 50)), "p99_host_visible_ms": float(np.percentile(lat, 99)),
            "bytes_in_per_batch": in_bytes, "bytes_out_per_batch": int(4 * (n + 1) + 4 * total_ids),
            "note": "host buffers in, CSR out, page-locked memory; latency = one blocking bmq_match_batch (upload + kernels + download); "
                    "throughput = bmq_match_submit/bmq_match_wait with three batches in flight (one caller thread); formats: the same with bmq_match_submit_fmt -- "
                    "counts = row pointers only (all BatchDistReply carries), ranges = matched (begin, count) id ranges the consumer expands, "
                    "grouped = (topic, route) pairs by DelivererKey; every rate is bounded by the %.1f MB of topics that go IN over PCIe" % (in_bytes / 1e6)}


def fanout_group_leg(eng, d_row, d_ids, n, dev, torch, np, reps=10):
    """SURVEY.md 8f-4, the step behind the match: bmq_fanout_group_dev (segmented sort of the CSR by DelivererKey) on the CSR the last
    batch left in HBM.  Outside the timed region; never fails the bench line."""
    import bifromq_amd as B
    try:
        total = int(d_row[n].item())
        if total == 0 or total > (1 << 28):
            return {"skipped": "%d pairs" % total}
        gcap = 1 << 16
        ot, orr = torch.zeros(total, dtype=torch.int32, device=dev), torch.zeros(total, dtype=torch.int32, device=dev)
        goff, grep = torch.zeros(gcap + 1, dtype=torch.int32, device=dev), torch.zeros(gcap, dtype=torch.int32, device=dev)
        a = (d_row.data_ptr(), d_ids.data_ptr(), n, total, ot.data_ptr(), orr.data_ptr(), goff.data_ptr(), grep.data_ptr(), gcap)
        t0 = time.perf_counter()
        ng, sp = eng.fanout_group_device(*a)  # first call: parses the key tail of every route the batch touches
        first = (time.perf_counter() - t0) * 1e3
        ms = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ng, sp = eng.fanout_group_device(*a)
            ms.append((time.perf_counter() - t0) * 1e3)
        sizes = np.diff(goff[:ng + 1].cpu().numpy().astype(np.int64))
        return {"pairs": total, "groups": int(ng), "special": int(sp), "ms_first_call": first, "ms": float(np.median(ms)),
                "pairs_per_s": total / (float(np.median(ms)) * 1e-3), "largest_group_pairs": int(sizes.max()) if ng else 0,
                "note": "bmq_fanout_group_dev on the device-resident CSR of one batch: (topic, route) pairs regrouped by "
                        "DelivererKey(subBrokerId, delivererKey) as DeliverExecutorGroup.submit + the deliverer's batcher do; wall time of "
                        "the C-ABI call (it returns when the group table is complete)"}
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}


def attach_traffic(out, workload, world):
    """roofline.traffic = HBM bytes per launch of the workload's dominant kernel from the committed rocprofv3 --pmc passes
    (tools/profile_round.sh + tools/collect_profiles.py).  The file records the hash of the kernel sources it was measured with: a stale
    measurement is not reported."""
    out["roofline"]["traffic_source"] = None
    for tf in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("traffic_r") and f.endswith(".json")), reverse=True):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
        except Exception:
            continue
        if world > 1:  # the counter passes ran the single-GPU configuration (a rank of N holds 1/N of the index)
            out["roofline"]["traffic_source"] = "profiles/%s was measured at n_gpus = 1: not reported for a shard" % tf
        elif tj.get("kernel_sources_sha") != kernel_sources_sha() and not kernel_code_unchanged(tj, tj.get(workload + "_kernel")):
            out["roofline"]["traffic_source"] = "profiles/%s is stale (the sources changed since, and so did the code of the measured kernel): not reported" % tf
        elif tj.get(workload + "_kernel") not in (None, out["roofline"].get("kernel")):
            out["roofline"]["traffic_source"] = "profiles/%s holds the traffic of %s, this run's dominant kernel is %s: not reported" % (
                tf, tj.get(workload + "_kernel"), out["roofline"].get("kernel"))
        else:
            out["roofline"]["traffic"] = tj.get(workload)
            out["roofline"]["traffic_source"] = "profiles/" + tf
            if tj.get("kernel_sources_sha") != kernel_sources_sha():
                out["roofline"]["traffic_source"] += (" (sources under bifromq_amd/csrc/ changed since it was measured; the machine code of %s in the library "
                                                      "is byte for byte the measured one: tools/kernel_isa.py)" % tj.get(workload + "_kernel"))
        break


def kernel_code_unchanged(tj, kernel):
    """the measured kernel's machine code (code bytes + kernel descriptor, tools/kernel_isa.py) in the library this run loads is the one the traffic
    file recorded: a change elsewhere in the sources does not retire a measurement of a kernel it did not touch"""
    want = (tj.get("kernel_isa_sha") or {}).get(kernel)
    if not want:
        return False
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_isa
        return kernel_isa.kernel_hashes().get(kernel) == want
    except Exception:  # noqa: BLE001
        return False
    finally:
        sys.path.pop(0)


def kernel_sources_sha():
    """hash of the sources the match kernels are built from: ties a PMC traffic measurement to the code it was taken with"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "bifromq_amd", "csrc")
    for f in sorted(os.listdir(csrc)):  # everything bmq_engine.o is built from: kernels AND the launch geometry in bmq_engine.hip
        if f.endswith((".h", ".inc", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def emit_json(out):
    """ONE JSON line, and the last thing on stdout: flush the C runtime's buffer first (RCCL prints its version banner
    through it) so that nothing lands after the line."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    # The line is out and every engine object has been closed explicitly.  What is left is the interpreter's shutdown of the HIP / RCCL /
    # OpenMP runtimes the process loaded (PyTorch's, the library's, the oracle's), whose at-exit handlers race each other once in a while
    # (seen once in round 3 as a glibc "double free" AFTER the line, rc != 0): leave without running them.
    sys.stderr.flush()
    profiled = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if not profiled:  # (a profiler's tool library writes its output in an exit handler: under rocprofv3 the normal way out is taken)
        os._exit(0)


def bench_retain(args, rank, world, local_rank, dev, dist):
    """configs[3]: retain-store direction -- 1M retained topics (1 tenant), batches of 100k wildcard SUBSCRIBE filters.
    Every rank holds a replica (the config is single-tenant); unit = one filter fully resolved to its topic-id list."""
    import numpy as np
    import torch

    import bifromq_amd as B

    seed = 0xB1F20004
    w = B.Workload(seed, 1, 1, 0)
    n_topics, n = 1_000_000, min(args.topics, 100_000) if args.topics != 1_000_000 else 100_000
    data, off, tt = w.retain(seed, n_topics, filters=False)
    eng = B.Engine(device=local_rank, kernel_timing=True)
    # every retained topic carries (timestamp, expiry) as IRetainTopicIndex.add hands them over (RS/RetainStoreCoProc.java:240-255): the
    # complete-set legs ignore them, the match(limit, now) leg below picks live topics with `now` in the middle of the expiry instants
    rng_t = np.random.default_rng(0xB1F2)
    base_ms = 1_700_000_000_000
    ts_hlc = ((base_ms + rng_t.integers(0, 100_000, n_topics)).astype(np.uint64) << np.uint64(16))
    expiry_s = rng_t.choice(np.array([30, 60, 3600, 0x7FFFFFFF], dtype=np.uint32), n_topics)
    t0 = time.time()
    eng.retain_rebuild(w.tenants(), tt, packed_topics=(data, off), timestamps=ts_hlc, expiry=expiry_s)
    t_build = time.time() - t0
    tdata, toff = w.tenants_packed()
    d_tenants = torch.from_numpy(tdata.copy()).to(dev)
    d_tenant_off = torch.from_numpy(toff.astype(np.int32)).to(dev)
    batches = []
    for b in range(args.batches):
        fdata, foff, ft = w.retain(seed + 1 + b + 100 * rank, n, filters=True)
        batches.append((torch.from_numpy(fdata).to(dev), torch.from_numpy(foff.astype(np.int32)).to(dev),
                        torch.from_numpy(ft.astype(np.int32)).to(dev), (fdata, foff, ft) if b == 0 else None))
    cap = 64 * n
    d_row = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    d_ids = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def step(i):
        nonlocal d_ids, cap
        bt = batches[i % len(batches)]
        while True:
            eng.retain_match_batch_device(d_tenants.data_ptr(), d_tenant_off.data_ptr(), 1, bt[2].data_ptr(), bt[0].data_ptr(),
                                          bt[1].data_ptr(), n, d_row.data_ptr(), d_ids.data_ptr(), cap, d_total.data_ptr())
            try:
                return eng.finish()
            except B.BmqError as ex:
                if ex.code != -3:
                    raise
                cap = int(d_total.item()) * 2
                d_ids = torch.zeros(cap, dtype=torch.int32, device=dev)

    import gc  # (as in main(): no cyclic collection inside a timed region)
    gc.collect()
    gc.freeze()
    gc.disable()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    lat, walk_ms, expand_ms, alg = [], [], [], []
    n_match = n_visit = 0
    t_start = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        step(i)
        lat.append((time.perf_counter() - ts) * 1e3)
        st = eng.stats()
        walk_ms.append(st.ms_walk)
        expand_ms.append(st.ms_expand)
        alg.append(st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit + 4 * st.n_match)
        n_match += st.n_match
        n_visit += st.n_visit
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if rank != 0:
            dist.destroy_process_group()
            return
    kw, ke = float(np.mean(walk_ms)), float(np.mean(expand_ms))
    dom_name, k_ms = ("k_retain_walk", kw) if kw >= ke else ("k_expand", ke)
    # ONE dominant kernel and ITS share of the algorithmic bytes: the walk reads the filters and touches the nodes
    # (len + 8 + 32 N_visit), the expansion writes the ids (4 N_match)
    walk_bytes = float(np.mean(alg)) - 4.0 * n_match / args.steps
    exp_bytes = 4.0 * n_match / args.steps
    achieved = (walk_bytes if dom_name == "k_retain_walk" else exp_bytes) / (k_ms * 1e-3) / 1e9
    limited = None
    if world == 1:  # (before the churn leg: the index as loaded)
        limited = {"clean": retain_limited_leg(eng, w, batches[0][3], base_ms + 95_000, np)}
    churn = retain_churn_leg(eng, w, data, off, n_topics, step, torch, np) if world == 1 and not args.no_churn else None
    if limited is not None and churn is not None:
        limited["churned"] = retain_limited_leg(eng, w, batches[0][3], base_ms + 95_000, np)
    compaction = None
    if churn is not None:  # the churned index (50 k dead ids, 50 k overlay topics) folded into a fresh bulk load, beside the matcher and with the stall
        try:
            compaction = retain_compaction_leg(eng, step, torch, np)
        except Exception as ex:  # noqa: BLE001
            compaction = {"error": repr(ex)}
    out = {"metric": "retain-direction filter matches/sec (whole node)", "value": world * n * args.steps / elapsed,
           "unit": "filters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u32", "data": "synthetic",
           "config": {"workload": "C4: 1M retained topics (1 tenant), %d wildcard filters per batch "
                                  "(50%% one '+', 30%% trailing '#', 20%% both)" % n,
                      "parallelism": "replica per GPU (single-tenant config does not shard)"},
           "p99_batch_ms": float(np.percentile(lat, 99)), "topics_per_filter": n_match / (n * args.steps),
           "nodes_touched_per_filter": n_visit / (n * args.steps),
           "kernel_ms": {"k_retain_walk": float(np.mean(walk_ms)), "k_expand": float(np.mean(expand_ms))},
           "host_s": {"rebuild": t_build},
           "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": 8000.0,
                        "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                        "own_algorithmic_bytes_per_launch": walk_bytes if dom_name == "k_retain_walk" else exp_bytes,
                        "own_bytes_rule": "k_retain_walk: len(filter) + 8 + 32 * N_visit per filter; k_expand: 4 * N_match",
                        "frac_batch": float(np.mean(alg)) / (k_ms * 1e-3) / 8e12,
                        "frac_pipeline": float(np.mean(alg)) / ((kw + ke) * 1e-3) / 8e12,
                        "algorithmic_bytes_per_launch": float(np.mean(alg)),
                        "kernel_own_frac": {"k_retain_walk": walk_bytes / (max(kw, 1e-9) * 1e-3) / 8e12, "k_expand": exp_bytes / (max(ke, 1e-9) * 1e-3) / 8e12}},
           "churn": churn, "limited": limited, "compaction": compaction}
    attach_traffic(out, "c4", world)
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle as O
        lt = O.LevelTrie(1)
        raw = data.tobytes()
        tn = w.tenants()
        for i in range(n_topics):
            lt.add(tn[0], raw[off[i]:off[i + 1]], i)
        fdata, foff, ft = batches[0][3]
        m = min(n, 20000)
        cores = effective_cpus()
        res, sec = lt.match_batch(tn, ft[:m], (fdata, foff[:m + 1]), threads=cores)
        out["cpu_baseline"] = {"value": m / sec, "unit": "filters/s", "cores": cores, "logical_cpus_visible": os.cpu_count(), "kind": "port",
                               "sample": "first %d filters of batch 0 against the full 1M-topic TopicLevelTrie restatement on "
                                         "%d threads; %.1f s" % (m, cores, sec)}
    if world > 1:
        dist.destroy_process_group()
    emit_json(out)


def retain_compaction_leg(eng, step, torch, np):
    """bmq_retain_compact_begin / _build / _swap on the churned C4 index while the main thread matches 100 k-filter batches back to back
    (TopicLevelTrie contracts online, UTIL/index/TopicLevelTrie.java:257-384), then bmq_retain_compact -- the same generation change under the
    engine lock -- for the length of the stall it replaces."""
    import threading

    def pct(v):
        v = np.sort(np.asarray(v))
        return {"n": int(len(v)), "p50": float(v[len(v) // 2]), "p99": float(v[min(len(v) - 1, int(len(v) * 0.99))]), "max": float(v[-1])}

    def clocked(k):
        out = []
        for i in range(k):
            t0 = time.perf_counter()
            step(i)
            out.append((time.perf_counter() - t0) * 1e3)
        return out

    clocked(3)
    idle = clocked(30)
    info0 = eng.retain_info()
    t0 = time.perf_counter()
    eng.retain_compact_begin()
    begin_ms = (time.perf_counter() - t0) * 1e3
    fail, build_s = [], [0.0]

    def builder():
        try:
            t1 = time.perf_counter()
            eng.retain_compact_build()
            build_s[0] = time.perf_counter() - t1
        except Exception as ex:  # noqa: BLE001
            fail.append(repr(ex))

    th = threading.Thread(target=builder)
    th.start()
    during = []
    while th.is_alive():
        during.extend(clocked(2))
    th.join()
    if fail:
        eng.retain_compact_abort()
        return {"error": fail[0]}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    carried, replayed = eng.retain_compact_swap()
    swap_ms = (time.perf_counter() - t0) * 1e3
    info1 = eng.retain_info()
    after = clocked(10)
    st = eng.stats()
    t0 = time.perf_counter()
    eng.retain_compact()  # the stall the three calls replace (on the index they left: same size)
    blocking_s = time.perf_counter() - t0
    return {"what": "bmq_retain_compact_begin / _build / _swap while 100 k-filter batches are matched back to back by the calling thread",
            "begin_ms": begin_ms, "build_s": build_s[0], "swap_ms": swap_ms, "topics_carried": int(carried), "ops_replayed": int(replayed),
            "batch_ms_idle": pct(idle), "batch_ms_while_building": pct(during), "batch_ms_after_swap": pct(after),
            "kernel_ms_after_swap": {"k_retain_walk": st.ms_walk, "k_expand": st.ms_expand},
            "before": {"retained": int(info0.n_topics), "loaded_removed": int(info0.loaded_removed), "added_ids": int(info0.added_ids)},
            "after": {"retained": int(info1.n_topics), "loaded_removed": int(info1.loaded_removed), "added_ids": int(info1.added_ids), "generation": int(info1.generation)},
            "bmq_retain_compact_blocking_s": blocking_s}


def retain_limited_leg(eng, w, batch0, now_ms, np, limit=10, reps=5):
    """RetainStoreCoProc.match(tenant, filter, limit, now) as production calls it (RS/RetainStoreCoProc.java:167-190, limit =
    RetainMessageMatchLimit = 10, Setting.java:77): bmq_retain_match_limited for the 100 k filters of batch 0, host buffers in, the <= 10 live
    ids per filter out -- nothing is expanded: k_retain_walk (+ k_retain_overlay on a churned index) and k_limit_select over the matched id
    ranges.  filters/s is the wall time of the C-ABI call (upload + kernels + download); the roofline is the walk's own bytes over its time."""
    import ctypes as C

    import bifromq_amd as B
    from bifromq_amd.engine import pinned

    fdata, foff, ft = batch0
    n = len(foff) - 1
    lib = B._lib.lib()
    tdata, toff = w.tenants_packed()
    bufs = {}
    for name, src, dt in (("t", tdata, np.uint8), ("to", toff, np.uint32), ("ft", ft, np.uint32), ("f", fdata, np.uint8), ("fo", foff, np.uint32),
                          ("lim", np.full(n, limit, dtype=np.uint32), np.uint32)):
        bufs[name] = pinned(len(src) + 16, dt)
        bufs[name][:len(src)] = src
    row, ids, cnt = pinned(n + 1, np.uint32), pinned(n * limit + 16, np.uint32), pinned(n, np.uint32)
    need = C.c_uint64()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    def call():
        rc = lib.bmq_retain_match_limited(eng.h, ptr(bufs["t"]), ptr(bufs["to"]), 1, ptr(bufs["ft"]), ptr(bufs["f"]), ptr(bufs["fo"]), n, ptr(bufs["lim"]),
                                          now_ms, ptr(row), ptr(ids), n * limit + 16, C.byref(need), ptr(cnt))
        if rc:
            raise RuntimeError("bmq_retain_match_limited failed: %d %s" % (rc, lib.bmq_last_error(eng.h)))

    call()
    ms, kw, kl = [], [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        ms.append((time.perf_counter() - t0) * 1e3)
        st = eng.stats()
        kw.append(st.ms_walk)
        kl.append(st.ms_expand)
    st = eng.stats()
    walk_bytes = st.topic_bytes + 8 * st.n_topics + 32 * st.n_visit
    med = float(np.median(ms))
    return {"what": "bmq_retain_match_limited: %d filters, limit %d, now in the middle of the expiry instants; host buffers in, ids out" % (n, limit),
            "value": n / (med * 1e-3), "unit": "filters/s", "call_ms_median": med, "call_ms": [round(x, 3) for x in ms],
            "kernel_ms": {"k_retain_walk (+ k_retain_overlay beside it)": float(np.mean(kw)), "k_limit_select + rowptr + compact": float(np.mean(kl))},
            "ids_returned": int(need.value), "matches_counted": int(cnt[:n].astype(np.int64).sum()),
            "roofline": {"bound": "hbm", "kernel": "k_retain_walk", "achieved": walk_bytes / (float(np.mean(kw)) * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": walk_bytes / (float(np.mean(kw)) * 1e-3) / 8e12, "own_algorithmic_bytes_per_launch": walk_bytes,
                         "own_bytes_rule": "len(filter) + 8 + 32 * N_visit per filter (nothing is expanded)"}}


def retain_churn_leg(eng, w, data, off, n_topics, step, torch, np, n_ops=100_000, reps=6):
    """SURVEY row a10 at configs[3] size: batches of 100 k IRetainTopicIndex.add / remove (half removes of retained topics, half adds of
    new ones; every other batch undoes the one before it) through bmq_retain_apply_batch from page-locked buffers -- three kernels on the
    engine stream -- and the 100 k-filter batch on the churned index (dead-aware expansion + overlay walk)."""
    import ctypes as C

    import bifromq_amd as B
    from bifromq_amd.engine import pack

    rng = np.random.default_rng(99)
    raw = data.tobytes()
    pick = rng.choice(n_topics, n_ops // 2, replace=False)
    old = sorted({raw[off[i]:off[i + 1]] for i in pick})
    new = [b"churn/n%d/x%d" % (j % 977, j) for j in range(n_ops - len(old))]
    tdata, toff = w.tenants_packed()
    lib = B._lib.lib()

    def batch(removes, adds):
        topics = list(removes) + list(adds)
        codes = np.array([1] * len(removes) + [0] * len(adds), dtype=np.uint8)
        order = rng.permutation(len(topics))
        d, o = pack([topics[i] for i in order])
        return tuple(torch.from_numpy(x).pin_memory() for x in (d, o, codes[order]))

    fwd, back = batch(old, new), batch(new, old)
    t_tenants, t_toff = torch.from_numpy(tdata.copy()).pin_memory(), torch.from_numpy(toff.astype(np.uint32).view(np.int32)).pin_memory()
    ms = []
    for r in range(reps):
        d, o, c = fwd if r % 2 == 0 else back
        t0 = time.perf_counter()
        rc = lib.bmq_retain_apply_batch(eng.h, t_tenants.data_ptr(), t_toff.data_ptr(), 1, None, d.data_ptr(), o.data_ptr(), c.data_ptr(), None, None, len(c), None)
        ms.append((time.perf_counter() - t0) * 1e3)
        if rc:
            raise RuntimeError("bmq_retain_apply_batch failed: %d %s" % (rc, lib.bmq_last_error(eng.h)))
    d, o, c = fwd  # leave the index churned: 50 k bulk-loaded ids dead, 50 k overlay topics live
    lib.bmq_retain_apply_batch(eng.h, t_tenants.data_ptr(), t_toff.data_ptr(), 1, None, d.data_ptr(), o.data_ptr(), c.data_ptr(), None, None, len(c), None)
    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    match_ms = (time.perf_counter() - t0) / 3 * 1e3
    st = eng.stats()
    info = eng.retain_info()
    return {"ops_per_batch": n_ops, "apply_ms": [round(x, 3) for x in ms], "apply_ms_median_warm": float(np.median(ms[1:])),
            "match_ms_per_step_on_churned_index": match_ms, "kernel_ms_on_churned_index": {"k_retain_walk": st.ms_walk, "k_retain_expand_dyn": st.ms_expand},
            "index": {"retained": int(info.n_topics), "loaded_removed": int(info.loaded_removed), "added_ids": int(info.added_ids), "overlay_nodes": int(info.overlay_nodes)},
            "note": "wall time of the C-ABI call (upload from pinned memory + locate x2, commit, rank kernels + read-back); the first call allocates"}


def effective_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (cpu.max).  The GPU boxes of this
    project show 256 logical CPUs and grant a quota of 16: threads beyond that only take turns (and get throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
        except Exception:  # noqa: BLE001
            pass
    try:  # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p_ > 0:
            n = min(n, max(1, int(q / p_ + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return n


def cpu_baseline(args, w, host_batch, n, engine_csr=None):
    """The reference's algorithm (structural restatement of TenantRouteMatcher.matchAll, oracle/bmq_oracle.cpp) in the
    reference's production call pattern -- one matchAll(singleton(topic)) per publish (TenantRouteCache.java:180-193) --
    on all host cores, on a bounded sample: the first S tenants of rank 0's shard and the publishes of batch 0 that
    address them."""
    import numpy as np

    from oracle import oracle as O

    cores = effective_cpus()  # threads = the CPUs the cgroup grants (more threads only take turns)
    S = min(args.cpu_sample_tenants, w.n_tenants)
    first = w.tenant_first()
    kb, ko = w.keys_packed()
    hi = int(first[S])
    sub_off = ko[:hi + 1].copy()
    kv = O.KV(packed=(kb[:int(sub_off[-1]) + 1], sub_off))
    data, off, tt = host_batch
    sel = np.nonzero(tt < S)[0][:args.cpu_sample_topics]
    raw = data.tobytes()
    topics = [raw[off[i]:off[i + 1]] for i in sel]
    packed = O.pack(topics)
    res, sec = kv.match_singletons(w.tenants()[:S], tt[sel], packed, threads=cores)
    # How far the engine's rows are from this STRUCTURAL restatement (the reference loses routes to quirks (ii) / (iv) in
    # some rows; the engine is bit-exact against the semantic oracle): every differing row is counted and checked against the semantic oracle.
    parity = None
    if engine_csr is not None:
        try:
            from tests import util as U
            row, ids = engine_csr
            got_rp, got = U.csr_select(row, ids, sel)  # ids of the first tenants are ranks in the sub-KV of exactly those tenants
            rawk = kb[:int(sub_off[-1]) + 1].tobytes()
            differ = U.assert_csr_equal_modulo_quirk_ii(lambda: [rawk[sub_off[i]:sub_off[i + 1]] for i in range(hi)], kv.key, w.tenants()[:S], tt[sel],
                                                        res.row_ptr.astype(np.int64), res.routes, got_rp, got)
            n_sem = U.assert_differing_rows_semantic("bench " + args.workload, kv, w.tenants()[:S], tt[sel], packed, differ, got_rp, got,
                                                     livelocks=res.livelocks)
            parity = {"rows_compared": int(len(sel)), "rows_differing_from_reference_restatement": int(len(differ)),
                      "differing_rows_equal_semantic_oracle": int(n_sem), "reference_livelocks": int(res.livelocks),
                      "rule": "rows equal the restatement's id for id, except rows in which the reference LOSES routes (engine is a superset there, "
                              "every extra route belongs to a filter F for which a filter F/\"\"... exists); all of those rows equal the semantic oracle"}
        except AssertionError as ex:
            parity = {"FAILED": repr(ex)}
    # SURVEY 8d(1) also asks for the whole-batch mode: ONE matchAll(Set<topic>) per tenant with all its publishes of the batch
    # (tenants spread over the host threads; a tenant's call is sequential, as in the reference)
    from concurrent.futures import ThreadPoolExecutor
    tn = w.tenants()
    by_tenant = {}
    for i, t in zip(sel.tolist(), tt[sel].tolist()):
        by_tenant.setdefault(int(t), []).append(raw[off[i]:off[i + 1]])
    jobs = sorted(by_tenant.items(), key=lambda kv_: -len(kv_[1]))

    def whole(job):
        t, tps = job
        r = kv.match_all(tn[t], sorted(set(tps)))  # a Set<String>: duplicates collapse (TenantRouteMatcher.java:73-78)
        return r.livelocks, len(r.routes)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(cores, max(1, len(jobs)))) as ex:
        wb = list(ex.map(whole, jobs))
    sec_wb = time.perf_counter() - t0
    n_distinct = sum(len(set(v)) for v in by_tenant.values())
    return {"value": len(sel) / sec, "unit": "topics/s", "cores": cores, "logical_cpus_visible": os.cpu_count(), "kind": "port",
            "sample": "%d publishes of batch 0 addressed to the first %d tenants (%d route keys) of rank 0's shard; one "
                      "matchAll(singleton(topic)) per publish on %d threads; %.1f s" % (len(sel), S, hi, cores, sec),
            "reference_livelocks_stepped_over": int(res.livelocks),
            "parity": parity,
            "whole_batch": {"value": len(sel) / sec_wb, "unit": "topics/s", "distinct_topics": n_distinct, "seconds": sec_wb,
                            "threads": min(cores, max(1, len(jobs))),
                            "note": "the same sample, one matchAll(Set<topic>) per tenant (%d calls, the hottest tenant's call bounds the "
                                    "wall time: a call is sequential); publishes / wall time" % len(jobs),
                            "reference_livelocks_stepped_over": int(sum(x[0] for x in wb))}}


if __name__ == "__main__":
    main()
