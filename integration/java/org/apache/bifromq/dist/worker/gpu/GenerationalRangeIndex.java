/*
 * Compaction of the route index WITHOUT a stall, on the JVM's side of the boundary: two native engine handles for the length of a
 * compaction (INTEGRATION.md section 3; the same procedure as bifromq_amd/generations.py, whose bookkeeping is tested there).
 * What it answers: TopicLevelTrie contracts tombed nodes as it goes (bifromq-util/.../index/TopicLevelTrie.java:257-384); the native index
 * only grows between rebuilds, and bmq_compact rebuilds INSIDE the engine while every entry point waits.
 * NOT compiled in this repository (no JDK in its build image).
 *
 *   serving handle A: matchers pin it per call, mutations go to it (and, while a compaction runs, to the log)
 *   1. start logging   2. export A's live route keys, create + rebuild handle B from them (A keeps serving)
 *   3. replay the log into B in rounds, the last one under the lock   4. swap; A is destroyed when its last pinned caller has left
 * Route ids are renumbered by the swap exactly as by bmq_compact: whatever is kept per route id (GpuTenantRouteMatcher.RangeIndex's
 * id -> Matching cache, the batching front, the route cache) belongs to ONE generation and is re-created on the new handle.
 */
package org.apache.bifromq.dist.worker.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.LongBuffer;
import java.util.ArrayList;
import java.util.List;
import java.util.concurrent.atomic.AtomicInteger;
import java.util.concurrent.locks.ReentrantLock;

final class GenerationalRangeIndex implements AutoCloseable {
    /** One native engine and the callers inside it. */
    static final class Generation {
        final long engine;
        final int number;
        final AtomicInteger pins = new AtomicInteger();
        volatile boolean retired;

        Generation(long engine, int number) {
            this.engine = engine;
            this.number = number;
        }
    }

    /** One mutation as DistWorkerCoProc's post-commit closure hands it over (:188-209). */
    record Op(byte[] routeKey, boolean delete) {
    }

    private static final int EXPORT_CHUNK = 1 << 20;
    private static final int MAX_REPLAY_ROUNDS = 8;
    private final int device;
    private final ReentrantLock lock = new ReentrantLock(); // guards current, log
    private volatile Generation current;
    private List<Op> log; // non-null while a compaction runs

    GenerationalRangeIndex(int device) {
        this.device = device;
        this.current = new Generation(NativeMatcher.create(device), 0);
    }

    /** Callers match through a pinned generation: ids they get are ids OF that generation (resolve them through its engine). */
    Generation pin() {
        lock.lock();
        try {
            current.pins.incrementAndGet();
            return current;
        } finally {
            lock.unlock();
        }
    }

    void unpin(Generation g) {
        if (g.pins.decrementAndGet() == 0 && g.retired) {
            NativeMatcher.destroy(g.engine);
        }
    }

    /** Mutations reach the serving generation in commit order, and the log of a running compaction. */
    void apply(List<Op> ops) {
        lock.lock();
        try {
            applyTo(current.engine, ops);
            if (log != null) {
                log.addAll(ops);
            }
        } finally {
            lock.unlock();
        }
    }

    /** Worth it when the garbage the churn left behind is a noticeable share of the index (bmq_index_info.garbage_bytes). */
    boolean compactionPays() {
        long[] info = new long[11];
        NativeMatcher.indexInfo(current.engine, info);
        return info[10] * 4 > info[6] || info[9] > 2 * info[0]; // garbageBytes > deviceBytes / 4, or half of the ids handed out are dead
    }

    /** Builds the next generation beside the serving one and swaps.  One at a time (the caller's maintenance thread). */
    void compactOnline() {
        final Generation a;
        final long nIds;
        lock.lock();
        try {
            a = current;
            // (what can throw comes BEFORE the pin and the log exist: a failing indexInfo used to leak both -- ADVICE r4)
            long[] info = new long[11];
            NativeMatcher.indexInfo(a.engine, info);
            nIds = info[9];
            a.pins.incrementAndGet(); // A outlives the export whatever happens
            log = new ArrayList<>();
        } finally {
            lock.unlock();
        }
        long b = 0;
        try {
            // 2. A's live keys: a key deleted meanwhile comes back empty (its delete is in the log: a no-op on B), a key added meanwhile has
            //    an id >= nIds (its put is in the log).  The KV scan order is the key order: sort before the bulk load.
            List<byte[]> live = new ArrayList<>();
            ByteBuffer out = NativeMatcher.hostAlloc(96L * EXPORT_CHUNK);
            IntBuffer ids = ByteBuffer.allocateDirect(4 * EXPORT_CHUNK).order(ByteOrder.nativeOrder()).asIntBuffer();
            LongBuffer off = ByteBuffer.allocateDirect(8 * (EXPORT_CHUNK + 1)).order(ByteOrder.nativeOrder()).asLongBuffer();
            for (long lo = 0; lo < nIds; lo += EXPORT_CHUNK) {
                int n = (int) Math.min(EXPORT_CHUNK, nIds - lo);
                for (int i = 0; i < n; i++) {
                    ids.put(i, (int) (lo + i));
                }
                long bytes = NativeMatcher.routeKeys(a.engine, ids, n, out, off);
                if (bytes < 0) { // the chunk's keys are longer than 96 bytes on average: a buffer of the size asked for
                    NativeMatcher.hostFree(out);
                    out = NativeMatcher.hostAlloc(-bytes);
                    NativeMatcher.routeKeys(a.engine, ids, n, out, off);
                }
                for (int i = 0; i < n; i++) {
                    int len = (int) (off.get(i + 1) - off.get(i));
                    if (len > 0) {
                        byte[] k = new byte[len];
                        out.position((int) off.get(i)).get(k);
                        live.add(k);
                    }
                }
            }
            NativeMatcher.hostFree(out);
            live.sort(java.util.Arrays::compareUnsigned);
            b = NativeMatcher.create(device);
            rebuildFrom(b, live);
            // 3. replay: outside the lock while the log keeps filling, the last round under it; 4. swap
            for (int round = 0; ; round++) {
                List<Op> chunk;
                lock.lock();
                try {
                    chunk = log;
                    log = new ArrayList<>();
                    if (chunk.isEmpty() || round >= MAX_REPLAY_ROUNDS) {
                        applyTo(b, chunk);
                        log = null;
                        current = new Generation(b, a.number + 1);
                        a.retired = true;
                        b = 0;
                        return;
                    }
                } finally {
                    lock.unlock();
                }
                applyTo(b, chunk);
            }
        } finally {
            lock.lock();
            try {
                if (current == a) {
                    log = null; // failed before the swap: A goes on serving
                }
            } finally {
                lock.unlock();
            }
            if (b != 0) {
                NativeMatcher.destroy(b);
            }
            unpin(a);
        }
    }

    @Override
    public void close() {
        Generation g = current;
        g.retired = true;
        if (g.pins.get() == 0) {
            NativeMatcher.destroy(g.engine);
        }
    }

    private static void applyTo(long engine, List<Op> ops) {
        if (ops.isEmpty()) {
            return;
        }
        int bytes = 0;
        for (Op op : ops) {
            bytes += op.routeKey().length;
        }
        ByteBuffer keys = ByteBuffer.allocateDirect(bytes + 16).order(ByteOrder.nativeOrder());
        IntBuffer keyOff = ByteBuffer.allocateDirect(4 * (ops.size() + 1)).order(ByteOrder.nativeOrder()).asIntBuffer();
        ByteBuffer kinds = ByteBuffer.allocateDirect(ops.size());
        int at = 0;
        for (int i = 0; i < ops.size(); i++) {
            keyOff.put(i, at);
            keys.put(ops.get(i).routeKey());
            at += ops.get(i).routeKey().length;
            kinds.put(i, (byte) (ops.get(i).delete() ? 1 : 0));
        }
        keyOff.put(ops.size(), at);
        NativeMatcher.routesApply(engine, keys, keyOff, kinds, ops.size());
    }

    private static void rebuildFrom(long engine, List<byte[]> sortedKeys) {
        long bytes = 0;
        for (byte[] k : sortedKeys) {
            bytes += k.length;
        }
        ByteBuffer keys = NativeMatcher.hostAlloc(bytes + 16); // page-locked: the bulk load reads it at PCIe speed
        IntBuffer keyOff = ByteBuffer.allocateDirect(4 * (sortedKeys.size() + 1)).order(ByteOrder.nativeOrder()).asIntBuffer();
        int at = 0;
        for (int i = 0; i < sortedKeys.size(); i++) {
            keyOff.put(i, at);
            keys.put(sortedKeys.get(i));
            at += sortedKeys.get(i).length;
        }
        keyOff.put(sortedKeys.size(), at);
        NativeMatcher.rebuild(engine, keys, keyOff, sortedKeys.size());
        NativeMatcher.hostFree(keys);
    }
}
