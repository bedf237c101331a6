/*
 * Drop-in for org.apache.bifromq.retain.store.index.RetainTopicIndex behind IRetainTopicIndex
 * (bifromq-retain/bifromq-retain-store/src/main/java/org/apache/bifromq/retain/store/index/IRetainTopicIndex.java:27-35).
 * NOT compiled in this repository (no JDK in its build image).
 *
 * The engine's retained-topic ids are RANKS of (tenant, level list): every add/remove shifts them.  The adapter therefore takes
 * a read lock around "match + resolve ids" and the write lock around add/remove -- the reference's callers already behave that
 * way: add/remove run on the range's apply thread post-commit (RetainStoreCoProc.java:240-255), match on query threads.
 * RetainStoreCoProc.match(limit, now) itself (RetainStoreCoProc.java:167-190) is better served by matchLimited(): the engine
 * returns the first `limit` topics that have NOT expired, so the coproc only point-gets messages it will actually return.
 */
package org.apache.bifromq.dist.worker.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.LongBuffer;
import java.nio.charset.StandardCharsets;
import java.util.HashSet;
import java.util.Set;
import java.util.concurrent.locks.ReentrantReadWriteLock;
import org.apache.bifromq.retain.store.index.IRetainTopicIndex;
import org.apache.bifromq.retain.store.index.RetainedMsgInfo;

final class GpuRetainTopicIndex implements IRetainTopicIndex {
    private final long engine;
    private final ReentrantReadWriteLock lock = new ReentrantReadWriteLock();

    GpuRetainTopicIndex(long engine) {
        this.engine = engine;
    }

    private static ByteBuffer direct(int bytes) {
        return ByteBuffer.allocateDirect(bytes).order(ByteOrder.nativeOrder());
    }

    private void apply(String tenantId, String topic, byte op, long timestamp, int expirySeconds) {
        byte[] t = topic.getBytes(StandardCharsets.UTF_8);
        ByteBuffer topics = direct(t.length + 16).put(t);
        IntBuffer off = direct(8).asIntBuffer().put(0, 0).put(1, t.length);
        ByteBuffer ops = direct(1).put(0, op);
        LongBuffer ts = direct(8).asLongBuffer().put(0, timestamp);
        IntBuffer ex = direct(4).asIntBuffer().put(0, expirySeconds);
        lock.writeLock().lock();
        try {
            NativeMatcher.retainApplyEx(engine, tenantId.getBytes(StandardCharsets.UTF_8), topics, off, ops, ts, ex, 1);
        } finally {
            lock.writeLock().unlock();
        }
    }

    @Override
    public void add(String tenantId, String topic, long timestamp, int expirySeconds) {
        apply(tenantId, topic, (byte) 0, timestamp, expirySeconds); // an add of a topic that is there replaces its stamp
    }

    @Override
    public void remove(String tenantId, String topic) {
        apply(tenantId, topic, (byte) 1, 0, 0);
    }

    private RetainedMsgInfo info(int id) {
        ByteBuffer out = direct(512);
        long[] tenantLen = new long[1];
        int len = NativeMatcher.retainTopic(engine, id, out, tenantLen);
        if (len < 0) {
            out = direct(-len);
            len = NativeMatcher.retainTopic(engine, id, out, tenantLen);
        }
        byte[] raw = new byte[len];
        out.get(0, raw);
        int tl = (int) tenantLen[0];
        long[] stamp = new long[3];
        NativeMatcher.retainTopicInfo(engine, id, stamp);
        return new RetainedMsgInfo(new String(raw, 0, tl, StandardCharsets.UTF_8), new String(raw, tl, len - tl, StandardCharsets.UTF_8),
            stamp[0], (int) stamp[1]);
    }

    /** limit < 0: the complete match set (IRetainTopicIndex.match); else RetainStoreCoProc.match(limit, now). */
    Set<RetainedMsgInfo> matchLimited(String tenantId, String topicFilter, int limit, long nowMs) {
        byte[] tn = tenantId.getBytes(StandardCharsets.UTF_8), f = topicFilter.getBytes(StandardCharsets.UTF_8);
        ByteBuffer tenants = direct(tn.length + 16).put(tn);
        IntBuffer tenantOff = direct(8).asIntBuffer().put(0, 0).put(1, tn.length);
        ByteBuffer filters = direct(f.length + 16).put(f);
        IntBuffer filterOff = direct(8).asIntBuffer().put(0, 0).put(1, f.length);
        IntBuffer ft = direct(4).asIntBuffer().put(0, 0);
        IntBuffer lim = direct(4).asIntBuffer().put(0, limit < 0 ? -1 : limit); // 0xFFFFFFFF = no limit
        IntBuffer row = direct(8).asIntBuffer(), counts = direct(4).asIntBuffer();
        IntBuffer ids = direct(4 * Math.max(64, limit < 0 ? 4096 : limit)).asIntBuffer();
        Set<RetainedMsgInfo> out = new HashSet<>();
        lock.readLock().lock(); // ids are ranks: resolve them before the next add/remove can shift them
        try {
            long got = NativeMatcher.retainMatchLimited(engine, tenants, tenantOff, 1, ft, filters, filterOff, 1, lim, nowMs, row, ids, counts);
            while (got < 0) {
                ids = direct((int) (4 * -got)).asIntBuffer();
                got = NativeMatcher.retainMatchLimited(engine, tenants, tenantOff, 1, ft, filters, filterOff, 1, lim, nowMs, row, ids, counts);
            }
            for (int k = 0; k < got; k++) {
                out.add(info(ids.get(k)));
            }
        } finally {
            lock.readLock().unlock();
        }
        return out;
    }

    @Override
    public Set<RetainedMsgInfo> match(String tenantId, String topicFilter) {
        return matchLimited(tenantId, topicFilter, -1, 0);
    }

    @Override
    public Set<RetainedMsgInfo> findAll() {
        Set<RetainedMsgInfo> out = new HashSet<>();
        lock.readLock().lock();
        try {
            long[] n = new long[2];
            NativeMatcher.retainFindAll(engine, n); // the ids are 0 .. n-1
            for (int id = 0; id < n[0]; id++) {
                out.add(info(id));
            }
        } finally {
            lock.readLock().unlock();
        }
        return out;
    }
}
