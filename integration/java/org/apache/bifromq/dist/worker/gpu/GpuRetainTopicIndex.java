/*
 * Drop-in for org.apache.bifromq.retain.store.index.RetainTopicIndex behind IRetainTopicIndex
 * (bifromq-retain/bifromq-retain-store/src/main/java/org/apache/bifromq/retain/store/index/IRetainTopicIndex.java:27-35).
 * NOT compiled in this repository (no JDK in its build image).
 *
 * The engine's retained-topic ids are STABLE handles (include/bmq.h): an id names the same (tenant, topic) until the next bulk load
 * (retainInfo: generation), whatever add / remove does meanwhile -- so "match, then resolve the ids" needs no lock against the
 * range's apply thread (RetainStoreCoProc.java:240-255), exactly as the reference's concurrent trie needs none.  A topic removed
 * between the match and the resolution is skipped (retainTopicInfo says it is gone), which is one of the orders the reference's
 * own race allows.  The apply loop of the coproc should hand a whole pass over as ONE NativeMatcher.retainApplyBatch; add / remove
 * below are the single-op forms the interface asks for.
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
import org.apache.bifromq.retain.store.index.IRetainTopicIndex;
import org.apache.bifromq.retain.store.index.RetainedMsgInfo;

final class GpuRetainTopicIndex implements IRetainTopicIndex {
    private final long engine;

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
        NativeMatcher.retainApplyEx(engine, tenantId.getBytes(StandardCharsets.UTF_8), topics, off, ops, ts, ex, 1);
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
        try {
            NativeMatcher.retainTopicInfo(engine, id, stamp);
        } catch (IllegalStateException gone) {
            return null; // removed since the match
        }
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
        long got = NativeMatcher.retainMatchLimited(engine, tenants, tenantOff, 1, ft, filters, filterOff, 1, lim, nowMs, row, ids, counts);
        while (got < 0) {
            ids = direct((int) (4 * -got)).asIntBuffer();
            got = NativeMatcher.retainMatchLimited(engine, tenants, tenantOff, 1, ft, filters, filterOff, 1, lim, nowMs, row, ids, counts);
        }
        for (int k = 0; k < got; k++) {
            RetainedMsgInfo m = info(ids.get(k));
            if (m != null) {
                out.add(m);
            }
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
        long[] n = new long[2];
        NativeMatcher.retainFindAll(engine, n);
        IntBuffer ids = direct(4 * (int) Math.max(64, n[0] + 1024)).asIntBuffer();
        long got = NativeMatcher.retainLiveIds(engine, null, ids); // ids are handles, not 0 .. n-1
        while (got < 0) {
            ids = direct((int) (4 * -got)).asIntBuffer();
            got = NativeMatcher.retainLiveIds(engine, null, ids);
        }
        for (int k = 0; k < got; k++) {
            RetainedMsgInfo m = info(ids.get(k));
            if (m != null) {
                out.add(m);
            }
        }
        return out;
    }
}
