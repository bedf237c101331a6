/*
 * Drop-in for org.apache.bifromq.dist.worker.cache.TenantRouteMatcher behind ITenantRouteMatcher
 * (bifromq-dist-worker/.../cache/ITenantRouteMatcher.java:37; created per (range, tenant) by
 * TenantRouteCacheFactory.create, TenantRouteCacheFactory.java:67-71).
 * NOT compiled in this repository (no JDK in its build image): a sketch complete enough to show every interaction with
 * the native side.  MatchedRoutes stays the reference's own class, fed in ascending route-id (= KV key) order, so the
 * fan-out caps and throttle events behave exactly as today (MatchedRoutes.java:87-141).
 */
package org.apache.bifromq.dist.worker.gpu;

import com.google.protobuf.ByteString;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.charset.StandardCharsets;
import java.util.HashMap;
import java.util.Map;
import java.util.Set;
import java.util.concurrent.ConcurrentHashMap;
import java.util.function.Function;
import org.apache.bifromq.dist.worker.cache.IMatchedRoutes;
import org.apache.bifromq.dist.worker.cache.ITenantRouteMatcher;
import org.apache.bifromq.dist.worker.cache.MatchedRoutes;
import org.apache.bifromq.dist.worker.schema.KVSchemaUtil;
import org.apache.bifromq.dist.worker.schema.cache.GroupMatching;
import org.apache.bifromq.dist.worker.schema.cache.Matching;
import org.apache.bifromq.dist.worker.schema.cache.NormalMatching;
import org.apache.bifromq.plugin.eventcollector.IEventCollector;

final class GpuTenantRouteMatcher implements ITenantRouteMatcher {
    /** One per KV range replica: engine + batching front, shared by the matchers of all tenants of the range. */
    static final class RangeIndex implements AutoCloseable {
        final long engine;
        final long batcher;
        // route id -> Matching, valid for one epoch (bmq_rebuild / bmq_routes_apply bump it)
        private volatile long cachedEpoch = -1;
        private volatile ConcurrentHashMap<Integer, Matching> matchings = new ConcurrentHashMap<>();
        private final Function<ByteString, ByteString> valueOfKey; // reader.get(routeKey): incarnation / RouteGroup

        RangeIndex(int device, Function<ByteString, ByteString> valueOfKey) {
            this.engine = NativeMatcher.create(device);
            this.batcher = NativeMatcher.batcherCreate(engine, 0);
            this.valueOfKey = valueOfKey;
        }

        Matching matchingOf(int routeId, long epoch) {
            if (epoch != cachedEpoch) {
                synchronized (this) {
                    if (epoch != cachedEpoch) {
                        matchings = new ConcurrentHashMap<>();
                        cachedEpoch = epoch;
                    }
                }
            }
            return matchings.computeIfAbsent(routeId, id -> {
                ByteBuffer out = ByteBuffer.allocateDirect(512);
                int len = NativeMatcher.routeKey(engine, id, out);
                if (len < 0) {
                    out = ByteBuffer.allocateDirect(-len);
                    len = NativeMatcher.routeKey(engine, id, out);
                }
                out.limit(len);
                ByteString key = ByteString.copyFrom(out);
                return KVSchemaUtil.buildMatchRoute(key, valueOfKey.apply(key)); // KVSchemaUtil.java:73-89
            });
        }

        @Override
        public void close() {
            NativeMatcher.batcherDestroy(batcher); // before the engine
            NativeMatcher.destroy(engine);
        }
    }

    private final String tenantId;
    private final byte[] tenantBytes;
    private final RangeIndex index;
    private final IEventCollector eventCollector;

    GpuTenantRouteMatcher(String tenantId, RangeIndex index, IEventCollector eventCollector) {
        this.tenantId = tenantId;
        this.tenantBytes = tenantId.getBytes(StandardCharsets.UTF_8);
        this.index = index;
        this.eventCollector = eventCollector;
    }

    @Override
    public Map<String, IMatchedRoutes> matchAll(Set<String> topics, int maxPersistentFanoutCount, int maxGroupFanoutCount) {
        // pack the topics: UTF-8 bytes + int offsets, direct buffers
        int n = topics.size();
        byte[][] utf8 = new byte[n][];
        int total = 0, i = 0;
        for (String t : topics) {
            utf8[i] = t.getBytes(StandardCharsets.UTF_8);
            total += utf8[i++].length;
        }
        ByteBuffer bytes = ByteBuffer.allocateDirect(total + 16).order(ByteOrder.nativeOrder());
        IntBuffer off = ByteBuffer.allocateDirect(4 * (n + 1)).order(ByteOrder.nativeOrder()).asIntBuffer();
        off.put(0, 0);
        for (i = 0; i < n; i++) {
            bytes.put(utf8[i]);
            off.put(i + 1, bytes.position());
        }
        IntBuffer rowPtr = ByteBuffer.allocateDirect(4 * (n + 1)).order(ByteOrder.nativeOrder()).asIntBuffer();
        IntBuffer ids = ByteBuffer.allocateDirect(4 * Math.max(64, 16 * n)).order(ByteOrder.nativeOrder()).asIntBuffer();
        long[] epoch = new long[1];
        // the calling matchExecutor thread parks here until the GPU launch that carries these topics is done; the calls of all
        // threads waiting at this moment share that launch (bmq_batcher_match_all)
        long got = NativeMatcher.batcherMatchAll(index.batcher, tenantBytes, bytes, off, n, rowPtr, ids, epoch);
        if (got < 0) { // ids too small: grow and ask again
            ids = ByteBuffer.allocateDirect((int) (4 * -got)).order(ByteOrder.nativeOrder()).asIntBuffer();
            got = NativeMatcher.batcherMatchAll(index.batcher, tenantBytes, bytes, off, n, rowPtr, ids, epoch);
        }
        Map<String, IMatchedRoutes> out = new HashMap<>();
        i = 0;
        for (String topic : topics) { // every input topic is a key, also with 0 routes (TenantRouteMatcherTest.java:90-110)
            MatchedRoutes mr = new MatchedRoutes(tenantId, topic, eventCollector, maxPersistentFanoutCount, maxGroupFanoutCount);
            for (int k = rowPtr.get(i); k < rowPtr.get(i + 1); k++) { // ascending id == KV key order
                Matching m = index.matchingOf(ids.get(k), epoch[0]);
                switch (m.type()) {
                    case Normal -> mr.addNormalMatching((NormalMatching) m);
                    case Group -> mr.putGroupMatching((GroupMatching) m);
                    default -> {
                    }
                }
            }
            out.put(topic, mr);
            i++;
        }
        return out;
    }
}
