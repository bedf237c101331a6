/*
 * Drop-in for org.apache.bifromq.dist.worker.cache.TenantRouteMatcher behind ITenantRouteMatcher
 * (bifromq-dist-worker/.../cache/ITenantRouteMatcher.java:37; created per (range, tenant) by
 * TenantRouteCacheFactory.create, TenantRouteCacheFactory.java:67-71).
 * NOT compiled in this repository (no JDK in its build image): complete enough to show every interaction with the native side.
 * MatchedRoutes stays the reference's own class and is fed in KV KEY order, so the fan-out caps and throttle events behave
 * exactly as today (MatchedRoutes.java:87-141).  Route ids are STABLE handles (include/bmq.h): ranks after a rebuild, later
 * ids for routes subscribed since; an id resolves to the same key until the next rebuild ("generation"), a deleted route's id
 * resolves to an empty key -- so the id -> Matching cache is never wrong, whatever the apply thread is doing meanwhile.
 */
package org.apache.bifromq.dist.worker.gpu;

import com.google.protobuf.ByteString;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.LongBuffer;
import java.util.ArrayList;
import java.util.List;
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
        /** (route key, Matching) per route id of ONE generation; dropped as a whole when bmq_rebuild re-numbers the routes. */
        record Entry(ByteString key, Matching matching) {
        }

        private volatile long cachedGeneration = -1;
        private volatile ConcurrentHashMap<Integer, Entry> entries = new ConcurrentHashMap<>();
        private final Function<ByteString, ByteString> valueOfKey; // reader.get(routeKey): incarnation / RouteGroup

        RangeIndex(int device, Function<ByteString, ByteString> valueOfKey) {
            this.engine = NativeMatcher.create(device);
            this.batcher = NativeMatcher.batcherCreate(engine, 0);
            this.valueOfKey = valueOfKey;
        }

        /** DistWorkerCoProc.reset (DW/DistWorkerCoProc.java:283-291): the keys of reader.iterator(), packed. */
        void reset(ByteBuffer keys, IntBuffer keyOff, int n) {
            NativeMatcher.rebuild(engine, keys, keyOff, n); // a new generation: every cached id is void
        }

        /** Post-commit closure of DistWorkerCoProc.mutate (:188-209): added / removed route keys, in commit order. */
        void refresh(ByteBuffer keys, IntBuffer keyOff, ByteBuffer ops, int n) {
            NativeMatcher.routesApply(engine, keys, keyOff, ops, n);
            // a re-subscribe keeps its id but may carry a new incarnation (or a new RouteGroup value): the Matching resolved for
            // that id before is stale.  The cache is keyed by id and the put keys' ids are not known here, so everything resolved
            // so far is dropped; the next resolve() gathers what it needs again in one native call.
            entries.clear();
        }

        /** After route keys were re-put through another path (GpuSubscriptionCache.refresh): drop what was resolved before. */
        void forgetResolved() {
            entries.clear();
        }

        /** ids -> (key, Matching), resolving the unknown ones with ONE native gather. */
        List<Entry> resolve(IntBuffer ids, int from, int to) {
            long gen = NativeMatcher.generation(engine);
            if (gen != cachedGeneration) {
                synchronized (this) {
                    if (gen != cachedGeneration) {
                        entries = new ConcurrentHashMap<>();
                        cachedGeneration = gen;
                    }
                }
            }
            ConcurrentHashMap<Integer, Entry> map = entries;
            int missing = 0;
            for (int k = from; k < to; k++) {
                if (!map.containsKey(ids.get(k))) {
                    missing++;
                }
            }
            if (missing > 0) {
                IntBuffer want = ByteBuffer.allocateDirect(4 * missing).order(ByteOrder.nativeOrder()).asIntBuffer();
                for (int k = from; k < to; k++) {
                    if (!map.containsKey(ids.get(k))) {
                        want.put(ids.get(k));
                    }
                }
                LongBuffer off = ByteBuffer.allocateDirect(8 * (missing + 1)).order(ByteOrder.nativeOrder()).asLongBuffer();
                ByteBuffer bytes = ByteBuffer.allocateDirect(128 * missing);
                long got = NativeMatcher.routeKeys(engine, want, missing, bytes, off);
                if (got < 0) {
                    bytes = ByteBuffer.allocateDirect((int) -got);
                    got = NativeMatcher.routeKeys(engine, want, missing, bytes, off);
                }
                for (int j = 0; j < missing; j++) {
                    int len = (int) (off.get(j + 1) - off.get(j));
                    if (len == 0) {
                        continue; // unsubscribed between the match and now: the route is gone, as a later matchAll would say
                    }
                    ByteBuffer slice = bytes.duplicate();
                    slice.position((int) off.get(j)).limit((int) off.get(j + 1));
                    ByteString key = ByteString.copyFrom(slice);
                    map.put(want.get(j), new Entry(key, KVSchemaUtil.buildMatchRoute(key, valueOfKey.apply(key)))); // KVSchemaUtil.java:73-89
                }
            }
            List<Entry> out = new ArrayList<>(to - from);
            for (int k = from; k < to; k++) {
                Entry e = map.get(ids.get(k));
                if (e != null) {
                    out.add(e);
                }
            }
            return out;
        }

        @Override
        public void close() {
            NativeMatcher.batcherDestroy(batcher); // before the engine
            NativeMatcher.destroy(engine);
        }
    }

    /** The KV key of one route of a RefreshEntriesTask: KVSchemaUtil.toNormalRouteKey / toGroupRouteKey (KVSchemaUtil.java:108-120). */
    static ByteString routeKeyOf(String tenantId, org.apache.bifromq.type.RouteMatcher matcher, Matching m) {
        return m.type() == Matching.Type.Normal
            ? KVSchemaUtil.toNormalRouteKey(tenantId, matcher, ((NormalMatching) m).receiverUrl())
            : KVSchemaUtil.toGroupRouteKey(tenantId, matcher);
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
    public Map<String, IMatchedRoutes> matchAll(Set<String> topicSet, int maxPersistentFanoutCount, int maxGroupFanoutCount) {
        // in ORDER: neighbouring topics share the trie lines of their common prefixes (the walk kernel sends a third fewer L2 requests for an
        // ordered batch, profiles/r05b/README.md); only speed depends on it, so String order (UTF-16) is as good as byte order.  A
        // BatchDistRequest's topics arrive sorted already (DistWorkerCoProc.proto:75-83)
        Set<String> topics = topicSet instanceof java.util.SortedSet ? topicSet : new java.util.TreeSet<>(topicSet);
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
        while (got < 0) { // ids too small: -got ints are needed (IntBuffer capacities are in ints)
            ids = ByteBuffer.allocateDirect((int) (4 * -got)).order(ByteOrder.nativeOrder()).asIntBuffer();
            got = NativeMatcher.batcherMatchAll(index.batcher, tenantBytes, bytes, off, n, rowPtr, ids, epoch);
        }
        Map<String, IMatchedRoutes> out = new HashMap<>();
        i = 0;
        for (String topic : topics) { // every input topic is a key, also with 0 routes (TenantRouteMatcherTest.java:90-110)
            MatchedRoutes mr = new MatchedRoutes(tenantId, topic, eventCollector, maxPersistentFanoutCount, maxGroupFanoutCount);
            List<RangeIndex.Entry> row = index.resolve(ids, rowPtr.get(i), rowPtr.get(i + 1));
            // MatchedRoutes caps first-come in KV key order.  Ids are key ranks only for routes loaded by the last rebuild, so when a
            // cap can bind the row is ordered by key bytes first (ByteString's unsigned lexicographical order == KV order).
            int persistent = 0, groups = 0;
            for (RangeIndex.Entry e : row) {
                if (e.matching().type() == Matching.Type.Group) {
                    groups++;
                } else if (((NormalMatching) e.matching()).subBrokerId() == 1) {
                    persistent++;
                }
            }
            if (persistent > maxPersistentFanoutCount || groups > maxGroupFanoutCount) {
                row.sort((a, b) -> ByteString.unsignedLexicographicalComparator().compare(a.key(), b.key()));
            }
            for (RangeIndex.Entry e : row) {
                Matching m = e.matching();
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
