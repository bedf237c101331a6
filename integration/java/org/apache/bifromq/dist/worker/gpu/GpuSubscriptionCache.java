/*
 * Drop-in for org.apache.bifromq.dist.worker.cache.SubscriptionCache behind ISubscriptionCache
 * (bifromq-dist-worker/.../cache/ISubscriptionCache.java:30-40; created per range replica in DistWorkerCoProcFactory.java:91-93).
 * NOT compiled in this repository (no JDK in its build image): complete enough to show every interaction with the native side.
 *
 * The reference keeps, per tenant, a Caffeine cache topic -> MatchedRoutes plus a TopicIndex of the cached topics, loads a miss
 * with matchAll(singleton(topic)) and patches entries on route mutations (TenantRouteCache.java:116-296).  Here all of that sits
 * on the native side (bmq_route_cache_*: include/bmq.h): get() is one JNI call -- a hit is answered from host memory, a miss joins
 * the GPU launch that carries every other miss of the moment -- and refresh() hands the mutated route keys to the engine, which
 * applies them to the HBM-resident index and drops the cached topics their filters match.
 * The MatchedRoutes fan-out caps (MatchedRoutes.java:87-141) are applied natively too, in KV key order, when a row is loaded: what
 * get() returns is IMatchedRoutes.routes() (TenantRouteCache.java:299-301).  This class hands the tenant's MaxPersistentFanout /
 * MaxGroupFanout settings down (routeCacheSetCaps; a changed setting re-matches exactly the cached rows MatchedRoutes.adjust would reload)
 * and forwards the throttle events of every load to the IEventCollector.
 * What stays Java: turning route ids into Matching objects (GpuTenantRouteMatcher.RangeIndex.resolve, cached per generation).
 */
package org.apache.bifromq.dist.worker.gpu;

import static org.apache.bifromq.plugin.eventcollector.ThreadLocalEventPool.getLocal;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.charset.StandardCharsets;
import java.util.LinkedHashSet;
import java.util.List;
import java.util.Map;
import java.util.Set;
import java.util.concurrent.CompletableFuture;
import java.util.concurrent.ConcurrentHashMap;
import java.util.concurrent.Executor;
import org.apache.bifromq.basekv.proto.Boundary;
import org.apache.bifromq.dist.worker.cache.ISubscriptionCache;
import org.apache.bifromq.dist.worker.cache.task.AddRoutesTask;
import org.apache.bifromq.dist.worker.cache.task.RefreshEntriesTask;
import org.apache.bifromq.dist.worker.schema.KVSchemaUtil;
import org.apache.bifromq.dist.worker.schema.cache.Matching;
import org.apache.bifromq.plugin.eventcollector.IEventCollector;
import org.apache.bifromq.plugin.eventcollector.distservice.GroupFanoutThrottled;
import org.apache.bifromq.plugin.eventcollector.distservice.PersistentFanoutThrottled;
import org.apache.bifromq.plugin.settingprovider.ISettingProvider;
import org.apache.bifromq.plugin.settingprovider.Setting;
import org.apache.bifromq.type.RouteMatcher;

final class GpuSubscriptionCache implements ISubscriptionCache {
    private final GpuTenantRouteMatcher.RangeIndex index; // engine + batching front + id -> Matching
    private final long cache;
    private final Executor matchExecutor; // dist_worker_match_parallelism threads, DistWorkerCoProcFactory.java:74-88
    private static final ThreadLocal<IntBuffer> IDS =
        ThreadLocal.withInitial(() -> ByteBuffer.allocateDirect(4 * 4096).order(ByteOrder.nativeOrder()).asIntBuffer());

    private final ISettingProvider settingProvider;
    private final ConcurrentHashMap<String, Long> capsSent = new ConcurrentHashMap<>(); // tenant -> (maxPF << 32 | maxGF) last handed down

    GpuSubscriptionCache(GpuTenantRouteMatcher.RangeIndex index, ISettingProvider settingProvider, IEventCollector eventCollector,
                         Executor matchExecutor) {
        this.index = index;
        this.settingProvider = settingProvider;
        this.matchExecutor = matchExecutor;
        // 0, 0: DistMaxCachedRoutesPerTenant (200 000) and DistTopicMatchExpirySeconds (60 s), the reference's defaults
        this.cache = NativeMatcher.routeCacheCreate(index.engine, index.batcher, 0, 0);
        // MatchedRoutes.java:95-101,124-130: every route a load throws away is reported, with the filter of the rejected route
        NativeMatcher.routeCacheSetEventSink(cache, (tenant, topic, type, routeId, maxCount) -> {
            String tenantId = new String(tenant, StandardCharsets.UTF_8);
            String tp = new String(topic, StandardCharsets.UTF_8);
            IntBuffer one = IntBuffer.wrap(new int[] {routeId});
            index.resolve(one, 0, 1).forEach(e -> eventCollector.report(type == 0
                ? getLocal(PersistentFanoutThrottled.class).tenantId(tenantId).topic(tp).mqttTopicFilter(e.matching().mqttTopicFilter())
                    .maxCount(maxCount)
                : getLocal(GroupFanoutThrottled.class).tenantId(tenantId).topic(tp).mqttTopicFilter(e.matching().mqttTopicFilter())
                    .maxCount(maxCount)));
        });
    }

    /** TenantRouteCache.java:174-175 asks the setting provider per task loop; a changed answer is handed down before the get. */
    private void syncCaps(String tenantId, byte[] tn) {
        int pf = settingProvider.provide(Setting.MaxPersistentFanout, tenantId);
        int gf = settingProvider.provide(Setting.MaxGroupFanout, tenantId);
        long packed = ((long) pf << 32) | (gf & 0xFFFFFFFFL);
        Long prev = capsSent.put(tenantId, packed);
        if (prev == null || prev != packed) {
            NativeMatcher.routeCacheSetCaps(cache, tn, pf, gf);
        }
    }

    /** SubscriptionCache.get (SubscriptionCache.java:117-122): the matched routes of (tenant, topic). */
    @Override
    public CompletableFuture<Set<Matching>> get(String tenantId, String topic) {
        // bmq_route_cache_get_async: a hit completes the future before this method returns; a miss joins the launch of all misses of the
        // moment and completes from the batching front's dispatcher thread -- no matcher thread is parked meanwhile (the reference parks one
        // per miss: TenantRouteCache.java:180-193).  Resolving ids to Matching objects may touch the KV store: handed to matchExecutor.
        CompletableFuture<int[]> ids = new CompletableFuture<>();
        byte[] tnBytes = tenantId.getBytes(StandardCharsets.UTF_8);
        syncCaps(tenantId, tnBytes);
        NativeMatcher.routeCacheGetAsync(cache, tnBytes, topic.getBytes(StandardCharsets.UTF_8),
            System.currentTimeMillis(), (status, routeIds, epoch) -> {
                if (status == 0) {
                    ids.complete(routeIds);
                } else {
                    ids.completeExceptionally(new IllegalStateException("bmq_route_cache_get_async: " + status));
                }
            });
        return ids.thenApplyAsync(routeIds -> {
            Set<Matching> out = new LinkedHashSet<>(routeIds.length * 2);
            IntBuffer buf = IntBuffer.wrap(routeIds);
            // ids -> Matching, cached per generation, the unknown ones resolved with ONE native gather; a route unsubscribed between
            // the match and now is simply absent
            index.resolve(buf, 0, routeIds.length).forEach(e -> out.add(e.matching()));
            return out;
        }, matchExecutor);
    }

    /** The same, parking the calling thread for a miss (what the reference's cache loader does on matchExecutor). */
    Set<Matching> getBlocking(String tenantId, String topic) {
        byte[] tn = tenantId.getBytes(StandardCharsets.UTF_8);
        byte[] tp = topic.getBytes(StandardCharsets.UTF_8);
        syncCaps(tenantId, tn);
        long[] epoch = new long[1];
        IntBuffer ids = IDS.get();
        long n = NativeMatcher.routeCacheGet(cache, tn, tp, System.currentTimeMillis(), ids, epoch);
        while (n < 0) { // the row is longer than the buffer: grow and ask again (it may have grown once more meanwhile)
            ids = ByteBuffer.allocateDirect((int) (-n + 64) * 4).order(ByteOrder.nativeOrder()).asIntBuffer();
            IDS.set(ids);
            n = NativeMatcher.routeCacheGet(cache, tn, tp, System.currentTimeMillis(), ids, epoch);
        }
        Set<Matching> out = new LinkedHashSet<>((int) n * 2);
        index.resolve(ids, 0, (int) n).forEach(e -> out.add(e.matching()));
        return out;
    }

    /** SubscriptionCache.isCached (:125-131) = !TopicIndex.match(filterLevels).isEmpty() */
    @Override
    public boolean isCached(String tenantId, List<String> filterLevels) {
        return NativeMatcher.routeCacheIsCached(cache, tenantId.getBytes(StandardCharsets.UTF_8),
            String.join("/", filterLevels).getBytes(StandardCharsets.UTF_8)) != 0;
    }

    /**
     * SubscriptionCache.refresh (:134-141), called from the post-commit closure of DistWorkerCoProc.mutate (:188-209) in commit
     * order: the added / removed routes become ONE bmq_route_cache_apply (ops 0 = put, 1 = delete).
     */
    @Override
    public void refresh(Map<String, RefreshEntriesTask> tenantRefreshTasks) {
        PackedKeys p = new PackedKeys();
        tenantRefreshTasks.forEach((tenantId, task) -> {
            byte op = (byte) (task instanceof AddRoutesTask ? 0 : 1);
            for (Map.Entry<RouteMatcher, Set<Matching>> e : task.routes.entrySet()) {
                for (Matching m : e.getValue()) {
                    p.add(GpuTenantRouteMatcher.routeKeyOf(tenantId, e.getKey(), m), op); // KVSchemaUtil.toNormalRouteKey / toGroupRouteKey
                }
            }
        });
        if (p.n > 0) {
            NativeMatcher.routeCacheApply(cache, p.bytes(), p.offsets(), p.ops(), p.n);
            index.forgetResolved(); // a re-subscribe keeps its id but may carry a new incarnation in the value
        }
    }

    /** SubscriptionCache.reset(boundary) (:144-146): the range's boundary changed; cached matches of the old boundary are void. */
    @Override
    public void reset(Boundary boundary) {
        NativeMatcher.routeCacheReset(cache);
    }

    /** IKVRangeCoProc.reset (DistWorkerCoProc.java:283-291): full reload from reader.iterator(); nothing old is served meanwhile. */
    void rebuild(ByteBuffer keys, IntBuffer keyOff, int n) {
        NativeMatcher.routeCacheRebuild(cache, keys, keyOff, n);
    }

    @Override
    public void close() {
        NativeMatcher.routeCacheDestroy(cache); // before the batcher and the engine (RangeIndex.close)
    }

    /** route keys + op codes packed into direct buffers (16 bytes of padding behind the last key, as include/bmq.h asks) */
    static final class PackedKeys {
        private ByteBuffer bytes = ByteBuffer.allocateDirect(1 << 16).order(ByteOrder.nativeOrder());
        private IntBuffer off = ByteBuffer.allocateDirect(4 * 1025).order(ByteOrder.nativeOrder()).asIntBuffer().put(0, 0);
        private ByteBuffer ops = ByteBuffer.allocateDirect(1024);
        int n;

        void add(com.google.protobuf.ByteString key, byte op) {
            if (bytes.remaining() < key.size() + 16) {
                bytes = grow(bytes, bytes.position() + key.size() + 16);
            }
            if (n + 2 > off.capacity()) {
                IntBuffer o = ByteBuffer.allocateDirect(8 * off.capacity()).order(ByteOrder.nativeOrder()).asIntBuffer();
                for (int i = 0; i <= n; i++) {
                    o.put(i, off.get(i));
                }
                off = o;
                // ops is written with absolute puts (its position stays 0): copy by index -- a flip()-based copy would copy nothing
                // and turn every earlier delete into a put
                ByteBuffer g = ByteBuffer.allocateDirect(2 * ops.capacity());
                for (int i = 0; i < n; i++) {
                    g.put(i, ops.get(i));
                }
                ops = g;
            }
            key.copyTo(bytes);
            ops.put(n, op);
            off.put(++n, bytes.position());
        }

        private static ByteBuffer grow(ByteBuffer b, int need) {
            ByteBuffer g = ByteBuffer.allocateDirect(Math.max(need, 2 * b.capacity())).order(ByteOrder.nativeOrder());
            b.flip();
            return g.put(b);
        }

        ByteBuffer bytes() {
            return bytes;
        }

        IntBuffer offsets() {
            return off;
        }

        ByteBuffer ops() {
            return ops;
        }
    }
}
