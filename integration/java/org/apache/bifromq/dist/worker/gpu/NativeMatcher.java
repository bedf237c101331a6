/*
 * Java face of libbmq.so (include/bmq.h) through integration/jni/bmq_jni.c.
 * NOT compiled in this repository (its build image has no JDK); it is the file a bifromq maintainer adds next to
 * bifromq-dist-worker's cache package.  All buffers are DIRECT and in native byte order.
 */
package org.apache.bifromq.dist.worker.gpu;

import java.nio.ByteBuffer;
import java.nio.IntBuffer;
import java.nio.LongBuffer;

final class NativeMatcher {
    static {
        System.loadLibrary("bmq_jni"); // links against libbmq.so
    }

    private NativeMatcher() {
    }

    // ---- engine ----
    static native long create(int device);

    /** bmq_config.dedup_sorted: for a caller that hands over the publishes of a BatchDistRequest in the request's own order -- "sorted by tenantId
     *  and topic" (DistWorkerCoProc.proto:75-83) -- repeats included: batches of at least dedupMinTopics rows are reduced to their distinct rows on the
     *  device by comparing neighbours (every row keeps its own row in the result).  matchAll(Set) callers need none of it: a set has no repeats. */
    static native long createOrdered(int device, int dedupMinTopics);

    /** bmq_config.region_slack: the filter trie's regions hold nodes x (1 + regionSlack / 4) buckets -- 0: the default (6, load factor 0.2),
     *  1: half the region memory for about 8 % more time in the match kernel (INTEGRATION.md, "Memory against speed"). */
    static native long createCompact(int device, int regionSlack);

    static native void destroy(long engine);

    /** IKVRangeCoProc.reset(): all route keys of the range (a KV scan: ascending; route id = rank).  Built on the GPU. */
    static native void rebuild(long engine, ByteBuffer keys, IntBuffer keyOff, int n);

    /** Post-commit AddRoutesTask / RemoveRoutesTask: ops[i] 0 = put, 1 = delete; applied in order. */
    static native void routesApply(long engine, ByteBuffer keys, IntBuffer keyOff, ByteBuffer ops, int n);

    /** bmq_routes_apply_async / _wait: uploaded beside the batch in flight, applied behind it; the buffers stay untouched until routesApplyWait
     *  (or the next call that needs the route index) has returned -- that call throws what routesApply would have thrown. */
    static native void routesApplyAsync(long engine, ByteBuffer keys, IntBuffer keyOff, ByteBuffer ops, int n);

    static native void routesApplyWait(long engine);

    static native long epoch(long engine);

    /** +1 per rebuild: route ids are stable handles within a generation (a deleted route's id resolves to an empty key). */
    static native long generation(long engine);

    /** bmq_index_info: {routes, tenants, nodes, tokens, trieSlots, dictSlots, deviceBytes, epoch, generation, nextRouteId, garbageBytes}. */
    static native void indexInfo(long engine, long[] out11);

    /** bmq_compact_begin / _poll / _swap / _abort: the next generation of the route index is built beside the serving one INSIDE the handle (its own
     *  stream, keys that never leave HBM); compactPoll(maxIds) from a maintenance thread until it returns 1000, then compactSwap (no batch in flight;
     *  route ids are re-numbered: everything keyed by route id is re-created).  GenerationalRangeIndex is the same with two handles. */
    static native void compactBegin(long engine);

    static native int compactPoll(long engine, int maxIds);

    static native void compactSwap(long engine, long[] out2);

    static native void compactAbort(long engine);

    /** One device gather: outOff[n + 1] byte offsets into out.  @return bytes, or -(needed) */
    static native long routeKeys(long engine, IntBuffer ids, int n, ByteBuffer out, LongBuffer outOff);

    /** Page-locked direct buffer (bmq_host_alloc): hand these to matchSubmit / routesApply for full PCIe speed. */
    static native ByteBuffer hostAlloc(long bytes);

    static native void hostFree(ByteBuffer buf);

    /** Up to three batches in flight: upload of batch i+1 and download of batch i-1 overlap the kernels of batch i. @return ticket */
    static native int matchSubmit(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant,
                                  ByteBuffer topics, IntBuffer topicOff, int nTopics);

    /** @return number of ids, or -(needed) when outIds is too small (the ticket is consumed either way: re-submit) */
    static native long matchWait(long engine, int ticket, IntBuffer outRowPtr, IntBuffer outIds);

    /** Result formats that fit the wire (bmq.h BMQ_FMT_*): 0 ids, 1 fan-out counts, 2 matched id ranges, 3 pairs grouped by DelivererKey. */
    static native int matchSubmitFmt(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant,
                                     ByteBuffer topics, IntBuffer topicOff, int nTopics, int format);

    /** Fan-out of topic i = outRowPtr[i + 1] - outRowPtr[i] (all a BatchDistReply carries). @return total fan-out */
    static native long matchWaitCounts(long engine, int ticket, IntBuffer outRowPtr);

    /** @return number of ranges, or -(needed); info4 = {ranges, side ids, ids, rows whose expanded ids must be ordered} */
    static native long matchWaitRanges(long engine, int ticket, IntBuffer outRowPtr, IntBuffer outRangePtr, IntBuffer outRanges,
                                       IntBuffer outSideIds, long[] info4);

    /** @return number of (topic, route) pairs, or -(needed); groups2 = {groups, special bits} */
    static native long matchWaitGrouped(long engine, int ticket, IntBuffer outTopic, IntBuffer outRoute, IntBuffer outGroupOff,
                                        IntBuffer outGroupRep, int[] groups2);

    /** @return key length, or -(needed) when out is too small */
    static native int routeKey(long engine, int routeId, ByteBuffer out);

    // ---- dist direction ----
    /** @return number of ids, or -(needed) when outIds is too small */
    static native long matchBatch(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant,
                                  ByteBuffer topics, IntBuffer topicOff, int nTopics, IntBuffer outRowPtr, IntBuffer outIds);

    static native long matchAll(long engine, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, int nTopics,
                                int maxPersistentFanout, int maxGroupFanout, IntBuffer outRowPtr, IntBuffer outIds,
                                IntBuffer outEvents, long[] nEventsOut);

    // ---- batching front ----
    static native long batcherCreate(long engine, int maxBatchTopics);

    static native void batcherDestroy(long batcher);

    /** Blocks until the launch that carries these topics has finished; epochOut[0] = epoch of the index the batch saw. */
    static native long batcherMatchAll(long batcher, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, int nTopics,
                                       IntBuffer outRowPtr, IntBuffer outIds, long[] epochOut);

    // ---- retain direction ----
    static native void retainRebuild(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant,
                                     ByteBuffer topics, IntBuffer topicOff, int nTopics);

    static native void retainApply(long engine, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, ByteBuffer ops, int n);

    /** Adds / removes of several tenants in one call, applied by kernels behind the batches in flight; outTopicIds may be null. */
    static native void retainApplyBatch(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer opTenant,
                                        ByteBuffer topics, IntBuffer topicOff, ByteBuffer ops, LongBuffer timestampHlc,
                                        IntBuffer expirySeconds, int n, IntBuffer outTopicIds);

    /** Maintenance: a fresh bulk load of the live topics (a new generation of ids). */
    static native void retainCompact(long engine);

    /** The same without the stall (bmq_retain_compact_begin / _build / _swap / _abort): begin = snapshot, build = the load with NO engine lock held
     *  (matching and add / remove go on, what they change is logged), swap = upload + replay; topic ids are re-numbered. */
    static native void retainCompactBegin(long engine);

    static native void retainCompactBuild(long engine);

    static native void retainCompactSwap(long engine, long[] out2);

    static native void retainCompactAbort(long engine);

    /** The persistent matcher behind batcherMatchAll / routeCacheGet (bmq_poller_*): out8 = {enabled, running, starts, served, fallback, unserved,
     *  timeouts, badInput}; pollerControl: 0 off, 1 on, 2 leave now. */
    static native void pollerStats(long engine, long[] out8);

    static native void pollerControl(long engine, int what);

    /** out9 = {topics, tenants, idBound, loaded, loadedRemoved, addedIds, overlayNodes, epoch, generation} */
    static native void retainInfo(long engine, long[] out9);

    /** Ids of the retained topics (of one tenant, or of all: tenant == null), ascending. @return count, or -(needed) */
    static native long retainLiveIds(long engine, byte[] tenant, IntBuffer outIds);

    /** IRetainTopicIndex.add(tenantId, topic, timestamp, expirySeconds) / remove: ops[i] 0 = add, 1 = remove. */
    static native void retainApplyEx(long engine, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, ByteBuffer ops,
                                     LongBuffer timestampHlc, IntBuffer expirySeconds, int n);

    /** out = {timestampHlc, expirySeconds, expireAtMs} */
    static native void retainTopicInfo(long engine, int topicId, long[] out);

    /** out = {number of topics (the ids are 0 .. n-1), retain epoch} */
    static native void retainFindAll(long engine, long[] out);

    static native long retainMatchLimited(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants,
                                          IntBuffer filterTenant, ByteBuffer filters, IntBuffer filterOff, int nFilters,
                                          IntBuffer limits, long nowMs, IntBuffer outRowPtr, IntBuffer outTopicIds,
                                          IntBuffer outCounts);

    static native int retainTopic(long engine, int topicId, ByteBuffer out, long[] tenantLenOut);

    // ---- range pruning / routers ----
    /** TenantRangeLookupCache.lookup for a batch of topics of one tenant: outKeep[t * nCand + c] = 1 if candidate range c is kept. */
    static native void rangeLookup(long engine, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, int nTopics, ByteBuffer candKind,
                                   ByteBuffer first, IntBuffer firstOff, ByteBuffer last, IntBuffer lastOff, int nCand, ByteBuffer outKeep);

    /** MatchCallRangeRouter.rangeLookup: boundaries in BoundaryUtil.compare order; mode 0 = reference rules, 1 = exact. */
    static native void retainRangeLookup(byte[] tenant, ByteBuffer filters, IntBuffer filterOff, int nFilters, ByteBuffer rangeFlags,
                                         ByteBuffer start, IntBuffer startOff, ByteBuffer end, IntBuffer endOff, int nRanges, int mode,
                                         ByteBuffer outKeep);

    /** The effective router as an object (built when the range landscape changes, asked per request). */
    static native long routerCreate(ByteBuffer rangeFlags, ByteBuffer start, IntBuffer startOff, ByteBuffer end, IntBuffer endOff, int nRanges);

    static native void routerDestroy(long router);

    static native void routerRetainLookup(long router, byte[] tenant, ByteBuffer filters, IntBuffer filterOff, int nFilters, int mode,
                                          ByteBuffer outKeep);

    // ---- fan-out grouping ----
    /** (topic, route) pairs of a match batch regrouped by DelivererKey; out = {nGroups, special}. @return pairs, or -(groups needed) */
    static native long fanoutGroup(long engine, IntBuffer rowPtr, IntBuffer routeIds, int nTopics, IntBuffer outTopic, IntBuffer outRoute,
                                   IntBuffer outGroupOff, IntBuffer outGroupRep, long[] out);

    // ---- route cache (ISubscriptionCache on the engine's side) ----
    static native long routeCacheCreate(long engine, long batcher, long maxRoutesPerTenant, long expiryMs);

    static native void routeCacheDestroy(long cache);

    /** ISubscriptionCache.get: @return number of route ids, or -(needed); epochOut[0] = the engine epoch they were matched at */
    static native long routeCacheGet(long cache, byte[] tenant, byte[] topic, long nowMs, IntBuffer outIds, long[] epochOut);

    /**
     * A whole BatchDistRequest (DistWorkerCoProc.batchDist): cached rows are copied from the host, every miss of the request travels in
     * ONE launch.  outHit[i] = 1 if row i came from the cache.  @return number of ids, or -(needed)
     */
    static native long routeCacheGetBatch(long cache, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant,
                                          ByteBuffer topics, IntBuffer topicOff, int nTopics, long nowMs, IntBuffer outRowPtr,
                                          IntBuffer outIds, ByteBuffer outHit);

    /** Completed by the native side: on the calling thread for a hit, on the batching front's dispatcher thread for a miss. */
    interface RouteCallback {
        void onRoutes(int status, int[] routeIds, long epoch);
    }

    /** ISubscriptionCache.get as a future: nobody blocks on the GPU. */
    static native void routeCacheGetAsync(long cache, byte[] tenant, byte[] topic, long nowMs, RouteCallback cb);

    static native int routeCacheIsCached(long cache, byte[] tenant, byte[] filter);

    static native void routeCacheApply(long cache, ByteBuffer keys, IntBuffer keyOff, ByteBuffer ops, int n);

    static native void routeCacheRebuild(long cache, ByteBuffer keys, IntBuffer keyOff, int n);

    static native void routeCacheReset(long cache);

    /**
     * Drops the entries idle for expiryMs (what Caffeine's scheduler thread does) and the whole cache of every tenant nobody asked
     * about for 2 x expiryMs (SubscriptionCache.java:79-107); call it from a timer. @return entries dropped
     */
    static native long routeCacheExpire(long cache, long nowMs);

    /** The tenant's MaxPersistentFanout / MaxGroupFanout (ISettingProvider): rows are capped natively, in KV key order. */
    static native void routeCacheSetCaps(long cache, byte[] tenant, int maxPersistentFanout, int maxGroupFanout);

    /** IEventCollector.report for the routes a load throws away: type 0 = PersistentFanoutThrottled, 1 = GroupFanoutThrottled. */
    interface ThrottleSink {
        void onThrottle(byte[] tenant, byte[] topic, int type, int routeId, int maxCount);
    }

    static native void routeCacheSetEventSink(long cache, ThrottleSink sink);

    /** out[0..4] = hits, misses, evictions, entries, cached routes of the tenant (TenantRouteCache.java:141-147); false: no cache. */
    static native boolean routeCacheTenantStats(long cache, byte[] tenant, long[] out);
}
