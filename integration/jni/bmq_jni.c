/* bmq_jni.c -- the JNI binding a bifromq maintainer adds to put libbmq.so behind the dist worker's match path
 * (INTEGRATION.md).  Java side: integration/java/org/apache/bifromq/dist/worker/gpu/NativeMatcher.java.
 *
 * Conventions: every buffer is a DIRECT ByteBuffer / IntBuffer in native byte order (zero copy, GetDirectBufferAddress);
 * tenant ids travel as byte[] (short, copied).  A method returning "long" gives the number of ids written, or -(needed) when
 * the output buffer is too small (the caller grows it and calls again); any other failure throws IllegalStateException with
 * bmq_last_error().  No JDK exists in this repository's build image: the file is compile-checked against jni_min.h. */
#ifdef BMQ_REAL_JNI
#include <jni.h>
#else
#include "jni_min.h"
#endif
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "bmq.h"

#define ENGINE(h) ((bmq_engine*)(intptr_t)(h))
#define BATCHER(h) ((bmq_batcher*)(intptr_t)(h))
#define ADDR(o) ((o) ? (*env)->GetDirectBufferAddress(env, (o)) : NULL)
/* GetDirectBufferCapacity counts ELEMENTS of the buffer's type: bytes for a ByteBuffer, ints for an IntBuffer */
#define CAP(o) ((o) ? (uint64_t)(*env)->GetDirectBufferCapacity(env, (o)) : 0u)
#define NM(name) Java_org_apache_bifromq_dist_worker_gpu_NativeMatcher_##name

static void throw_state(JNIEnv* env, bmq_engine* e, const char* what, int rc) {
    char msg[512];
    snprintf(msg, sizeof msg, "%s failed: %d %s", what, rc, e ? bmq_last_error(e) : "");
    jclass cls = (*env)->FindClass(env, "java/lang/IllegalStateException");
    if (cls) (*env)->ThrowNew(env, cls, msg);
}
/* NOSPACE -> -(needed); other errors -> exception, 0 */
static jlong result_of(JNIEnv* env, bmq_engine* e, const char* what, int rc, uint64_t need) {
    if (rc == BMQ_E_NOSPACE) return -(jlong)need;
    if (rc != BMQ_OK) {
        throw_state(env, e, what, rc);
        return 0;
    }
    return (jlong)need;
}

/* long create(int device) */
JNIEXPORT jlong JNICALL NM(create)(JNIEnv* env, jclass c, jint device) {
    (void)c;
    bmq_config cfg = {0};
    cfg.struct_size = sizeof cfg;
    cfg.device = device;
    bmq_engine* e = NULL;
    const int rc = bmq_engine_create(&cfg, &e);
    if (rc != BMQ_OK) {
        throw_state(env, NULL, "bmq_engine_create (a gfx950 device is required)", rc);
        return 0;
    }
    return (jlong)(intptr_t)e;
}
/* long createOrdered(int device, int dedupMinTopics)   -- bmq_config.dedup_sorted: the caller's batches are ordered by (tenant, topic), as the
 * packs of a BatchDistRequest are (DistWorkerCoProc.proto:75-83); batches of at least dedupMinTopics rows are reduced to their distinct rows by
 * comparing neighbours, every row keeps its own row in the result */
JNIEXPORT jlong JNICALL NM(createOrdered)(JNIEnv* env, jclass c, jint device, jint dedupMinTopics) {
    (void)c;
    bmq_config cfg = {0};
    cfg.struct_size = sizeof cfg;
    cfg.device = device;
    cfg.dedup_min_topics = dedupMinTopics > 0 ? (uint32_t)dedupMinTopics : 1u;
    cfg.dedup_sorted = 1;
    bmq_engine* e = NULL;
    const int rc = bmq_engine_create(&cfg, &e);
    if (rc != BMQ_OK) {
        throw_state(env, NULL, "bmq_engine_create (a gfx950 device is required)", rc);
        return 0;
    }
    return (jlong)(intptr_t)e;
}
/* long createCompact(int device, int regionSlack)   -- bmq_config.region_slack: memory against speed.  The filter trie's per-tenant regions hold
 * nodes x (1 + regionSlack / 4) buckets; 0 = the default (6: load factor 0.2), 1 = half the region memory for ~8 % more time in the match kernel */
JNIEXPORT jlong JNICALL NM(createCompact)(JNIEnv* env, jclass c, jint device, jint regionSlack) {
    (void)c;
    bmq_config cfg = {0};
    cfg.struct_size = sizeof cfg;
    cfg.device = device;
    cfg.region_slack = regionSlack > 0 ? (uint32_t)regionSlack : 0u;
    bmq_engine* e = NULL;
    const int rc = bmq_engine_create(&cfg, &e);
    if (rc != BMQ_OK) {
        throw_state(env, NULL, "bmq_engine_create (a gfx950 device is required; regionSlack <= 64)", rc);
        return 0;
    }
    return (jlong)(intptr_t)e;
}
/* void destroy(long engine) */
JNIEXPORT void JNICALL NM(destroy)(JNIEnv* env, jclass c, jlong h) {
    (void)env, (void)c;
    bmq_engine_destroy(ENGINE(h));
}

/* void rebuild(long engine, ByteBuffer keys, IntBuffer keyOff, int n)      <- IKVRangeCoProc.reset(): reader.iterator() keys */
JNIEXPORT void JNICALL NM(rebuild)(JNIEnv* env, jclass c, jlong h, jobject keys, jobject keyOff, jint n) {
    (void)c;
    const int rc = bmq_rebuild(ENGINE(h), (const uint8_t*)ADDR(keys), (const uint32_t*)ADDR(keyOff), (uint32_t)n);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_rebuild", rc);
}
/* void routesApply(long engine, ByteBuffer keys, IntBuffer keyOff, ByteBuffer ops, int n)   <- post-commit Add/RemoveRoutesTask */
JNIEXPORT void JNICALL NM(routesApply)(JNIEnv* env, jclass c, jlong h, jobject keys, jobject keyOff, jobject ops, jint n) {
    (void)c;
    const int rc = bmq_routes_apply(ENGINE(h), (const uint8_t*)ADDR(keys), (const uint32_t*)ADDR(keyOff), (const uint8_t*)ADDR(ops), (uint32_t)n);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_routes_apply", rc);
}
/* void routesApplyAsync(long engine, ByteBuffer keys, IntBuffer keyOff, ByteBuffer ops, int n) / void routesApplyWait(long engine)
 * bmq_routes_apply_async / _wait: the ops are uploaded beside the batch in flight and applied behind it; the call returns after the enqueue.
 * The three DIRECT buffers (page-locked ones from hostAlloc make the upload a DMA) must stay alive and unchanged until routesApplyWait --
 * or any other call that reads or changes the route index -- has returned; that call throws what routesApply would have thrown. */
JNIEXPORT void JNICALL NM(routesApplyAsync)(JNIEnv* env, jclass c, jlong h, jobject keys, jobject keyOff, jobject ops, jint n) {
    (void)c;
    const int rc = bmq_routes_apply_async(ENGINE(h), (const uint8_t*)ADDR(keys), (const uint32_t*)ADDR(keyOff), (const uint8_t*)ADDR(ops), (uint32_t)n);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_routes_apply_async", rc);
}
JNIEXPORT void JNICALL NM(routesApplyWait)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    const int rc = bmq_routes_apply_wait(ENGINE(h));
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_routes_apply_wait", rc);
}
/* long epoch(long engine) */
JNIEXPORT jlong JNICALL NM(epoch)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    bmq_index_info info;
    const int rc = bmq_index_info_get(ENGINE(h), &info);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_index_info_get", rc);
        return 0;
    }
    return (jlong)info.epoch;
}
/* long generation(long engine)     ids of different generations are unrelated (every rebuild re-numbers) */
JNIEXPORT jlong JNICALL NM(generation)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    bmq_index_info info;
    const int rc = bmq_index_info_get(ENGINE(h), &info);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_index_info_get", rc);
        return 0;
    }
    return (jlong)info.generation;
}
/* long routeKeys(long engine, IntBuffer ids, int n, ByteBuffer out, LongBuffer outOff)   one device gather for many ids;
 * outOff[n + 1] byte offsets into out, a dead id (unsubscribed meanwhile) gives an empty key; -> bytes, or -(needed) */
JNIEXPORT jlong JNICALL NM(routeKeys)(JNIEnv* env, jclass c, jlong h, jobject ids, jint n, jobject out, jobject outOff) {
    (void)c;
    uint64_t* off = (uint64_t*)ADDR(outOff);
    const int rc = bmq_route_keys(ENGINE(h), (const uint32_t*)ADDR(ids), (uint32_t)n, (uint8_t*)ADDR(out), CAP(out), off);
    return result_of(env, ENGINE(h), "bmq_route_keys", rc, off ? off[n] : 0);
}
/* ByteBuffer hostAlloc(long bytes)      page-locked memory as a direct buffer: full-speed, truly asynchronous PCIe transfers */
JNIEXPORT jobject JNICALL NM(hostAlloc)(JNIEnv* env, jclass c, jlong bytes) {
    (void)c;
    void* p = bmq_host_alloc((size_t)bytes);
    return p ? (*env)->NewDirectByteBuffer(env, p, bytes) : NULL;
}
/* void hostFree(ByteBuffer buf) */
JNIEXPORT void JNICALL NM(hostFree)(JNIEnv* env, jclass c, jobject buf) {
    (void)c;
    bmq_host_free(ADDR(buf));
}
/* int matchSubmit(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant, ByteBuffer topics,
 *                 IntBuffer topicOff, int nTopics)      -> ticket (0 .. BMQ_MAX_TICKETS - 1): three batches may be in flight */
JNIEXPORT jint JNICALL NM(matchSubmit)(JNIEnv* env, jclass c, jlong h, jobject tenants, jobject tenantOff, jint nTenants, jobject topicTenant,
                                       jobject topics, jobject topicOff, jint nTopics) {
    (void)c;
    int ticket = -1;
    const int rc = bmq_match_submit(ENGINE(h), (const uint8_t*)ADDR(tenants), (const uint32_t*)ADDR(tenantOff), (uint32_t)nTenants,
                                    (const uint32_t*)ADDR(topicTenant), (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                    (uint32_t)nTopics, &ticket);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_match_submit", rc);
    return ticket;
}
/* long matchWait(long engine, int ticket, IntBuffer outRowPtr, IntBuffer outIds)     -> number of ids, or -(needed) */
JNIEXPORT jlong JNICALL NM(matchWait)(JNIEnv* env, jclass c, jlong h, jint ticket, jobject outRowPtr, jobject outIds) {
    (void)c;
    uint64_t need = 0;
    const int rc = bmq_match_wait(ENGINE(h), ticket, (uint32_t*)ADDR(outRowPtr), (uint32_t*)ADDR(outIds), CAP(outIds), &need);
    return result_of(env, ENGINE(h), "bmq_match_wait", rc, need);
}
/* ---- result formats that fit the wire (include/bmq.h BMQ_FMT_*) ---- */
/* int matchSubmitFmt(long engine, ..., int nTopics, int format)   -> ticket (0 .. BMQ_MAX_TICKETS - 1) */
JNIEXPORT jint JNICALL NM(matchSubmitFmt)(JNIEnv* env, jclass c, jlong h, jobject tenants, jobject tenantOff, jint nTenants, jobject topicTenant,
                                          jobject topics, jobject topicOff, jint nTopics, jint format) {
    (void)c;
    int ticket = -1;
    const int rc = bmq_match_submit_fmt(ENGINE(h), (const uint8_t*)ADDR(tenants), (const uint32_t*)ADDR(tenantOff), (uint32_t)nTenants,
                                        (const uint32_t*)ADDR(topicTenant), (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                        (uint32_t)nTopics, (int)format, &ticket);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_match_submit_fmt", rc);
    return ticket;
}
/* long matchWaitCounts(long engine, int ticket, IntBuffer outRowPtr)   -> total fan-out of the batch; fan-out of topic i =
 * outRowPtr[i + 1] - outRowPtr[i]: all a BatchDistReply carries (DW/DistWorkerCoProc.java:535-538) */
JNIEXPORT jlong JNICALL NM(matchWaitCounts)(JNIEnv* env, jclass c, jlong h, jint ticket, jobject outRowPtr) {
    (void)c;
    uint64_t total = 0;
    const int rc = bmq_match_wait_counts(ENGINE(h), ticket, (uint32_t*)ADDR(outRowPtr), &total);
    return result_of(env, ENGINE(h), "bmq_match_wait_counts", rc, total);
}
/* long matchWaitRanges(long engine, int ticket, IntBuffer outRowPtr (may be null), IntBuffer outRangePtr, IntBuffer outRanges (begin, count pairs),
 *                      IntBuffer outSideIds, long[] info4)   -> number of ranges, or -(needed ranges) when a buffer is too small;
 * info4 = {ranges, side ids, ids of the batch, rows whose expanded ids the consumer must order} */
JNIEXPORT jlong JNICALL NM(matchWaitRanges)(JNIEnv* env, jclass c, jlong h, jint ticket, jobject outRowPtr, jobject outRangePtr, jobject outRanges,
                                            jobject outSideIds, jlongArray info4) {
    (void)c;
    bmq_ranges_info info;
    const int rc = bmq_match_wait_ranges(ENGINE(h), ticket, outRowPtr ? (uint32_t*)ADDR(outRowPtr) : NULL, (uint32_t*)ADDR(outRangePtr),
                                         (bmq_id_range*)ADDR(outRanges), CAP(outRanges) / 2, (uint32_t*)ADDR(outSideIds), CAP(outSideIds), &info);
    if (rc == BMQ_OK || rc == BMQ_E_NOSPACE) {
        const jlong v[4] = {(jlong)info.n_ranges, (jlong)info.n_side_ids, (jlong)info.n_ids, (jlong)info.n_overlapping_rows};
        (*env)->SetLongArrayRegion(env, info4, 0, 4, v);
    }
    return result_of(env, ENGINE(h), "bmq_match_wait_ranges", rc, info.n_ranges);
}
/* long matchWaitGrouped(long engine, int ticket, IntBuffer outTopic, IntBuffer outRoute, IntBuffer outGroupOff, IntBuffer outGroupRep, int[] groups2)
 *   -> number of (topic, route) pairs, or -(needed); groups2 = {groups, special bits}: what DeliverExecutorGroup.submit builds next */
JNIEXPORT jlong JNICALL NM(matchWaitGrouped)(JNIEnv* env, jclass c, jlong h, jint ticket, jobject outTopic, jobject outRoute, jobject outGroupOff,
                                             jobject outGroupRep, jintArray groups2) {
    (void)c;
    uint32_t ng = 0, special = 0;
    uint64_t total = 0;
    const int rc = bmq_match_wait_grouped(ENGINE(h), ticket, (uint32_t*)ADDR(outTopic), (uint32_t*)ADDR(outRoute), CAP(outTopic), (uint32_t*)ADDR(outGroupOff),
                                          (uint32_t*)ADDR(outGroupRep), (uint32_t)CAP(outGroupRep), &ng, &special, &total);
    const jint g[2] = {(jint)ng, (jint)special};
    (*env)->SetIntArrayRegion(env, groups2, 0, 2, g);
    return result_of(env, ENGINE(h), "bmq_match_wait_grouped", rc, total);
}
/* int routeKey(long engine, int routeId, ByteBuffer out)    -> key length; the adapter turns it into a Matching
 *                                                              (KVSchemaUtil.buildMatchRoute(routeKey, value)) */
JNIEXPORT jint JNICALL NM(routeKey)(JNIEnv* env, jclass c, jlong h, jint id, jobject out) {
    (void)c;
    uint32_t len = 0;
    const int rc = bmq_route_key(ENGINE(h), (uint32_t)id, (uint8_t*)ADDR(out), (uint32_t)CAP(out), &len);
    if (rc == BMQ_E_NOSPACE) return -(jint)len;
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_route_key", rc);
        return 0;
    }
    return (jint)len;
}

/* long matchBatch(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant,
 *                 ByteBuffer topics, IntBuffer topicOff, int nTopics, IntBuffer outRowPtr, IntBuffer outIds)
 * one call per BatchDistRequest: all DistPacks (tenants) x all cache-missing topics */
JNIEXPORT jlong JNICALL NM(matchBatch)(JNIEnv* env, jclass c, jlong h, jobject tenants, jobject tenantOff, jint nTenants,
                                       jobject topicTenant, jobject topics, jobject topicOff, jint nTopics, jobject outRowPtr,
                                       jobject outIds) {
    (void)c;
    uint64_t need = 0;
    const int rc = bmq_match_batch(ENGINE(h), (const uint8_t*)ADDR(tenants), (const uint32_t*)ADDR(tenantOff), (uint32_t)nTenants,
                                   (const uint32_t*)ADDR(topicTenant), (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                   (uint32_t)nTopics, (uint32_t*)ADDR(outRowPtr), (uint32_t*)ADDR(outIds), CAP(outIds), &need);
    return result_of(env, ENGINE(h), "bmq_match_batch", rc, need);
}

/* long matchAll(long engine, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, int nTopics, int maxPersistentFanout,
 *               int maxGroupFanout, IntBuffer outRowPtr, IntBuffer outIds, IntBuffer outEvents (4 ints each), long[] nEventsOut)
 * ITenantRouteMatcher.matchAll with the MatchedRoutes caps applied natively; events: see bmq_match_all */
JNIEXPORT jlong JNICALL NM(matchAll)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jobject topics, jobject topicOff, jint nTopics,
                                     jint maxPF, jint maxGF, jobject outRowPtr, jobject outIds, jobject outEvents, jlongArray nEventsOut) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    uint64_t need = 0;
    uint32_t n_events = 0;
    const int rc = bmq_match_all(ENGINE(h), (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                 (uint32_t)nTopics, maxPF, maxGF, (uint32_t*)ADDR(outRowPtr), (uint32_t*)ADDR(outIds), CAP(outIds), &need,
                                 (int32_t*)ADDR(outEvents), (uint32_t)(CAP(outEvents) / 4), &n_events);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    const jlong ne = (jlong)n_events;
    (*env)->SetLongArrayRegion(env, nEventsOut, 0, 1, &ne);
    return result_of(env, ENGINE(h), "bmq_match_all", rc, need);
}

/* ---- batching front: the call shape TenantRouteCache already has (one matchAll per cache miss, many threads) ---- */
/* long batcherCreate(long engine, int maxBatchTopics) */
JNIEXPORT jlong JNICALL NM(batcherCreate)(JNIEnv* env, jclass c, jlong h, jint maxBatchTopics) {
    (void)c;
    bmq_batcher_config cfg = {0};
    cfg.struct_size = sizeof cfg;
    cfg.max_batch_topics = (uint32_t)maxBatchTopics;
    bmq_batcher* b = NULL;
    const int rc = bmq_batcher_create(ENGINE(h), &cfg, &b);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_batcher_create", rc);
        return 0;
    }
    return (jlong)(intptr_t)b;
}
/* void batcherDestroy(long batcher) */
JNIEXPORT void JNICALL NM(batcherDestroy)(JNIEnv* env, jclass c, jlong b) {
    (void)env, (void)c;
    bmq_batcher_destroy(BATCHER(b));
}
/* long batcherMatchAll(long batcher, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, int nTopics, IntBuffer outRowPtr,
 *                      IntBuffer outIds, long[] epochOut)      blocks until the launch carrying these topics has finished */
JNIEXPORT jlong JNICALL NM(batcherMatchAll)(JNIEnv* env, jclass c, jlong b, jbyteArray tenant, jobject topics, jobject topicOff, jint nTopics,
                                            jobject outRowPtr, jobject outIds, jlongArray epochOut) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    uint64_t need = 0, epoch = 0;
    const int rc = bmq_batcher_match_all(BATCHER(b), (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                         (uint32_t)nTopics, (uint32_t*)ADDR(outRowPtr), (uint32_t*)ADDR(outIds), CAP(outIds), &need, &epoch);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    const jlong ep = (jlong)epoch;
    (*env)->SetLongArrayRegion(env, epochOut, 0, 1, &ep);
    return result_of(env, NULL, "bmq_batcher_match_all", rc, need);
}

/* ---- retain direction (IRetainTopicIndex) ---- */
/* void retainRebuild(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant, ByteBuffer topics,
 *                    IntBuffer topicOff, int nTopics) */
JNIEXPORT void JNICALL NM(retainRebuild)(JNIEnv* env, jclass c, jlong h, jobject tenants, jobject tenantOff, jint nTenants, jobject topicTenant,
                                         jobject topics, jobject topicOff, jint nTopics) {
    (void)c;
    const int rc = bmq_retain_rebuild(ENGINE(h), (const uint8_t*)ADDR(tenants), (const uint32_t*)ADDR(tenantOff), (uint32_t)nTenants,
                                      (const uint32_t*)ADDR(topicTenant), (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff), (uint32_t)nTopics);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_retain_rebuild", rc);
}
/* void retainApply(long engine, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, ByteBuffer ops, int n)   add = 0 / remove = 1 */
JNIEXPORT void JNICALL NM(retainApply)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jobject topics, jobject topicOff, jobject ops, jint n) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    const int rc = bmq_retain_apply(ENGINE(h), (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                    (const uint8_t*)ADDR(ops), (uint32_t)n);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_retain_apply", rc);
}
/* void retainApplyEx(long engine, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, ByteBuffer ops, LongBuffer timestampHlc,
 *                    IntBuffer expirySeconds, int n)        IRetainTopicIndex.add(tenant, topic, timestamp, expirySeconds) / remove */
JNIEXPORT void JNICALL NM(retainApplyEx)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jobject topics, jobject topicOff, jobject ops,
                                         jobject ts, jobject expiry, jint n) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    const int rc = bmq_retain_apply_ex(ENGINE(h), (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                       (const uint8_t*)ADDR(ops), (const uint64_t*)ADDR(ts), (const uint32_t*)ADDR(expiry), (uint32_t)n);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_retain_apply_ex", rc);
}
/* void retainApplyBatch(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer opTenant, ByteBuffer topics, IntBuffer topicOff,
 *                       ByteBuffer ops, LongBuffer timestampHlc, IntBuffer expirySeconds, int n, IntBuffer outTopicIds (may be null))
 * the adds / removes of one pass of the coproc's apply loop (RS/RetainStoreCoProc.java:240-255), all tenants, as kernels behind the batches in flight */
JNIEXPORT void JNICALL NM(retainApplyBatch)(JNIEnv* env, jclass c, jlong h, jobject tenants, jobject tenantOff, jint nTenants, jobject opTenant,
                                            jobject topics, jobject topicOff, jobject ops, jobject timestampHlc, jobject expirySeconds, jint n,
                                            jobject outTopicIds) {
    (void)c;
    const int rc = bmq_retain_apply_batch(ENGINE(h), (const uint8_t*)ADDR(tenants), (const uint32_t*)ADDR(tenantOff), (uint32_t)nTenants,
                                          (const uint32_t*)ADDR(opTenant), (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                          (const uint8_t*)ADDR(ops), (const uint64_t*)ADDR(timestampHlc), (const uint32_t*)ADDR(expirySeconds), (uint32_t)n,
                                          outTopicIds ? (uint32_t*)ADDR(outTopicIds) : NULL);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_retain_apply_batch", rc);
}
/* void retainCompact(long engine)     maintenance: a fresh bulk load of the live topics (a new generation of ids) */
JNIEXPORT void JNICALL NM(retainCompact)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    const int rc = bmq_retain_compact(ENGINE(h));
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_retain_compact", rc);
}
/* void retainInfo(long engine, long[] out9)   out = bmq_retain_info {topics, tenants, idBound, loaded, loadedRemoved, addedIds, overlayNodes, epoch, generation} */
JNIEXPORT void JNICALL NM(retainInfo)(JNIEnv* env, jclass c, jlong h, jlongArray out) {
    (void)c;
    bmq_retain_info ri;
    const int rc = bmq_retain_info_get(ENGINE(h), &ri);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_retain_info_get", rc);
        return;
    }
    const jlong v[9] = {(jlong)ri.n_topics, (jlong)ri.n_tenants, (jlong)ri.id_bound, (jlong)ri.loaded_topics, (jlong)ri.loaded_removed,
                        (jlong)ri.added_ids, (jlong)ri.overlay_nodes, (jlong)ri.epoch, (jlong)ri.generation};
    (*env)->SetLongArrayRegion(env, out, 0, 9, v);
}
/* Compaction without a stall, inside ONE handle (include/bmq.h: bmq_compact_begin / _poll / _swap / _abort): the next generation is built beside the
 * serving one, on its own stream, from keys that never leave HBM; the adapter paces it with compactPoll from a maintenance thread while its
 * matcher threads go on, and re-creates what belongs to a generation (id -> Matching cache, route cache) after compactSwap.
 * void compactBegin(long engine) / int compactPoll(long engine, int maxIds) -> progress in permille / void compactSwap(long engine, long[] out2
 * {keys carried over, ops replayed}) / void compactAbort(long engine) */
JNIEXPORT void JNICALL NM(compactBegin)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    const int rc = bmq_compact_begin(ENGINE(h));
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_compact_begin", rc);
}
JNIEXPORT jint JNICALL NM(compactPoll)(JNIEnv* env, jclass c, jlong h, jint maxIds) {
    (void)c;
    uint32_t done = 0;
    const int rc = bmq_compact_poll(ENGINE(h), (uint32_t)maxIds, &done);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_compact_poll", rc);
    return (jint)done;
}
JNIEXPORT void JNICALL NM(compactSwap)(JNIEnv* env, jclass c, jlong h, jlongArray out) {
    (void)c;
    uint64_t carried = 0, replayed = 0;
    const int rc = bmq_compact_swap(ENGINE(h), &carried, &replayed);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_compact_swap", rc);
        return;
    }
    const jlong v[2] = {(jlong)carried, (jlong)replayed};
    if (out) (*env)->SetLongArrayRegion(env, out, 0, 2, v);
}
JNIEXPORT void JNICALL NM(compactAbort)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    const int rc = bmq_compact_abort(ENGINE(h));
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_compact_abort", rc);
}
/* The retained-topic index's generation change without the stall (round 6; include/bmq.h: bmq_retain_compact_begin / _build / _swap / _abort):
 * retainCompactBegin from the maintenance thread (snapshot, tens of ms), retainCompactBuild (the load: seconds, no engine lock held),
 * retainCompactSwap(out2 {topics carried over, ops replayed}) -- topic ids are re-numbered: the adapter's id -> topic cache is dropped. */
JNIEXPORT void JNICALL NM(retainCompactBegin)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    const int rc = bmq_retain_compact_begin(ENGINE(h));
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_retain_compact_begin", rc);
}
JNIEXPORT void JNICALL NM(retainCompactBuild)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    const int rc = bmq_retain_compact_build(ENGINE(h));
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_retain_compact_build", rc);
}
JNIEXPORT void JNICALL NM(retainCompactSwap)(JNIEnv* env, jclass c, jlong h, jlongArray out) {
    (void)c;
    uint64_t carried = 0, replayed = 0;
    const int rc = bmq_retain_compact_swap(ENGINE(h), &carried, &replayed);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_retain_compact_swap", rc);
        return;
    }
    const jlong v[2] = {(jlong)carried, (jlong)replayed};
    if (out) (*env)->SetLongArrayRegion(env, out, 0, 2, v);
}
JNIEXPORT void JNICALL NM(retainCompactAbort)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    const int rc = bmq_retain_compact_abort(ENGINE(h));
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_retain_compact_abort", rc);
}
/* The persistent matcher behind the batching front (round 6; bmq_poller_*): nothing to call for it to work -- batcherMatchAll / routeCacheGet
 * use it --; pollerStats(out8 {enabled, running, starts, served, fallback, unserved, timeouts, badInput}) feeds the broker's meters,
 * pollerControl(what) switches it (0 off, 1 on, 2 leave now). */
JNIEXPORT void JNICALL NM(pollerStats)(JNIEnv* env, jclass c, jlong h, jlongArray out) {
    (void)c;
    bmq_poller_stats ps;
    const int rc = bmq_poller_stats_get(ENGINE(h), &ps);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_poller_stats_get", rc);
        return;
    }
    const jlong v[8] = {(jlong)ps.enabled, (jlong)ps.running, (jlong)ps.n_starts, (jlong)ps.n_served, (jlong)ps.n_fallback, (jlong)ps.n_unserved,
                        (jlong)ps.n_timeouts, (jlong)ps.n_bad_input};
    if (out) (*env)->SetLongArrayRegion(env, out, 0, 8, v);
}
JNIEXPORT void JNICALL NM(pollerControl)(JNIEnv* env, jclass c, jlong h, jint what) {
    (void)c;
    const int rc = bmq_poller_control(ENGINE(h), (int)what);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_poller_control", rc);
}
/* void indexInfo(long engine, long[] out11)   out = bmq_index_info {routes, tenants, nodes, tokens, trieSlots, dictSlots, deviceBytes, epoch, generation,
 * nextRouteId, garbageBytes} -- nextRouteId bounds the ids GenerationalRangeIndex exports, garbageBytes tells it when a compaction pays */
JNIEXPORT void JNICALL NM(indexInfo)(JNIEnv* env, jclass c, jlong h, jlongArray out) {
    (void)c;
    bmq_index_info ii;
    const int rc = bmq_index_info_get(ENGINE(h), &ii);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_index_info_get", rc);
        return;
    }
    const jlong v[11] = {(jlong)ii.n_routes,     (jlong)ii.n_tenants,    (jlong)ii.n_nodes, (jlong)ii.n_tokens,    (jlong)ii.trie_slots,   (jlong)ii.dict_slots,
                         (jlong)ii.device_bytes, (jlong)ii.epoch,        (jlong)ii.generation, (jlong)ii.next_route_id, (jlong)ii.garbage_bytes};
    (*env)->SetLongArrayRegion(env, out, 0, 11, v);
}
/* long retainLiveIds(long engine, byte[] tenant (null: all tenants), IntBuffer outIds)    -> number of retained topics, or -(needed) */
JNIEXPORT jlong JNICALL NM(retainLiveIds)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jobject outIds) {
    (void)c;
    jsize tl = tenant ? (*env)->GetArrayLength(env, tenant) : 0;
    jbyte* tn = tenant ? (*env)->GetByteArrayElements(env, tenant, NULL) : NULL;
    uint32_t n = 0;
    const int rc = bmq_retain_live_ids(ENGINE(h), (const uint8_t*)tn, (uint32_t)tl, (uint32_t*)ADDR(outIds), (uint32_t)CAP(outIds), &n);
    if (tenant) (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    return result_of(env, ENGINE(h), "bmq_retain_live_ids", rc, n);
}
/* void retainTopicInfo(long engine, int topicId, long[] out)     out = {timestampHlc, expirySeconds, expireAtMs} */
JNIEXPORT void JNICALL NM(retainTopicInfo)(JNIEnv* env, jclass c, jlong h, jint id, jlongArray out) {
    (void)c;
    uint64_t ts = 0, at = 0;
    uint32_t ex = 0;
    const int rc = bmq_retain_topic_info(ENGINE(h), (uint32_t)id, &ts, &ex, &at);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_retain_topic_info", rc);
        return;
    }
    const jlong v[3] = {(jlong)ts, (jlong)ex, (jlong)at};
    (*env)->SetLongArrayRegion(env, out, 0, 3, v);
}
/* void retainFindAll(long engine, long[] out)      out = {number of retained topics (their ids: retainLiveIds), retain epoch} */
JNIEXPORT void JNICALL NM(retainFindAll)(JNIEnv* env, jclass c, jlong h, jlongArray out) {
    (void)c;
    uint64_t n = 0, ep = 0;
    const int rc = bmq_retain_find_all(ENGINE(h), &n, &ep);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_retain_find_all", rc);
        return;
    }
    const jlong v[2] = {(jlong)n, (jlong)ep};
    (*env)->SetLongArrayRegion(env, out, 0, 2, v);
}
/* long retainMatchLimited(long engine, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer filterTenant, ByteBuffer filters,
 *                         IntBuffer filterOff, int nFilters, IntBuffer limits, long nowMs, IntBuffer outRowPtr, IntBuffer outTopicIds,
 *                         IntBuffer outCounts)
 * RetainStoreCoProc.match(limit, now) for one BatchMatchRequest: per filter the `limit` smallest topic ids that have not expired */
JNIEXPORT jlong JNICALL NM(retainMatchLimited)(JNIEnv* env, jclass c, jlong h, jobject tenants, jobject tenantOff, jint nTenants, jobject filterTenant,
                                               jobject filters, jobject filterOff, jint nFilters, jobject limits, jlong nowMs, jobject outRowPtr,
                                               jobject outTopicIds, jobject outCounts) {
    (void)c;
    uint64_t need = 0;
    const int rc = bmq_retain_match_limited(ENGINE(h), (const uint8_t*)ADDR(tenants), (const uint32_t*)ADDR(tenantOff), (uint32_t)nTenants,
                                            (const uint32_t*)ADDR(filterTenant), (const uint8_t*)ADDR(filters), (const uint32_t*)ADDR(filterOff),
                                            (uint32_t)nFilters, (const uint32_t*)ADDR(limits), (uint64_t)nowMs, (uint32_t*)ADDR(outRowPtr),
                                            (uint32_t*)ADDR(outTopicIds), CAP(outTopicIds), &need, (uint32_t*)ADDR(outCounts));
    return result_of(env, ENGINE(h), "bmq_retain_match_limited", rc, need);
}
/* int retainTopic(long engine, int topicId, ByteBuffer out, long[] tenantLenOut)   -> total length (tenant bytes then topic bytes) */
JNIEXPORT jint JNICALL NM(retainTopic)(JNIEnv* env, jclass c, jlong h, jint id, jobject out, jlongArray tenantLenOut) {
    (void)c;
    uint32_t len = 0, tl = 0;
    const int rc = bmq_retain_topic(ENGINE(h), (uint32_t)id, (uint8_t*)ADDR(out), (uint32_t)CAP(out), &len, &tl);
    const jlong t = (jlong)tl;
    (*env)->SetLongArrayRegion(env, tenantLenOut, 0, 1, &t);
    if (rc == BMQ_E_NOSPACE) return -(jint)len;
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_retain_topic", rc);
        return 0;
    }
    return (jint)len;
}

/* ---- range pruning / routers ---- */
/* void rangeLookup(long engine, byte[] tenant, ByteBuffer topics, IntBuffer topicOff, int nTopics, ByteBuffer candKind, ByteBuffer first,
 *                  IntBuffer firstOff, ByteBuffer last, IntBuffer lastOff, int nCand, ByteBuffer outKeep)
 * TenantRangeLookupCache.lookup for a batch of topics of one tenant: outKeep[t * nCand + c] */
JNIEXPORT void JNICALL NM(rangeLookup)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jobject topics, jobject topicOff, jint nTopics,
                                       jobject candKind, jobject first, jobject firstOff, jobject last, jobject lastOff, jint nCand, jobject outKeep) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    const int rc = bmq_range_lookup(ENGINE(h), (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                    (uint32_t)nTopics, (const uint8_t*)ADDR(candKind), (const uint8_t*)ADDR(first), (const uint32_t*)ADDR(firstOff),
                                    (const uint8_t*)ADDR(last), (const uint32_t*)ADDR(lastOff), (uint32_t)nCand, (uint8_t*)ADDR(outKeep));
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    if (rc != BMQ_OK) throw_state(env, ENGINE(h), "bmq_range_lookup", rc);
}
/* void retainRangeLookup(byte[] tenant, ByteBuffer filters, IntBuffer filterOff, int nFilters, ByteBuffer rangeFlags, ByteBuffer start,
 *                        IntBuffer startOff, ByteBuffer end, IntBuffer endOff, int nRanges, int mode, ByteBuffer outKeep)
 * MatchCallRangeRouter.rangeLookup over the effective router given as boundaries in BoundaryUtil.compare order; mode 0 = the
 * reference's findCandidates rules, 1 = exact; outKeep[f * nRanges + r] */
JNIEXPORT void JNICALL NM(retainRangeLookup)(JNIEnv* env, jclass c, jbyteArray tenant, jobject filters, jobject filterOff, jint nFilters,
                                             jobject rangeFlags, jobject start, jobject startOff, jobject end, jobject endOff, jint nRanges, jint mode,
                                             jobject outKeep) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    const int rc = bmq_retain_range_lookup((const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)ADDR(filters), (const uint32_t*)ADDR(filterOff),
                                           (uint32_t)nFilters, (const uint8_t*)ADDR(rangeFlags), (const uint8_t*)ADDR(start),
                                           (const uint32_t*)ADDR(startOff), (const uint8_t*)ADDR(end), (const uint32_t*)ADDR(endOff), (uint32_t)nRanges,
                                           (uint32_t)mode, (uint8_t*)ADDR(outKeep));
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    if (rc != BMQ_OK) throw_state(env, NULL, "bmq_retain_range_lookup", rc);
}

/* long routerCreate(ByteBuffer rangeFlags, ByteBuffer start, IntBuffer startOff, ByteBuffer end, IntBuffer endOff, int nRanges)
 * the effective router as an object: built when the range landscape changes (as the reference rebuilds its TreeMap), asked per request */
JNIEXPORT jlong JNICALL NM(routerCreate)(JNIEnv* env, jclass c, jobject rangeFlags, jobject start, jobject startOff, jobject end, jobject endOff,
                                         jint nRanges) {
    (void)c;
    bmq_router* r = NULL;
    const int rc = bmq_router_create((const uint8_t*)ADDR(rangeFlags), (const uint8_t*)ADDR(start), (const uint32_t*)ADDR(startOff),
                                     (const uint8_t*)ADDR(end), (const uint32_t*)ADDR(endOff), (uint32_t)nRanges, &r);
    if (rc != BMQ_OK) throw_state(env, NULL, "bmq_router_create", rc);
    return (jlong)(intptr_t)r;
}
JNIEXPORT void JNICALL NM(routerDestroy)(JNIEnv* env, jclass c, jlong router) {
    (void)env;
    (void)c;
    bmq_router_destroy((bmq_router*)(intptr_t)router);
}
/* void routerRetainLookup(long router, byte[] tenant, ByteBuffer filters, IntBuffer filterOff, int nFilters, int mode, ByteBuffer outKeep) */
JNIEXPORT void JNICALL NM(routerRetainLookup)(JNIEnv* env, jclass c, jlong router, jbyteArray tenant, jobject filters, jobject filterOff,
                                              jint nFilters, jint mode, jobject outKeep) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    const int rc = bmq_router_retain_lookup((const bmq_router*)(intptr_t)router, (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)ADDR(filters),
                                            (const uint32_t*)ADDR(filterOff), (uint32_t)nFilters, (uint32_t)mode, (uint8_t*)ADDR(outKeep));
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    if (rc != BMQ_OK) throw_state(env, NULL, "bmq_router_retain_lookup", rc);
}

/* ---- fan-out grouping ---- */
/* long fanoutGroup(long engine, IntBuffer rowPtr, IntBuffer routeIds, int nTopics, IntBuffer outTopic, IntBuffer outRoute,
 *                  IntBuffer outGroupOff, IntBuffer outGroupRep, long[] out)       out = {nGroups, special}
 * the (topic, route) pairs of a match batch regrouped by DelivererKey; -> pairs, or -(groups needed) when outGroupRep is too small */
JNIEXPORT jlong JNICALL NM(fanoutGroup)(JNIEnv* env, jclass c, jlong h, jobject rowPtr, jobject routeIds, jint nTopics, jobject outTopic,
                                        jobject outRoute, jobject outGroupOff, jobject outGroupRep, jlongArray out) {
    (void)c;
    uint32_t ng = 0, sp = 0;
    const uint32_t* row = (const uint32_t*)ADDR(rowPtr);
    const int rc = bmq_fanout_group(ENGINE(h), row, (const uint32_t*)ADDR(routeIds), (uint32_t)nTopics, (uint32_t*)ADDR(outTopic),
                                    (uint32_t*)ADDR(outRoute), CAP(outRoute), (uint32_t*)ADDR(outGroupOff), (uint32_t*)ADDR(outGroupRep),
                                    (uint32_t)CAP(outGroupRep), &ng, &sp);
    const jlong v[2] = {(jlong)ng, (jlong)sp};
    (*env)->SetLongArrayRegion(env, out, 0, 2, v);
    if (rc == BMQ_E_NOSPACE) return -(jlong)(ng ? ng : 1);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_fanout_group", rc);
        return 0;
    }
    return row ? (jlong)row[nTopics] : 0;
}

/* ---- route cache (ISubscriptionCache on the engine's side) ---- */
#define CACHE(h) ((bmq_route_cache*)(intptr_t)(h))
/* long routeCacheCreate(long engine, long batcher, long maxRoutesPerTenant, long expiryMs)     0 = the reference's defaults */
JNIEXPORT jlong JNICALL NM(routeCacheCreate)(JNIEnv* env, jclass c, jlong h, jlong b, jlong maxRoutes, jlong expiryMs) {
    (void)c;
    bmq_route_cache_config cfg = {0};
    cfg.struct_size = sizeof cfg;
    cfg.max_routes_per_tenant = (uint64_t)maxRoutes;
    cfg.expiry_ms = (uint64_t)expiryMs;
    bmq_route_cache* rc_ = NULL;
    const int rc = bmq_route_cache_create(ENGINE(h), BATCHER(b), &cfg, &rc_);
    if (rc != BMQ_OK) {
        throw_state(env, ENGINE(h), "bmq_route_cache_create", rc);
        return 0;
    }
    return (jlong)(intptr_t)rc_;
}
/* void routeCacheDestroy(long cache) */
static void sink_drop(JNIEnv* env, bmq_route_cache* cache);
JNIEXPORT void JNICALL NM(routeCacheDestroy)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    bmq_route_cache_destroy(CACHE(h)); /* no getter is inside any more (the caller's contract): nobody calls the sink from here on */
    sink_drop(env, CACHE(h));
}
/* long routeCacheGet(long cache, byte[] tenant, byte[] topic, long nowMs, IntBuffer outIds, long[] epochOut)
 * ISubscriptionCache.get(tenantId, topic): -> number of route ids, or -(needed); a hit never leaves the host */
JNIEXPORT jlong JNICALL NM(routeCacheGet)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jbyteArray topic, jlong nowMs, jobject outIds,
                                          jlongArray epochOut) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant), pl = (*env)->GetArrayLength(env, topic);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    jbyte* tp = (*env)->GetByteArrayElements(env, topic, NULL);
    uint32_t n = 0;
    uint64_t epoch = 0;
    const int rc = bmq_route_cache_get(CACHE(h), (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)tp, (uint32_t)pl, (uint64_t)nowMs,
                                       (uint32_t*)ADDR(outIds), (uint32_t)CAP(outIds), &n, &epoch);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, topic, tp, JNI_ABORT);
    const jlong ep = (jlong)epoch;
    (*env)->SetLongArrayRegion(env, epochOut, 0, 1, &ep);
    return result_of(env, NULL, "bmq_route_cache_get", rc, n);
}
/* int routeCacheIsCached(long cache, byte[] tenant, byte[] filter)      ISubscriptionCache.isCached: 1 / 0 */
JNIEXPORT jint JNICALL NM(routeCacheIsCached)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jbyteArray filter) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant), fl = (*env)->GetArrayLength(env, filter);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    jbyte* ft = (*env)->GetByteArrayElements(env, filter, NULL);
    const int rc = bmq_route_cache_is_cached(CACHE(h), (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)ft, (uint32_t)fl);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, filter, ft, JNI_ABORT);
    if (rc < 0) throw_state(env, NULL, "bmq_route_cache_is_cached", rc);
    return rc > 0;
}
/* void routeCacheApply(long cache, ByteBuffer keys, IntBuffer keyOff, ByteBuffer ops, int n)     ISubscriptionCache.refresh */
JNIEXPORT void JNICALL NM(routeCacheApply)(JNIEnv* env, jclass c, jlong h, jobject keys, jobject keyOff, jobject ops, jint n) {
    (void)c;
    const int rc = bmq_route_cache_apply(CACHE(h), (const uint8_t*)ADDR(keys), (const uint32_t*)ADDR(keyOff), (const uint8_t*)ADDR(ops), (uint32_t)n);
    if (rc != BMQ_OK) throw_state(env, NULL, "bmq_route_cache_apply", rc);
}
/* void routeCacheRebuild(long cache, ByteBuffer keys, IntBuffer keyOff, int n)      IKVRangeCoProc.reset through the cache */
JNIEXPORT void JNICALL NM(routeCacheRebuild)(JNIEnv* env, jclass c, jlong h, jobject keys, jobject keyOff, jint n) {
    (void)c;
    const int rc = bmq_route_cache_rebuild(CACHE(h), (const uint8_t*)ADDR(keys), (const uint32_t*)ADDR(keyOff), (uint32_t)n);
    if (rc != BMQ_OK) throw_state(env, NULL, "bmq_route_cache_rebuild", rc);
}
/* void routeCacheReset(long cache)      ISubscriptionCache.reset(boundary) */
JNIEXPORT void JNICALL NM(routeCacheReset)(JNIEnv* env, jclass c, jlong h) {
    (void)env, (void)c;
    (void)bmq_route_cache_reset(CACHE(h));
}
/* long routeCacheGetBatch(long cache, ByteBuffer tenants, IntBuffer tenantOff, int nTenants, IntBuffer topicTenant, ByteBuffer topics,
 *                         IntBuffer topicOff, int nTopics, long nowMs, IntBuffer outRowPtr, IntBuffer outIds, ByteBuffer outHit)
 * a whole BatchDistRequest (DistWorkerCoProc.batchDist): cached rows from the host, every miss in ONE launch; -> ids, or -(needed) */
JNIEXPORT jlong JNICALL NM(routeCacheGetBatch)(JNIEnv* env, jclass c, jlong h, jobject tenants, jobject tenantOff, jint nTenants, jobject topicTenant,
                                               jobject topics, jobject topicOff, jint nTopics, jlong nowMs, jobject outRowPtr, jobject outIds,
                                               jobject outHit) {
    (void)c;
    uint64_t need = 0;
    const int rc = bmq_route_cache_get_batch(CACHE(h), (const uint8_t*)ADDR(tenants), (const uint32_t*)ADDR(tenantOff), (uint32_t)nTenants,
                                             (const uint32_t*)ADDR(topicTenant), (const uint8_t*)ADDR(topics), (const uint32_t*)ADDR(topicOff),
                                             (uint32_t)nTopics, (uint64_t)nowMs, (uint32_t*)ADDR(outRowPtr), (uint32_t*)ADDR(outIds), CAP(outIds), &need,
                                             (uint8_t*)ADDR(outHit));
    return result_of(env, NULL, "bmq_route_cache_get_batch", rc, need);
}
/* long routeCacheExpire(long cache, long nowMs)      the sweep Caffeine's scheduler does: -> entries dropped */
JNIEXPORT jlong JNICALL NM(routeCacheExpire)(JNIEnv* env, jclass c, jlong h, jlong nowMs) {
    (void)env, (void)c;
    uint64_t n = 0;
    (void)bmq_route_cache_expire(CACHE(h), (uint64_t)nowMs, &n);
    return (jlong)n;
}

/* ---- futures: bmq_route_cache_get_async completes a Java callback object ---------------------------------------------------------- */
/* void routeCacheGetAsync(long cache, byte[] tenant, byte[] topic, long nowMs, RouteCallback cb)
 * cb.onRoutes(int status, int[] routeIds, long epoch) runs on the calling thread for a cache hit (before this method returns) and on the
 * batching front's dispatcher thread for a miss -- that thread is attached to the JVM as a daemon the first time it calls back.  The Java
 * side completes a CompletableFuture in onRoutes (GpuSubscriptionCache.get). */
static JavaVM* g_vm;
/* A native thread this file attached to the JVM (the batching front's dispatcher) is detached again when it exits: the key's destructor
 * runs at thread exit, i.e. inside bmq_batcher_destroy's join.  Threads that were Java threads all along are never detached. */
static pthread_key_t g_detach_key;
static pthread_once_t g_detach_once = PTHREAD_ONCE_INIT;
static void detach_at_exit(void* vm) {
    if (vm) (*(JavaVM*)vm)->DetachCurrentThread((JavaVM*)vm);
}
static void make_detach_key(void) { (void)pthread_key_create(&g_detach_key, detach_at_exit); }
/* the JNIEnv of the current thread, attaching it (as a daemon) if it is not a Java thread yet; NULL: no JVM to call into */
static JNIEnv* env_of_this_thread(void) {
    JNIEnv* env = NULL;
    if (!g_vm) return NULL;
    if ((*g_vm)->GetEnv(g_vm, (void**)&env, JNI_VERSION_1_8) == JNI_OK) return env;
    if ((*g_vm)->AttachCurrentThreadAsDaemon(g_vm, (void**)&env, NULL) != JNI_OK) return NULL;
    (void)pthread_once(&g_detach_once, make_detach_key);
    (void)pthread_setspecific(g_detach_key, (void*)g_vm);
    return env;
}
typedef struct async_ctx {
    jobject cb; /* global ref */
} async_ctx;
static jmethodID g_on_routes; /* RouteCallback.onRoutes(I[IJ)V, looked up once (all callbacks implement the one interface) */
static void async_done(void* user, int status, const uint32_t* ids, uint32_t n, uint64_t epoch) {
    async_ctx* a = (async_ctx*)user;
    JNIEnv* env = env_of_this_thread();
    if (!env) {
        free(a); /* no JVM to call back into: the future stays incomplete, as it would on any lost thread (the global ref dies with the JVM) */
        return;
    }
    jintArray arr = (*env)->NewIntArray(env, (jsize)n);
    if (g_on_routes && arr) {
        if (n) (*env)->SetIntArrayRegion(env, arr, 0, (jsize)n, (const jint*)ids);
        (*env)->CallVoidMethod(env, a->cb, g_on_routes, (jint)status, arr, (jlong)epoch);
    } else if (g_on_routes) { /* out of Java heap for the id array: complete the future exceptionally rather than never */
        if ((*env)->ExceptionCheck(env)) (*env)->ExceptionClear(env);
        (*env)->CallVoidMethod(env, a->cb, g_on_routes, (jint)BMQ_E_NOMEM, (jintArray)NULL, (jlong)epoch);
    }
    if ((*env)->ExceptionCheck(env)) (*env)->ExceptionClear(env); /* a throwing callback must not poison the dispatcher thread */
    if (arr) (*env)->DeleteLocalRef(env, arr);
    (*env)->DeleteGlobalRef(env, a->cb);
    free(a);
}
JNIEXPORT void JNICALL NM(routeCacheGetAsync)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jbyteArray topic, jlong nowMs, jobject cb) {
    (void)c;
    if (!g_vm) (*env)->GetJavaVM(env, &g_vm);
    async_ctx* a = (async_ctx*)malloc(sizeof *a);
    if (!a) {
        throw_state(env, NULL, "routeCacheGetAsync: out of memory", BMQ_E_NOMEM);
        return;
    }
    a->cb = (*env)->NewGlobalRef(env, cb);
    if (!a->cb) { /* out of memory for the reference: an OutOfMemoryError is pending */
        free(a);
        return;
    }
    if (!g_on_routes) {
        jclass cls = (*env)->GetObjectClass(env, cb);
        g_on_routes = (*env)->GetMethodID(env, cls, "onRoutes", "(I[IJ)V");
        (*env)->DeleteLocalRef(env, cls);
        if (!g_on_routes) { /* NoSuchMethodError is pending */
            (*env)->DeleteGlobalRef(env, a->cb);
            free(a);
            return;
        }
    }
    const jsize tl = (*env)->GetArrayLength(env, tenant), pl = (*env)->GetArrayLength(env, topic);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    jbyte* tp = (*env)->GetByteArrayElements(env, topic, NULL);
    const int rc = bmq_route_cache_get_async(CACHE(h), (const uint8_t*)tn, (uint32_t)tl, (const uint8_t*)tp, (uint32_t)pl, (uint64_t)nowMs, async_done, a);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, topic, tp, JNI_ABORT);
    if (rc != BMQ_OK) { /* not submitted: nobody will call back */
        (*env)->DeleteGlobalRef(env, a->cb);
        free(a);
        throw_state(env, NULL, "bmq_route_cache_get_async", rc);
    }
}

/* ---- fan-out caps and meters of the route cache -------------------------------------------------------------------------------------- */
/* void routeCacheSetCaps(long cache, byte[] tenant, int maxPersistentFanout, int maxGroupFanout)
 * ISettingProvider.provide(MaxPersistentFanout / MaxGroupFanout, tenantId) handed down (TenantRouteCache.java:174-175, 124-138) */
JNIEXPORT void JNICALL NM(routeCacheSetCaps)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jint maxPersistentFanout, jint maxGroupFanout) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    const int rc = bmq_route_cache_set_caps(CACHE(h), (const uint8_t*)tn, (uint32_t)tl, maxPersistentFanout, maxGroupFanout);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    if (rc != BMQ_OK) throw_state(env, NULL, "bmq_route_cache_set_caps", rc);
}
/* void routeCacheSetEventSink(long cache, ThrottleSink sink)
 * sink.onThrottle(byte[] tenant, byte[] topic, int type, int routeId, int maxCount) is IEventCollector.report(PersistentFanoutThrottled
 * (type 0) / GroupFanoutThrottled (type 1)), MatchedRoutes.java:95-101,124-130; it runs on the thread that completes the load (a Java
 * matcher thread, or the dispatcher thread of the batching front).  One sink PER CACHE: a dist worker hosts one GpuSubscriptionCache per
 * range and each installs a lambda that resolves route ids against ITS range's index -- the sink object travels as the `user` pointer of
 * bmq_route_cache_set_event_sink (a global ref held in a registry entry of the cache); the entries of a cache -- the current one and
 * those it replaced -- are deleted when the cache is destroyed.  (Round 3 kept one global sink per process: the events of the second and later caches went to the
 * first cache's lambda.) */
typedef struct sink_entry {
    struct sink_entry* next;
    bmq_route_cache* cache;
    jobject ref; /* global ref of the cache's ThrottleSink */
} sink_entry;
static sink_entry* g_sinks;
static pthread_mutex_t g_sinks_mu = PTHREAD_MUTEX_INITIALIZER;
static jmethodID g_on_throttle; /* ThrottleSink.onThrottle([B[BIII)V, looked up once (all sinks implement the one interface) */
static void throttle_event(void* user, const uint8_t* tenant, uint32_t tl, const uint8_t* topic, uint32_t pl, int32_t type, uint32_t route_id,
                           int32_t max_count) {
    sink_entry* se = (sink_entry*)user;
    JNIEnv* env = env_of_this_thread();
    if (!env || !se || !se->ref || !g_on_throttle) return;
    jbyteArray jt = (*env)->NewByteArray(env, (jsize)tl), jp = (*env)->NewByteArray(env, (jsize)pl);
    if (jt && jp) {
        if (tl) (*env)->SetByteArrayRegion(env, jt, 0, (jsize)tl, (const jbyte*)tenant);
        if (pl) (*env)->SetByteArrayRegion(env, jp, 0, (jsize)pl, (const jbyte*)topic);
        (*env)->CallVoidMethod(env, se->ref, g_on_throttle, jt, jp, (jint)type, (jint)route_id, (jint)max_count);
    }
    if ((*env)->ExceptionCheck(env)) (*env)->ExceptionClear(env);
    if (jt) (*env)->DeleteLocalRef(env, jt);
    if (jp) (*env)->DeleteLocalRef(env, jp);
}
/* frees every registry entry of the cache and their refs.  Only routeCacheDestroy calls it: a loader thread that read the cache's sink just
 * before a replacement may still call the OLD entry afterwards (bmq_route_cache_set_event_sink only swaps a pointer), so an entry that was
 * replaced or cleared stays allocated, with its global ref, until the cache is gone (freeing it at replace time was a use after
 * free). */
static void sink_drop(JNIEnv* env, bmq_route_cache* cache) {
    sink_entry* dead = NULL;
    pthread_mutex_lock(&g_sinks_mu);
    for (sink_entry** pp = &g_sinks; *pp;) {
        if ((*pp)->cache == cache) {
            sink_entry* e = *pp;
            *pp = e->next;
            e->next = dead;
            dead = e;
        } else pp = &(*pp)->next;
    }
    pthread_mutex_unlock(&g_sinks_mu);
    while (dead) {
        sink_entry* e = dead;
        dead = e->next;
        if (e->ref) (*env)->DeleteGlobalRef(env, e->ref);
        free(e);
    }
}
JNIEXPORT void JNICALL NM(routeCacheSetEventSink)(JNIEnv* env, jclass c, jlong h, jobject sink) {
    (void)c;
    if (!g_vm) (*env)->GetJavaVM(env, &g_vm);
    if (!sink) { /* reporting off: the cache stops calling; the entry stays until routeCacheDestroy (a call may be in flight) */
        (void)bmq_route_cache_set_event_sink(CACHE(h), NULL, NULL);
        return;
    }
    if (!g_on_throttle) {
        jclass cls = (*env)->GetObjectClass(env, sink);
        g_on_throttle = (*env)->GetMethodID(env, cls, "onThrottle", "([B[BIII)V");
        (*env)->DeleteLocalRef(env, cls);
        if (!g_on_throttle) return; /* NoSuchMethodError is pending */
    }
    sink_entry* se = (sink_entry*)calloc(1, sizeof *se);
    if (!se) {
        throw_state(env, NULL, "routeCacheSetEventSink: out of memory", BMQ_E_NOMEM);
        return;
    }
    se->cache = CACHE(h);
    se->ref = (*env)->NewGlobalRef(env, sink);
    if (!se->ref) { /* OutOfMemoryError is pending */
        free(se);
        return;
    }
    const int rc = bmq_route_cache_set_event_sink(CACHE(h), throttle_event, se); /* new loads call the NEW entry; one in flight may still call the old */
    if (rc != BMQ_OK) {
        (*env)->DeleteGlobalRef(env, se->ref);
        free(se);
        throw_state(env, NULL, "bmq_route_cache_set_event_sink", rc);
        return;
    }
    pthread_mutex_lock(&g_sinks_mu); /* (the entry this one replaces, if any, stays registered: freed with the cache) */
    se->next = g_sinks;
    g_sinks = se;
    pthread_mutex_unlock(&g_sinks_mu);
}
/* boolean routeCacheTenantStats(long cache, byte[] tenant, long[] out)     out[0..4] = hits, misses, evictions, entries, cached routes --
 * the MqttRouteCacheHitCount / MissCount / EvictCount counters and the MqttRouteCacheSize gauge of TenantRouteCache.java:141-147;
 * false: the tenant has no cache (its meters went with it) */
JNIEXPORT jboolean JNICALL NM(routeCacheTenantStats)(JNIEnv* env, jclass c, jlong h, jbyteArray tenant, jlongArray out) {
    (void)c;
    const jsize tl = (*env)->GetArrayLength(env, tenant);
    jbyte* tn = (*env)->GetByteArrayElements(env, tenant, NULL);
    bmq_route_cache_tenant_stats st;
    const int rc = bmq_route_cache_tenant_stats_get(CACHE(h), (const uint8_t*)tn, (uint32_t)tl, &st);
    (*env)->ReleaseByteArrayElements(env, tenant, tn, JNI_ABORT);
    if (rc == BMQ_E_STATE) return 0;
    if (rc != BMQ_OK) {
        throw_state(env, NULL, "bmq_route_cache_tenant_stats_get", rc);
        return 0;
    }
    const jlong v[5] = {(jlong)st.hits, (jlong)st.misses, (jlong)st.evictions, (jlong)st.entries, (jlong)st.cached_routes};
    (*env)->SetLongArrayRegion(env, out, 0, 5, v);
    return 1;
}
