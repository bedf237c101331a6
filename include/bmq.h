/*
 * bmq.h -- C ABI of the MI355X-native MQTT topic-match engine (libbmq.so).
 *
 * This is the drop-in boundary.  The reference (apache/bifromq, 100 % Java) has no FFI of its own;
 * each entry point below names the Java seam it sits behind (paths relative to the reference root,
 * DW = bifromq-dist/bifromq-dist-worker/src/main/java/org/apache/bifromq/dist/worker,
 * RS = bifromq-retain/bifromq-retain-store/src/main/java/org/apache/bifromq/retain/store,
 * SCHEMA = bifromq-dist/bifromq-dist-worker-schema/src/main/java/org/apache/bifromq/dist/worker/schema,
 * KVAPI = base-kv/base-kv-store-coproc-api/src/main/java/org/apache/bifromq/basekv/store/api).
 * INTEGRATION.md shows the JNI stub a maintainer would add on the Java side.
 *
 * Conventions
 *   - every function returns BMQ_OK (0) or a negative bmq_status; nothing throws across the ABI;
 *   - strings are packed: `bytes` + `off[n+1]` (uint32 byte offsets, off[0] == 0), UTF-8, not NUL-terminated;
 *   - a route id is a STABLE handle of one route key.  bmq_rebuild numbers the routes by the RANK of their KV key in
 *     unsigned-byte order (== KV iteration order, the order MatchedRoutes applies fan-out caps in,
 *     DW/cache/MatchedRoutes.java:87-141); a route added later by bmq_routes_apply gets the next unused id.  An id never
 *     changes and is never given to another key until the next bmq_rebuild (bmq_index_info.generation counts rebuilds);
 *     the id of a deleted route resolves to "no such route".  bmq_route_key() / bmq_route_keys() map ids back to keys;
 *     bmq_match_all() applies the fan-out caps in KEY order whatever the ids are;
 *   - match results are CSR: row_ptr[n+1] + ids, ids ascending inside each row;
 *   - buffers are caller-owned; *_dev variants take DEVICE pointers (HBM-resident inputs/outputs) and
 *     run asynchronously on the engine's HIP stream until bmq_match_finish().  Device string buffers (tenants,
 *     topics, filters) must be 16-byte aligned and readable 16 bytes past their last byte (the kernels read them in
 *     aligned words); the host-buffer variants stage and pad internally, so host callers have no such duty;
 *   - thread-safety: every call on one engine is serialised internally (matcher threads of the reference's
 *     ForkJoinPool, DW/DistWorkerCoProcFactory.java:74-88, may all call into it); ONE device batch may be in flight
 *     per engine through the *_dev protocol: bmq_match_batch_dev / bmq_retain_match_batch_dev take the engine for their caller
 *     until the SAME thread calls bmq_match_finish (other threads' calls wait; another thread's *_dev launch gets BMQ_E_STATE).
 *     bmq_match_submit / bmq_match_wait keep up to BMQ_MAX_TICKETS host batches in flight.  Use one engine per KV range replica.
 *   - the engine REQUIRES a gfx950 device for every match call.  There is no CPU fallback: without a
 *     device bmq_engine_create(device >= 0) fails with BMQ_E_NODEVICE.
 */
#ifndef BMQ_H
#define BMQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bmq_status {
    BMQ_OK = 0,
    BMQ_E_INVAL = -1,     /* bad argument / malformed route key                                   */
    BMQ_E_NODEVICE = -2,  /* no HIP device (or engine created host-only) -- match is impossible    */
    BMQ_E_NOSPACE = -3,   /* caller's output buffer too small; *out_needed tells the size          */
    BMQ_E_NOMEM = -4,     /* host or device allocation failed                                      */
    BMQ_E_HIP = -5,       /* a HIP runtime call failed; see bmq_last_error()                       */
    BMQ_E_RANGE = -6,     /* a 32-bit size limit of the ABI exceeded (>= 2^32 ids in one batch...) */
    BMQ_E_STATE = -7      /* call not valid in this state (e.g. match before rebuild)              */
} bmq_status;

typedef struct bmq_engine bmq_engine;

/* Tunables.  Zero-initialise, set struct_size = sizeof(bmq_config), override what you need.
 * The *_cap fields exist so tests can force the LDS overflow (global spill) paths; 0 = default. */
typedef struct bmq_config {
    uint32_t struct_size;
    int32_t device;            /* HIP device ordinal; -1 = host-only engine (build/inspect, no match)      */
    uint32_t wave_queue_cap;   /* per-wave LDS work stack of the walk kernel.  Its LDS geometry is compiled in, so the two caps are a  */
                               /* SELECTOR, not sizes: 0 = default (176 stack items / 152 range entries); 128 in either = the smallest */
                               /* lists (128 / 128: tests force the overflow paths with them).  Any other value is refused           */
                               /* (BMQ_E_INVAL): it would silently run the default.  Overflow is parked in global memory, never an error */
    uint32_t wave_pair_cap;    /* per-wave LDS matched-range buffer: 0 or 128, as above                                          */
    uint32_t slow_scratch_mb;  /* global scratch for the per-lane DFS slow path (default 64)               */
    uint32_t kernel_timing;    /* 1: HIP events around k_walk / k_expand of every batch -> bmq_stats.ms_walk /  */
                               /* ms_expand (two extra events per batch, ~4 us each on the stream); 0: ms_total  */
    uint32_t dedup_min_topics; /* a batch of at least this many topics is de-duplicated on the device first: identical      */
                               /* (tenant, topic) rows are walked once, every row keeps its own row in the result (matchAll   */
                               /* takes a Set<String>, TenantRouteMatcher.java:67-78).  0 = default = UINT32_MAX = never: on  */
                               /* the survey's Zipf batches it takes 16 % off the walk kernel and costs more than that in the */
                               /* two kernels around it                                                 */
    uint32_t dedup_sorted;     /* 1: the caller's batches are ORDERED by (tenant index, topic bytes) -- BatchDistRequest is "sorted by   */
                               /* tenantId and topic" (DistWorkerCoProc.proto:75-83) --, so the de-duplication above compares a row with */
                               /* the row before it instead of hashing, and the walk runs on a dense copy of the distinct rows.  Only    */
                               /* speed depends on the order: a row that equals no neighbour is matched on its own.  0 = hash.          */
                               /* Measured on the survey's 1 M-publish batch (profiles/r05b/): 56 us of de-duplication (hash: 125 us)  */
                               /* for 31 us less walking -- a loss there; a caller that sends every topic ONCE gains 17 %              */
    uint32_t region_slack;     /* a tenant's region of the filter trie holds nodes x (1 + region_slack / 4) buckets of two 32-byte slots:    */
                               /* 0 = default = 6 (load factor 0.2, 160 bytes of region per trie node); 1 = load factor 0.4 (64 + 16 bytes  */
                               /* per node: the layout of rounds 2-5; k_walk is 8 % slower on the survey's workload -- more second probes), */
                               /* up to 64.  A memory / speed trade, nothing else depends on it                                            */
    uint32_t reserved[4];
} bmq_config;

/* Counters of the last completed match batch (for roofline accounting). */
typedef struct bmq_stats {
    uint64_t n_topics;
    uint64_t n_visit;        /* filter-trie nodes discovered (root excluded), all topics                    */
    uint64_t n_match;        /* route ids emitted                                                           */
    uint64_t n_ranges;       /* matched (filter node) ranges                                                */
    uint64_t n_slow_topics;  /* topics resolved by the slow path (more than 16 levels)                     */
    uint64_t n_sorted_rows;  /* rows that needed the element-level fix-up sort                              */
    uint64_t topic_bytes;    /* sum of topic lengths                                                        */
    float ms_total;          /* HIP-event time of the whole batch on the engine stream (0 for batches of fewer */
                             /* than 4096 topics unless bmq_config.kernel_timing: the two events cost ~8 us)  */
    float ms_walk;           /* ... of the tokenise+walk kernel alone (0 unless bmq_config.kernel_timing)   */
    float ms_expand;         /* ... of the expand kernel alone (0 unless bmq_config.kernel_timing)          */
    uint32_t n_walked;       /* rows the walk kernel walked: n_topics, fewer with bmq_config.dedup_sorted (the distinct  */
                             /* rows), 0 = not counted (the hashing de-duplication)                                       */
    uint32_t n_split_blocks; /* 64-row blocks of the batch the expand kernel gave to four waves each (blocks of more than  */
                             /* ~2 x the ranges / ids of the mean block: batches of 65536 rows and more)                   */
    uint32_t reserved0;
} bmq_stats;

typedef struct bmq_index_info {
    uint64_t n_routes, n_tenants, n_nodes, n_tokens;
    uint64_t trie_slots, dict_slots;      /* table sizes in 32-byte slots (trie: sum of the tenant regions)  */
    uint64_t device_bytes;                /* HBM held by the index                                          */
    uint64_t epoch;                       /* +1 per bmq_rebuild / bmq_routes_apply                          */
    uint64_t generation;                  /* +1 per bmq_rebuild: ids of different generations are unrelated */
    uint64_t next_route_id;               /* ids handed out so far (live + deleted)                         */
    uint64_t garbage_bytes;               /* HBM held by abandoned regions / id lists until the next rebuild */
} bmq_index_info;

/* ---- lifecycle -------------------------------------------------------------------------------------- */
/* One engine per KV range replica, like DistWorkerCoProc's SubscriptionCache
 * (DW/DistWorkerCoProcFactory.java:91-93). */
int bmq_engine_create(const bmq_config* cfg, bmq_engine** out);
void bmq_engine_destroy(bmq_engine* e);
const char* bmq_last_error(const bmq_engine* e);
const char* bmq_version(void);

/* ---- dist direction: index maintenance -------------------------------------------------------------- */
/* Full (re)load from a KV scan -- replaces IKVRangeCoProc.reset(Boundary) (KVAPI/IKVRangeCoProc.java:64,
 * DW/DistWorkerCoProc.java:283-291).  keys = route keys in the layout of SCHEMA/KVSchemaUtil.java:91-130 (the packed bytes
 * must be readable 16 bytes past the last key).  The keys are uploaded and parsed, and the index is built, ON THE DEVICE
 * (builder kernels, bmq_build_core.h).  A KV iterator yields the keys strictly ascending: that is the fast path and route
 * id = position = rank; any other order is sorted + de-duplicated on the host first.  On failure the previous index is gone. */
int bmq_rebuild(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, uint32_t n_keys);

/* Post-commit route mutations -- replaces ISubscriptionCache.refresh(AddRoutesTask/RemoveRoutesTask)
 * (DW/DistWorkerCoProc.java:188-209, DW/cache/SubscriptionCache.java:127-134).  op[i]: 0 = put, 1 = delete.
 * Applied in order by builder kernels on the engine stream (between match batches, which therefore never see half a batch);
 * the call returns when the device has applied it.  A put of a key that is already there keeps its id, a delete of an absent
 * key is a no-op; the j-th put of the batch that adds a route gets id next_route_id + j.  A malformed key or op code fails
 * the whole batch with BMQ_E_INVAL before anything is changed. */
int bmq_routes_apply(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n);
/* The same without the wait (DistWorkerCoProc applies a batch of mutations and goes on serving, DW/DistWorkerCoProc.java:188-209): the ops
 * are uploaded on the engine's copy stream -- beside a batch handed over with bmq_match_submit* that still runs --, the builder kernels are
 * queued behind that batch, the call returns after the enqueue.  The buffers (page-locked memory from bmq_host_alloc, or the upload is
 * not asynchronous) must stay untouched until the batch's outcome has been fetched: bmq_routes_apply_wait returns what bmq_routes_apply
 * would have returned (nothing is changed by a batch that holds a malformed key).  Every later call that reads or changes the route index
 * fetches the outcome first -- a match launched behind a failed batch returns that batch's error.  At most one batch is open: a second
 * bmq_routes_apply[_async] completes the first. */
int bmq_routes_apply_async(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n);
int bmq_routes_apply_wait(bmq_engine* e);

/* Maintenance: re-build the index from its own live routes (the keys are gathered from the HBM key store).  Frees what churn leaves
 * behind until then -- abandoned tenant regions and id lists, trie nodes and dictionary tokens of filters nobody subscribes to any
 * more (bmq_index_info.garbage_bytes) -- and re-numbers the route ids to ranks: a new generation, like bmq_rebuild. */
int bmq_compact(bmq_engine* e);
/* The same without the stall (TopicLevelTrie contracts as it goes, UTIL/index/TopicLevelTrie.java:257-384; GenerationalRangeIndex.java /
 * bifromq_amd/generations.py did this with two handles on the caller's side, shipping every key through the host).  The next generation of
 * the route index is built BESIDE the serving one, inside the engine: on its own executor and a lowest-priority stream, from the serving
 * generation's live keys -- gathered HBM to HBM: the key bytes never leave the device, nothing is sorted on the host --, in chunks the caller
 * paces from a maintenance thread while its matcher threads go on:
 *   bmq_compact_begin   starts one: sizes the next generation's regions, pools and tables from the serving one's (BMQ_E_STATE while one is
 *                       running; bmq_rebuild / bmq_compact are refused until it is swapped or aborted).
 *   bmq_compact_poll    carries the live keys among the next max_ids route ids over.  The engine lock is held only while the build stream is
 *                       put behind what the serving generation was told so far and a copy of the chunk's key references is ENQUEUED; the
 *                       gather, the builder kernels and the waits touch the new generation alone.  *out_done_permille = 1000: ready to swap.
 *                       Matching and bmq_routes_apply[_async] go on meanwhile; what is mutated is logged.  Measured (bench.py, compaction
 *                       leg: 10 M routes, 1 M-publish batches back to back on another thread): batch p99 0.35 -> 0.60 ms while it runs.
 *   bmq_compact_swap    replays the log, swaps the generations (no batch may be in flight: BMQ_E_STATE) and frees the old one.  Route ids are
 *                       re-numbered: bmq_index_info.generation + 1, ids of the old generation mean nothing any more (as after bmq_compact).
 *   bmq_compact_abort   drops the half-built generation.
 * One compaction call at a time (they serialise among themselves; call them from a maintenance thread, not from the matcher threads); a
 * host-only engine runs the same procedure over the host executor.  The log of mutations lives in host memory and grows until
 * bmq_compact_swap / _abort: a caller that begins a compaction finishes it.  Only the route index is covered (retained topics:
 * bmq_retain_compact). */
int bmq_compact_begin(bmq_engine* e);
int bmq_compact_poll(bmq_engine* e, uint32_t max_ids, uint32_t* out_done_permille);
int bmq_compact_swap(bmq_engine* e, uint64_t* out_carried /* may be NULL */, uint64_t* out_replayed /* may be NULL */);
int bmq_compact_abort(bmq_engine* e);

int bmq_index_info_get(const bmq_engine* e, bmq_index_info* out);
/* id -> key (so the Java adapter can materialise Matching objects, SCHEMA/KVSchemaUtil.java:73-89).  BMQ_E_INVAL: no such
 * route (the id was never handed out, or its route has been deleted). */
int bmq_route_key(const bmq_engine* e, uint32_t route_id, uint8_t* out, uint32_t cap, uint32_t* out_len);
/* Many ids at once (one device gather): out_off[n + 1] byte offsets into out; a dead id gives an empty key.
 * BMQ_E_NOSPACE if cap < out_off[n] (offsets are still written). */
int bmq_route_keys(const bmq_engine* e, const uint32_t* route_ids, uint32_t n, uint8_t* out, uint64_t cap, uint64_t* out_off);
/* exact lookup (no wildcard semantics): ids of the routes stored under (tenant, topicFilter); for tests
 * and for RouteDetailCache-style inspection.  Writes up to cap ids, *out_n = total. */
int bmq_index_find(const bmq_engine* e, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filter,
                   uint32_t filter_len, uint32_t* out_ids, uint32_t cap, uint32_t* out_n);

/* ---- dist direction: match --------------------------------------------------------------------------- */
/* Batch form of ITenantRouteMatcher.matchAll (DW/cache/ITenantRouteMatcher.java:37; implementation
 * DW/cache/TenantRouteMatcher.java:67-161) across tenants, as DistWorkerCoProc.batchDist iterates
 * DistPack(tenant) x TopicMessagePack(topic) (DW/DistWorkerCoProc.java:515-552).
 *   tenants/tenant_off : the distinct tenant ids of the batch (n_tenants packed strings)
 *   topic_tenant[i]    : index into that table for topic i
 *   topics/topic_off   : n_topics packed topic strings
 * Output: out_row_ptr[n_topics+1], out_route_ids[*out_needed] (ascending per row).  If out_capacity is too
 * small nothing is written to out_route_ids, *out_needed is set and BMQ_E_NOSPACE returned. */
int bmq_match_batch(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                    const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off,
                    uint32_t n_topics, uint32_t* out_row_ptr, uint32_t* out_route_ids, uint64_t out_capacity,
                    uint64_t* out_needed);

/* Same, all seven data pointers are DEVICE pointers (inputs already resident in HBM; results stay in HBM).
 * Asynchronous on the engine stream; d_out_total (device uint64) receives the id count.  Call bmq_sync()
 * (or bmq_match_finish()) before reading results.  d_topics must be 16-byte aligned and readable up to
 * topic_off[n_topics] rounded up to the next multiple of 16 bytes. */
int bmq_match_batch_dev(bmq_engine* e, const uint8_t* d_tenants, const uint32_t* d_tenant_off, uint32_t n_tenants,
                        const uint32_t* d_topic_tenant, const uint8_t* d_topics, const uint32_t* d_topic_off,
                        uint32_t n_topics, uint32_t* d_out_row_ptr, uint32_t* d_out_route_ids,
                        uint64_t out_capacity, uint64_t* d_out_total);
/* Waits for the stream, resolves deferred conditions of the last *_dev batch (internal scratch growth ->
 * transparent re-run; BMQ_E_NOSPACE if out_capacity was too small) and fills stats. */
int bmq_match_finish(bmq_engine* e, uint64_t* out_total);
int bmq_sync(bmq_engine* e);
int bmq_stats_get(const bmq_engine* e, bmq_stats* out);
/* switches bmq_config.kernel_timing at run time (profilers / bench.py); takes effect with the next batch */
int bmq_set_kernel_timing(bmq_engine* e, int on);
/* The hipStream_t the engine launches on (as void*), so a harness can bracket it with HIP events. */
void* bmq_stream(const bmq_engine* e);

/* ---- asynchronous host-buffer match (the host-visible fast path) ------------------------------------------------------ */
/* Page-locked host memory.  Buffers from here move over PCIe by DMA at full speed and truly asynchronously; a JNI binding wraps
 * them with NewDirectByteBuffer.  (Any host pointer is accepted everywhere; pageable memory is staged by the HIP runtime.) */
void* bmq_host_alloc(size_t bytes);
void bmq_host_free(void* p);
/* bmq_match_batch split in two so that several batches can be in flight: the upload of batch i+1 (copy-in stream) and the download
 * of batch i-1 (copy-out stream, inside bmq_match_wait) overlap the kernels of batch i (engine stream).  Steady state: one batch
 * costs max(upload, kernels, download) instead of their sum.  Input buffers must stay valid and unchanged until the matching
 * bmq_match_wait returns.  *out_ticket is 0 .. BMQ_MAX_TICKETS - 1; BMQ_E_STATE when all are taken (three, so that a single caller thread
 * that blocks in the download of batch i still has batch i + 1 on the GPU and batch i + 2 on its way up).  bmq_match_wait blocks until the batch is
 * done, copies row_ptr[n_topics + 1] and the ids out (same NOSPACE protocol as bmq_match_batch) and frees the ticket.
 * Tickets may be waited for from any thread; bmq_routes_apply between a submit and its wait is applied BEHIND the submitted batch
 * (stream order). */
#define BMQ_MAX_TICKETS 3
int bmq_match_submit(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                     const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                     int* out_ticket);
int bmq_match_wait(bmq_engine* e, int ticket, uint32_t* out_row_ptr, uint32_t* out_route_ids, uint64_t out_capacity,
                   uint64_t* out_needed);

/* Tickets over DEVICE-ACCESSIBLE buffers: every pointer is HBM or page-locked host memory from bmq_host_alloc, which the GPU reads and
 * writes in place over PCIe (same alignment / padding duties as bmq_match_batch_dev).  Nothing is staged, nothing copied: for small
 * launches -- the batching front's: tens to thousands of topics -- the five uploads, three downloads and their synchronisation cost
 * more than the kernels.  The results (d_out_row_ptr[n + 1], d_out_route_ids, *d_out_total) are complete when bmq_match_wait_dev
 * returns; BMQ_E_NOSPACE + *out_total if out_capacity was short (row pointers are written).  Unlike bmq_match_batch_dev the engine is
 * not taken for the caller: BMQ_MAX_TICKETS such launches may be in flight, from any threads. */
int bmq_match_submit_dev(bmq_engine* e, const uint8_t* d_tenants, const uint32_t* d_tenant_off, uint32_t n_tenants,
                         const uint32_t* d_topic_tenant, const uint8_t* d_topics, const uint32_t* d_topic_off, uint32_t n_topics,
                         uint32_t* d_out_row_ptr, uint32_t* d_out_route_ids, uint64_t out_capacity, uint64_t* d_out_total,
                         int* out_ticket);
int bmq_match_wait_dev(bmq_engine* e, int ticket, uint64_t* out_total /* may be NULL */);

/* ---- host-visible result formats (host enqueue -> results visible on host) ------------------------------------------------------------------------------ */
/* The id CSR of a million-topic batch is ~77 MB over PCIe -- more than twice the topics that went in.  What the callers of the
 * reference's match path do next needs less, so a ticket can be submitted for one of these formats instead (bmq_match_submit_fmt takes the
 * arguments of bmq_match_submit + the format; every format has its own wait; waiting with another format's call is BMQ_E_STATE and
 * leaves the ticket in flight):
 *   BMQ_FMT_IDS      bmq_match_wait: row_ptr + ids.
 *   BMQ_FMT_COUNTS   bmq_match_wait_counts: out_row_ptr[n + 1] only -- the fan-out of topic i is row_ptr[i + 1] - row_ptr[i].  This is
 *                    all DistWorkerCoProc.batchDist replies with (DW/DistWorkerCoProc.java:535-538: BatchDistReply carries the fan-out per
 *                    topic).  4 bytes per topic come back; no id is ever written (the expansion kernel lays down the row pointers and leaves).
 *   BMQ_FMT_RANGES   bmq_match_wait_ranges: the MATCHED RANGES.  Every matched filter owns a run of route ids begin .. begin + count - 1
 *                    (its routes are neighbours in KV key order, SCHEMA/KVSchemaUtil.java:91-117), so a row is a handful of (begin, count)
 *                    pairs -- ~6 per topic where the id row has 18 entries.  out_range_ptr[n + 1] delimits the rows in out_ranges.  A range
 *                    whose count has BMQ_RANGE_SIDE set (filters touched by bmq_routes_apply since the last rebuild) lists its ids
 *                    explicitly: out_side_ids[begin .. begin + (count & ~BMQ_RANGE_SIDE)).  The result is self-contained: expanding
 *                    needs nothing of the index.  The ranges of a row come in ascending order of their first id (rows of more than 32
 *                    ranges: as matched); expanding them in that order yields the row's ids in ascending order -- exactly the row of
 *                    BMQ_FMT_IDS -- unless ranges overlap (interleaved id sets: only after churn), which the consumer sees while expanding
 *                    (first id <= last id of the range before it) and repairs by ordering that row's ids;
 *                    out_info->n_overlapping_rows says how many such rows the batch has (0 after a rebuild).  out_row_ptr (may be NULL)
 *                    receives the id row pointers as well, so that the consumer knows where every expanded row goes.
 *                    BMQ_E_NOSPACE when range_cap / side_cap are too small: out_info holds the sizes, row pointers are written, the ticket
 *                    is released (as with bmq_match_wait).
 *   BMQ_FMT_GROUPED  bmq_match_wait_grouped: the (topic, route) pairs of the batch regrouped by DelivererKey -- the output of
 *                    bmq_fanout_group (see there for the meaning of every array) without the CSR ever leaving the GPU. */
#define BMQ_FMT_IDS 0
#define BMQ_FMT_COUNTS 1
#define BMQ_FMT_RANGES 2
#define BMQ_FMT_GROUPED 3
#define BMQ_RANGE_SIDE 0x80000000u
typedef struct bmq_id_range {
    uint32_t begin, count;
} bmq_id_range;
typedef struct bmq_ranges_info {
    uint64_t n_ranges, n_side_ids;
    uint64_t n_ids;              /* what the id CSR of the batch would hold */
    uint64_t n_overlapping_rows; /* rows whose expanded ids the consumer has to order */
} bmq_ranges_info;
int bmq_match_submit_fmt(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                         const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                         int format, int* out_ticket);
int bmq_match_wait_counts(bmq_engine* e, int ticket, uint32_t* out_row_ptr, uint64_t* out_total /* may be NULL */);
int bmq_match_wait_ranges(bmq_engine* e, int ticket, uint32_t* out_row_ptr /* may be NULL */, uint32_t* out_range_ptr, bmq_id_range* out_ranges,
                          uint64_t range_cap, uint32_t* out_side_ids, uint64_t side_cap, bmq_ranges_info* out_info);
int bmq_match_wait_grouped(bmq_engine* e, int ticket, uint32_t* out_topic, uint32_t* out_route, uint64_t pair_cap, uint32_t* out_group_off,
                           uint32_t* out_group_rep, uint32_t group_cap, uint32_t* out_n_groups, uint32_t* out_special, uint64_t* out_total);

/* ---- multi-GPU exchange -------------------------------------------------------------------------------------- */
/* One process per GPU, tenants sharded by hash(tenantId) mod N; after the per-rank match ONE exchange step over RCCL / xGMI.
 * RCCL is loaded at run time (librccl.so.1); a process that already carries it (PyTorch) keeps using that copy.
 *   bmq_comm_unique_id : rank 0 creates the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by any means
 *   bmq_comm_init      : every rank, with its engine (ncclCommInitRank on the engine's device)
 *   bmq_exchange_fanout: per-topic fan-out counts of every rank to every rank -- all the reference sends upstream
 *                        (DW/DistWorkerCoProc.java:535-538).  d_row_ptr[n + 1] -> d_counts_all[world * n], all device pointers
 *   bmq_exchange_csr   : the complete CSR of every rank on every rank: totals (-> out_totals[world], host), row pointers
 *                        (d_rows_all[world * (n + 1)]), then a true all-gatherv of the ids into d_ids_all (rank r's ids start at
 *                        sum(out_totals[0 .. r))): one ncclBroadcast per rank with its exact size in one group.  n must be the
 *                        same on every rank; BMQ_E_NOSPACE if ids_cap < sum(out_totals) (totals are still reported)
 * Both run asynchronously on the engine's exchange stream, ordered behind everything queued on the engine stream when they are
 * called (the batch whose results they send), so they overlap the NEXT batch's match; bmq_exchange_wait blocks until done. */
int bmq_comm_unique_id(uint8_t out[128]);
int bmq_comm_init(bmq_engine* e, int world, int rank, const uint8_t id[128]);
void bmq_comm_destroy(bmq_engine* e);
int bmq_exchange_fanout(bmq_engine* e, const uint32_t* d_row_ptr, uint32_t n, uint32_t* d_counts_all);
int bmq_exchange_csr(bmq_engine* e, const uint32_t* d_row_ptr, const uint32_t* d_ids, uint32_t n, uint64_t total,
                     uint32_t* d_rows_all, uint32_t* d_ids_all, uint64_t ids_cap, uint64_t* out_totals);
int bmq_exchange_wait(bmq_engine* e);
/* The ids bmq_exchange_csr gathers are rank-local handles: rank r's rows are d_rows_all[r * (n + 1) ...], its ids the stretch of
 * d_ids_all behind sum(out_totals[0 .. r)) -- the position frames them as (rank, id); a consumer resolves an id on the rank that owns it.
 *
 * The node-wide batch: ONE publish batch for all tenants arrives on every GPU; a rank matches the topics whose tenant
 * it owns (d_owner[tenant index] == rank) and those of tenants split by filter over all ranks (d_owner < 0: the reference's analogue is
 * a hot range split by DW/hinter/FanoutSplitHinter.java:49, dist-server sums the per-range fan-outs, BatchDistServerCall.java:186-205).
 * All pointers are device pointers; the part is picked by kernels on the engine stream (mask, two prefix sums, scatter):
 *   d_out_sel[m]        global topic index of the part's k-th topic
 *   d_out_topics/_off   the part's topics packed (the buffer must hold the batch's bytes + 32 and be 16-byte aligned: it is handed to
 *                       bmq_match_batch_dev as is), d_out_off[m + 1]
 *   d_out_tenant[m]     tenant index of every topic (the batch's tenant table stays as it is)
 * *out_n = m, *out_bytes = bytes of the part: the one host read of the step. */
int bmq_partition_batch_dev(bmq_engine* e, const int32_t* d_owner, uint32_t n_tenants, int32_t rank, const uint32_t* d_topic_tenant,
                            const uint8_t* d_topics, const uint32_t* d_topic_off, uint32_t n_topics, uint32_t* d_out_sel, uint8_t* d_out_topics,
                            uint32_t* d_out_off, uint32_t* d_out_tenant, uint32_t* out_n, uint64_t* out_bytes);

/* ---- batching front ------------------------------------------------------------------- */
/* Production asks for one topic per call: TenantRouteCache issues matchAll(singleton(topic)) per cache miss from
 * the matchExecutor pool (DW/cache/TenantRouteCache.java:180-193, DW/DistWorkerCoProcFactory.java:74-88).  A batcher
 * collects the calls of all threads that are waiting right now into ONE bmq_match_batch (leader/follower: no timer,
 * no extra thread; the batch is whatever piled up while the previous one was on the GPU) and hands every caller
 * its rows (identical (tenant, topic) pairs of one launch are matched once).  Any number of threads may call bmq_batcher_match_all concurrently; the call blocks until the batch
 * that contains the request has been matched.  bmq_rebuild / bmq_routes_apply may run concurrently (they serialise
 * with the batches on the engine lock); *out_epoch tells which epoch the returned route ids are ranks of. */
typedef struct bmq_batcher bmq_batcher;
typedef struct bmq_batcher_config {
    uint32_t struct_size;
    uint32_t max_batch_topics; /* upper bound of one launch (default 2^20); a single larger request still runs alone */
    uint32_t reserved[6];
} bmq_batcher_config;
typedef struct bmq_batcher_stats {
    uint64_t n_requests, n_topics, n_batches, max_batch_topics;
    uint64_t n_deduped; /* requested topics that shared a row with an identical (tenant, topic) of the same launch */
} bmq_batcher_stats;
int bmq_batcher_create(bmq_engine* e, const bmq_batcher_config* cfg /* may be NULL */, bmq_batcher** out);
/* Waits for the calls in flight; calls arriving afterwards fail with BMQ_E_STATE.  Destroy before the engine. */
void bmq_batcher_destroy(bmq_batcher* b);
/* ITenantRouteMatcher.matchAll(topics) for ONE tenant, caps not applied (INT_MAX), through the collector.
 * out_row_ptr[n_topics + 1] is relative to this request; BMQ_E_NOSPACE + *out_needed if out_capacity is too small
 * (row pointers are still written). */
int bmq_batcher_match_all(bmq_batcher* b, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics,
                          const uint32_t* topic_off, uint32_t n_topics, uint32_t* out_row_ptr,
                          uint32_t* out_route_ids, uint64_t out_capacity, uint64_t* out_needed, uint64_t* out_epoch);
/* A whole multi-tenant batch (the arguments of bmq_match_batch) as ONE launch of its own, taking its turn with the collected launches;
 * *out_epoch = the epoch of the index the batch saw (bmq_match_batch alone cannot tell).  For callers that already hold many topics:
 * bmq_route_cache_get_batch matches the cache misses of one BatchDistRequest this way. */
int bmq_batcher_match_batch(bmq_batcher* b, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                            const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, uint32_t* out_row_ptr, uint32_t* out_route_ids,
                            uint64_t out_capacity, uint64_t* out_needed, uint64_t* out_epoch);
/* Asynchronous form for loaders that return a future (Caffeine's AsyncCache in TenantRouteCache.java:116-139): the topic is
 * copied and packed into the batch being collected and the call returns at once; a dispatcher thread owned by the batcher
 * (started by the first submit) matches whatever has been collected whenever the engine is free and then calls
 * cb(user, status, ids, n_ids, epoch) once per request, on the dispatcher thread, in submission order.  `ids` is only
 * valid during the callback.  A submitter blocks only while max_batch_topics requests are already waiting.
 * bmq_batcher_destroy matches and calls back everything submitted before it. */
typedef void (*bmq_batcher_cb)(void* user, int status, const uint32_t* route_ids, uint32_t n_ids, uint64_t epoch);
int bmq_batcher_submit(bmq_batcher* b, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len,
                       bmq_batcher_cb cb, void* user);
int bmq_batcher_stats_get(bmq_batcher* b, bmq_batcher_stats* out);

/* ---- the persistent matcher behind the batching front (round 6; bmq_poll_kernel.h) ------------------------------------------------------
 * A generation of the batching front of at most 64 topics is not LAUNCHED any more: a few resident one-wave workgroups (k_poll) poll a ring
 * of request descriptors in page-locked host memory, run the walk + expansion of k_walk / k_expand on the generation whose doorbell rang and
 * write the CSR in place; the leader spins on the slot's completion word.  The reference's call shape -- one matchAll(singleton(topic)) per
 * cache miss from the matchExecutor pool (DW/cache/TenantRouteCache.java:180-193, DW/DistWorkerCoProcFactory.java:74-88) -- pays a
 * PCIe round trip per generation instead of a kernel launch and an event.  It needs no call of its own: bmq_batcher_match_all /
 * bmq_route_cache_get use it whenever it is enabled (the default on a device engine).
 *   - the index never changes under it: bmq_rebuild, bmq_routes_apply[_async], bmq_compact, bmq_compact_swap and bmq_engine_destroy stop it
 *     first (it finishes the generation in hand, a few microseconds); the next generation starts it again.  The FIRST generation after a
 *     start pays the launch (~15 us); a generation whose doorbell it never saw is launched the old way by its leader;
 *   - it leaves the GPU by itself after 20 ms without a doorbell and after 2 s whatever happens;
 *   - a generation not answered within 250 ms marks it wedged: it is told to leave, never started again on this engine
 *     (bmq_poller_stats.n_timeouts), and the generation -- like every later one -- is launched the old way: callers see results, not errors. */
typedef struct bmq_poller_stats {
    uint32_t enabled;     /* generations of <= 64 topics go to it */
    uint32_t running;     /* a k_poll launch is resident right now */
    uint64_t n_starts;    /* k_poll launches so far */
    uint64_t n_served;    /* generations answered by it */
    uint64_t n_fallback;  /* generations handed back or never seen: launched the old way */
    uint64_t n_unserved;  /* ... of which: rang while it was leaving */
    uint64_t n_timeouts;  /* generations that waited 250 ms in vain (the poller is off for good after the first) */
    uint64_t n_bad_input; /* generations a wave refused to touch: offsets or tenant indices out of range (handed back, launched the old way) */
} bmq_poller_stats;
enum { BMQ_POLLER_DISABLE = 0, BMQ_POLLER_ENABLE = 1, BMQ_POLLER_STOP = 2 /* leave now; the next generation starts it again */,
       BMQ_POLLER_TEST_IGNORE_DOORBELLS = 3 /* test hook: doorbells are seen and not answered -- the leaders' time-out path */ };
int bmq_poller_stats_get(bmq_engine* e, bmq_poller_stats* out);
int bmq_poller_control(bmq_engine* e, int what);

/* ---- route cache ---------------------------------------------------------------------------------- */
/* ISubscriptionCache (DW/cache/ISubscriptionCache.java:30-40) on the engine's side of the boundary: SubscriptionCache ->
 * TenantRouteCache (DW/cache/TenantRouteCache.java:116-296: topic -> matched routes, loaded by matchAll(singleton(topic)), bounded by
 * DistMaxCachedRoutesPerTenant with weight max(1, #routes after the caps), expireAfterAccess DistTopicMatchExpirySeconds) with its TopicIndex
 * (DW/TopicIndex.java:39-156: the cached topics, queried with the topic filter of a route mutation).  A hit is answered on the host;
 * a miss goes through the batching front, i.e. into ONE GPU launch with every other miss of the moment.
 *   get       ISubscriptionCache.get(tenantId, topic) = IMatchedRoutes.routes() (DW/cache/TenantRouteCache.java:299-301): the matched
 *             route ids (ascending) AFTER MatchedRoutes' fan-out caps (DW/cache/MatchedRoutes.java:87-141: persistent and group routes
 *             admitted first-come in KV key order), *out_epoch = the engine epoch they were matched at.  The caps are the tenant's
 *             (bmq_route_cache_set_caps; defaults of Setting.java:60-61: MaxPersistentFanout = INT_MAX, MaxGroupFanout = 100); every route
 *             a load throws away is reported to the event sink (bmq_route_cache_set_event_sink) before the load returns; a hit reports
 *             nothing (the reference reports when it loads, too).  An entry met under caps other than the ones it was loaded with follows
 *             MatchedRoutes.adjust (:150-200): it is re-matched when a raised cap could admit more routes or a lowered one is exceeded,
 *             else it adopts the new caps.  The cache weighs an entry by its capped size (TenantRouteCache.java:108).
 *             now_ms is the caller's clock (Caffeine's Ticker).  BMQ_E_NOSPACE + *out_n if cap is short.
 *   is_cached ISubscriptionCache.isCached(tenantId, filterLevels) = !index.match(filterLevels).isEmpty(): 1 / 0
 *   apply     ISubscriptionCache.refresh(AddRoutesTask / RemoveRoutesTask): bmq_routes_apply on the engine, then every cached topic
 *             one of the mutated filters matches is dropped and reloads on its next get (the reference patches those entries in
 *             place -- TenantRouteCache.java:224-277 -- which yields the same route set a reload computes).  A load that was matched
 *             before a mutation but finishes after it is recognised by its epoch and not cached.
 *   rebuild   IKVRangeCoProc.reset: bmq_rebuild + drop everything (route ids of different generations are unrelated); nothing is
 *             served from the cache while the engine swaps the index.   reset: drop everything (ISubscriptionCache.reset(boundary)).
 * Any number of threads may call get / is_cached; apply / rebuild / reset come from one thread at a time (the range's apply thread)
 * and may run concurrently with the getters.  Destroy before the batcher. */
typedef struct bmq_route_cache bmq_route_cache;
typedef struct bmq_route_cache_config {
    uint32_t struct_size;
    uint32_t mutation_log_entries;   /* per tenant; default 4096 */
    uint64_t max_routes_per_tenant;  /* DistMaxCachedRoutesPerTenant, default 200000 */
    uint64_t expiry_ms;              /* DistTopicMatchExpirySeconds, default 60000 */
    uint64_t shards_per_tenant;      /* power of two, default 16: a tenant's topics are spread over that many independently locked */
                                     /* slices, so one hot tenant does not serialise its callers; the route budget is the tenant's */
    uint64_t direct_batch_topics;    /* bmq_route_cache_get_batch: a request of at least this many topics (default 8192) is matched in one */
                                     /* launch without consulting or filling the cache -- at that size the GPU is faster than the probes */
    int32_t default_max_persistent_fanout; /* MaxPersistentFanout of tenants without bmq_route_cache_set_caps; 0 = INT_MAX (Setting.java:61) */
    int32_t default_max_group_fanout;      /* MaxGroupFanout ...; 0 = 100 (Setting.java:60) */
    uint64_t tenant_idle_ms;         /* a tenant nobody called get for since that long loses its whole cache at the next bmq_route_cache_expire */
                                     /* (DW/cache/SubscriptionCache.java:79-107); 0 = 2 x expiry_ms, as there */
} bmq_route_cache_config;
typedef struct bmq_route_cache_stats {
    uint64_t hits, misses, evictions, invalidations, expired;
    uint64_t stale_loads;            /* loads overtaken by a mutation of a matching filter: returned to their caller, not cached */
    uint64_t entries, cached_routes; /* now */
    uint64_t tenants;                /* tenant caches alive now */
    uint64_t tenants_expired;        /* tenant caches destroyed after tenant_idle_ms without a get (their counters stay in the sums above) */
} bmq_route_cache_stats;
/* What TenantRouteCache registers per tenant (DW/cache/TenantRouteCache.java:141-147: MqttRouteCacheHitCount / MissCount / EvictCount
 * counters, MqttRouteCacheSize gauge); like there, the meters go when the tenant's cache is destroyed. */
typedef struct bmq_route_cache_tenant_stats {
    uint64_t hits, misses, evictions;
    uint64_t entries, cached_routes; /* estimatedSize, current weight */
    uint64_t last_get_ms;
    int32_t max_persistent_fanout, max_group_fanout; /* the caps in force */
} bmq_route_cache_tenant_stats;
int bmq_route_cache_create(bmq_engine* e, bmq_batcher* b, const bmq_route_cache_config* cfg /* may be NULL */, bmq_route_cache** out);
void bmq_route_cache_destroy(bmq_route_cache* c);
int bmq_route_cache_get(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len, uint64_t now_ms,
                        uint32_t* out_route_ids, uint32_t cap, uint32_t* out_n, uint64_t* out_epoch);
/* The same as a future (ISubscriptionCache.get returns a CompletableFuture; Caffeine's AsyncLoadingCache, TenantRouteCache.java:116-139):
 * a hit calls cb(user, BMQ_OK, ids, n, epoch) on the calling thread before the call returns; a miss is handed to bmq_batcher_submit and
 * cb runs on the batcher's dispatcher thread once the launch that carries it has finished (the row is cached first, under the same
 * epoch rule).  `route_ids` is only valid during the callback.  Nobody blocks on the GPU: a few threads keep millions of gets per
 * second in flight. */
typedef void (*bmq_route_cache_cb)(void* user, int status, const uint32_t* route_ids, uint32_t n_ids, uint64_t epoch);
int bmq_route_cache_get_async(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len,
                              uint64_t now_ms, bmq_route_cache_cb cb, void* user);
/* DistWorkerCoProc.batchDist (DW/DistWorkerCoProc.java:515-552) asks the cache once per (tenant, topic) of a BatchDistRequest; this is the
 * whole request in one call (arguments and CSR output as bmq_match_batch): the rows of cached topics are copied from the cache, all the
 * others -- identical (tenant, topic) pairs once -- are matched in ONE launch (bmq_batcher_match_batch) and cached under the epoch rule.
 * out_hit[i] (may be NULL) = 1 if row i came from the cache.  BMQ_E_NOSPACE + *out_needed if out_capacity is short (row pointers are
 * written; what was loaded is cached, so the second call hits).  Requests of bmq_route_cache_config.direct_batch_topics topics and more
 * skip the cache altogether: topics must then be readable 16 bytes past their end, as for bmq_match_batch. */
int bmq_route_cache_get_batch(bmq_route_cache* c, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                              const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, uint64_t now_ms, uint32_t* out_row_ptr,
                              uint32_t* out_route_ids, uint64_t out_capacity, uint64_t* out_needed, uint8_t* out_hit);
int bmq_route_cache_is_cached(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filter, uint32_t filter_len);
int bmq_route_cache_apply(bmq_route_cache* c, const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n);
int bmq_route_cache_rebuild(bmq_route_cache* c, const uint8_t* keys, const uint32_t* key_off, uint32_t n_keys);
int bmq_route_cache_reset(bmq_route_cache* c);
/* Caffeine expires idle entries from a scheduler thread (TenantRouteCache.java:104-114: expireAfterAccess + Scheduler.systemScheduler());
 * here get() drops an expired entry when it meets one and this sweep -- to be called now and then by the owner -- drops the rest: every entry
 * not accessed for expiry_ms at now_ms -- and every TENANT nobody called get for since tenant_idle_ms, whole (SubscriptionCache.java:79-107:
 * the tenant cache expires 2 x the match expiry after its last get; isCached / refresh do not count).  *out_dropped (may be NULL) =
 * entries dropped. */
int bmq_route_cache_expire(bmq_route_cache* c, uint64_t now_ms, uint64_t* out_dropped);
int bmq_route_cache_stats_get(bmq_route_cache* c, bmq_route_cache_stats* out);
/* BMQ_E_STATE: the tenant has no cache (nothing loaded for it yet, or destroyed after tenant_idle_ms) */
int bmq_route_cache_tenant_stats_get(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, bmq_route_cache_tenant_stats* out);
/* The tenant's MaxPersistentFanout / MaxGroupFanout (ISettingProvider.provide(..., tenantId), asked per task loop in
 * DW/cache/TenantRouteCache.java:174-175 and again on every fan-out check, :124-138).  Takes effect with the next get: cached rows the
 * change can affect are re-matched, the others adopt the new caps (MatchedRoutes.adjust).  Survives the tenant cache's expiry. */
int bmq_route_cache_set_caps(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, int32_t max_persistent_fanout,
                             int32_t max_group_fanout);
/* IEventCollector.report(PersistentFanoutThrottled / GroupFanoutThrottled) (DW/cache/MatchedRoutes.java:95-101,124-130): called once
 * per route a load rejects -- type 0 = PersistentFanoutThrottled, 1 = GroupFanoutThrottled; route_id's key gives mqttTopicFilter
 * (bmq_route_key) -- on the thread that completes the load (the caller of get / get_batch, the batcher's dispatcher thread for
 * get_async), before the rows are handed out.  NULL switches reporting off.  Callback and user pointer are replaced together (one atomic
 * pointer to the pair).  May be called while getters run: a load that picked the previous sink up just before the call may still report to
 * it once more afterwards, so a `user` object that was ever installed has to stay valid until bmq_route_cache_destroy has returned. */
typedef void (*bmq_route_cache_event_cb)(void* user, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len,
                                         int32_t type, uint32_t route_id, int32_t max_count);
int bmq_route_cache_set_event_sink(bmq_route_cache* c, bmq_route_cache_event_cb cb, void* user);

/* ---- host-side mirror of MatchedRoutes (fan-out caps in KV order) -------------------------------------- */
/* One ITenantRouteMatcher.matchAll(topics, maxPersistentFanout, maxGroupFanout) call for one tenant,
 * including DW/cache/MatchedRoutes.java:87-141: persistent (subBrokerId == 1) and group fan-out caps applied
 * first-come in key order, throttle events reported.  events: 4 int32 each {type 0=PersistentFanoutThrottled
 * 1=GroupFanoutThrottled, topic index, rejected route id, max count}. */
int bmq_match_all(bmq_engine* e, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics,
                  const uint32_t* topic_off, uint32_t n_topics, int32_t max_persistent_fanout,
                  int32_t max_group_fanout, uint32_t* out_row_ptr, uint32_t* out_route_ids, uint64_t out_capacity,
                  uint64_t* out_needed, int32_t* out_events, uint32_t events_cap, uint32_t* out_n_events);

/* MatchedRoutes (DW/cache/MatchedRoutes.java:87-141) over rows somebody else matched (the route cache caps what the batching front
 * loaded): row r = route_ids[row_ptr[r] .. row_ptr[r + 1]).  out_row_ptr[n_rows + 1] / out_route_ids (at least row_ptr[n_rows] ids) = the
 * rows after the caps, ascending ids; out_class_counts (may be NULL) [2 r] / [2 r + 1] = persistent / group routes kept in row r, or
 * 0xFFFFFFFF when the row was not longer than either cap and therefore never classified; events as in bmq_match_all with the row
 * index in place of the topic index.  Ids of routes deleted since the match are dropped from classified rows. */
int bmq_routes_cap(bmq_engine* e, const uint32_t* row_ptr, const uint32_t* route_ids, uint32_t n_rows, int32_t max_persistent_fanout,
                   int32_t max_group_fanout, uint32_t* out_row_ptr, uint32_t* out_route_ids, uint32_t* out_class_counts,
                   int32_t* out_events, uint32_t events_cap, uint32_t* out_n_events);

/* ---- fan-out grouping: the step behind the match --------------------------------------------------------------- */
/* DistWorkerCoProc.batchDist hands every topic's matched routes to DeliverExecutorGroup.submit (DW/DeliverExecutorGroup.java:112-241);
 * each NormalMatching becomes a DeliveryCall (DW/DeliverExecutor.java:85-90) that the deliverer batches by
 * DelivererKey(subBrokerId, delivererKey) (bifromq-deliverer/.../DelivererKey.java:22, BatchDeliveryCall.java:71-76: per key, tenant ->
 * message pack -> set of MatchInfo).  This is that regrouping for a whole match batch, as a segmented sort of its CSR on the device:
 *   in : row_ptr[n_topics + 1], route_ids[row_ptr[n_topics]]                  (what bmq_match_batch* returned)
 *   out: out_topic[k], out_route[k], k < row_ptr[n_topics]: the (topic index, route id) pairs ordered by (group, topic, route id);
 *        out_group_off[*out_n_groups + 1]: group g owns the pairs out_group_off[g] .. out_group_off[g + 1];
 *        out_group_rep[g]: a route id of the group (bmq_route_key of it gives the group's subBrokerId and delivererKey: the first
 *        and the third NUL-separated part of its receiverUrl, SCHEMA/KVSchemaUtil.java:56-58), 0xFFFFFFFE for the group of
 *        shared-subscription routes (flag 2 / 3: DeliverExecutorGroup.java:243-279 picks the receiver per message), 0xFFFFFFFF for
 *        the group of ids whose route has been deleted since the match.
 * A group = one DelivererKey, exactly (key bytes are compared, not only hashes).  The normal groups come first, in no particular
 * order (the reference's batcher map is a HashMap), then the shared-subscription group (*out_special bit 0), then the dead group
 * (bit 1).  The fan-out caps of submit() are not applied here: the count caps were applied by MatchedRoutes in key order
 * (bmq_match_all), the byte / bandwidth throttles depend on the message.
 * BMQ_E_NOSPACE: pair_cap < row_ptr[n_topics], or group_cap < *out_n_groups (the pairs are written, the group table is not).
 * Works on a host-only engine too (it is not a match: the same per-pair functions run on host threads).
 * _dev: all six array arguments are device pointers (the CSR may be the one bmq_match_batch_dev left in HBM, after
 * bmq_match_finish); runs on the engine stream and returns when the groups are complete. */
int bmq_fanout_group(bmq_engine* e, const uint32_t* row_ptr, const uint32_t* route_ids, uint32_t n_topics, uint32_t* out_topic,
                     uint32_t* out_route, uint64_t pair_cap, uint32_t* out_group_off, uint32_t* out_group_rep, uint32_t group_cap,
                     uint32_t* out_n_groups, uint32_t* out_special);
int bmq_fanout_group_dev(bmq_engine* e, const uint32_t* d_row_ptr, const uint32_t* d_route_ids, uint32_t n_topics, uint64_t total,
                         uint32_t* d_out_topic, uint32_t* d_out_route, uint32_t* d_out_group_off, uint32_t* d_out_group_rep,
                         uint32_t group_cap, uint32_t* out_n_groups, uint32_t* out_special);

/* ---- dist-server side range pruning ---------------------------------------------------------------------- */
/* TenantRangeLookupCache.lookup (bifromq-dist/bifromq-dist-server/src/main/java/org/apache/bifromq/dist/server/scheduler/
 * TenantRangeLookupCache.java:62-109) for a batch of topics of ONE tenant against the tenant's candidate KV ranges in boundary
 * order: which ranges can hold a route whose filter matches the topic, judged -- exactly as the reference does -- from each
 * range's Fact{firstGlobalFilterLevels, lastGlobalFilterLevels} (Fact.proto:27-34) alone.
 *   cand_kind[c]: 0 = the range has no Fact (always kept), 1 = Fact without first/last (empty range: dropped), 2 = first/last given
 *   first / last : packed strings, entry c = the global filter levels (tenant id first) joined by NUL
 *   out_keep[t * n_cand + c] = 1 if candidate c is kept for topic t, else 0 (a candidate behind the point where the reference's
 *   loop stops -- no expansion filter at or after its first filter -- is dropped, as there). */
int bmq_range_lookup(bmq_engine* e, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics, const uint32_t* topic_off,
                     uint32_t n_topics, const uint8_t* cand_kind, const uint8_t* first, const uint32_t* first_off, const uint8_t* last,
                     const uint32_t* last_off, uint32_t n_cand, uint8_t* out_keep);

/* ---- route-key codec (SCHEMA/KVSchemaUtil.java:91-130, SCHEMA/cache/RouteDetailCache.java:53-117) ------- */
/* flag: 1 normal (receiver = receiverUrl), 2 unordered share, 3 ordered share (receiver = group name).
 * filter = MQTT topic filter WITHOUT a $share/$oshare prefix.  Returns key length (writes if <= cap). */
uint32_t bmq_route_key_encode(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filter,
                              uint32_t filter_len, uint8_t flag, const uint8_t* receiver, uint32_t receiver_len,
                              uint8_t* out, uint32_t cap);
/* Decode: spans (offset,len) into `key` for tenant, escaped filter (levels joined by NUL), receiver.
 * Returns flag (1..3) or BMQ_E_INVAL. */
int bmq_route_key_decode(const uint8_t* key, uint32_t key_len, uint32_t spans[6]);
int32_t bmq_java_string_hash(const uint8_t* utf8, uint32_t len);

/* ---- retain store key schema (bifromq-retain/bifromq-retain-store-schema/.../schema/KVSchemaUtil.java:44-73,
 * LevelHash.java:30-50) ---------------------------------------------------------------------------------------------------------- */
/* retainMessageKey(tenantId, topic) = 0x00 | u16be(len tenant) | tenant | u16be(#levels) | LevelHash(levels) | escape(topic);
 * LevelHash = one byte per level (FNV-1a 32 over the level's UTF-16 code units, lowest byte).  Returns the key length. */
uint32_t bmq_retain_message_key(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len, uint8_t* out,
                                uint32_t cap);
/* What MatchCallRangeRouter.rangeLookup derives from ONE topic filter before it consults the range router
 * (bifromq-retain-server/.../scheduler/MatchCallRangeRouter.java:60-134): for a filter without wildcards the exact
 * retainMessageKey; otherwise retainKeyPrefix(tenant, levels, filterPrefix) with levels = level count (a trailing '#' not counted)
 * and filterPrefix = the levels in front of the first wildcard, plus LevelHash(filterPrefix) (the pruning key of findCandidates).
 * Returns a bit mask: 1 = the filter has a wildcard, 2 = it ends with '#' (matches keys of MORE levels too); negative on error. */
int bmq_retain_filter_route(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filter, uint32_t filter_len,
                            uint8_t* out_key_prefix, uint32_t cap, uint32_t* out_key_len, uint8_t* out_level_hash, uint32_t hash_cap,
                            uint32_t* out_hash_len, uint32_t* out_levels);

/* ---- KV range router: the step between a match request and the range replicas that serve it ----------- */
/* The reference's client-side router is a TreeMap<Boundary, KVRangeSetting> in BoundaryUtil.compare order
 * (base-kv/base-kv-type-proto/.../utils/BoundaryUtil.java:142-195); here it is an array of n_ranges boundaries in that order:
 *   range_flags[r]        : bit 0 = the boundary has a start key, bit 1 = it has an end key (an absent side is open; a present key may
 *                           be empty: NULL_BOUNDARY = no start, end "")
 *   start/start_off, end/end_off : packed keys, entry r empty where the side is absent
 * BMQ_E_INVAL if the array is not strictly ascending in that order.
 * KVRangeRouterUtil.findByKey (base-kv/base-kv-store-client/.../client/KVRangeRouterUtil.java:41-52): *out_index = the range that holds
 * `key`, or -1.  KVRangeRouterUtil.findByBoundary (:54-103): the ranges it returns are contiguous: [*out_first, *out_first + *out_count);
 * query_flags as range_flags.  (Where TreeMap.subMap would throw "fromKey > toKey" -- a router that does not cover the query's start --
 * the result is empty.) */
int bmq_router_find_by_key(const uint8_t* range_flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end,
                           const uint32_t* end_off, uint32_t n_ranges, const uint8_t* key, uint32_t key_len, int32_t* out_index);
int bmq_router_find_by_boundary(const uint8_t* range_flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end,
                                const uint32_t* end_off, uint32_t n_ranges, uint8_t query_flags, const uint8_t* q_start, uint32_t q_start_len,
                                const uint8_t* q_end, uint32_t q_end_len, uint32_t* out_first, uint32_t* out_count);
/* MatchCallRangeRouter.rangeLookup(tenantId, topicFilters, effectiveRouter) (bifromq-retain/bifromq-retain-server/.../scheduler/
 * MatchCallRangeRouter.java:56-134) for a batch of topic filters of ONE tenant: out_keep[f * n_ranges + r] = 1 if filter f has to be
 * sent to range r -- a plain topic to the range holding its retainMessageKey, a filter of a fixed level count to the ranges
 * overlapping [retainKeyPrefix, upperBound), a filter ending in '#' to the ranges overlapping [retainKeyPrefix, end of the tenant)
 * minus those findCandidates prunes by LevelHash (:96-133).  BMQ_E_INVAL: a plain topic no range holds (the reference asserts), or a
 * boundary key findCandidates cannot parse (the reference throws).  One deviation: a filter prefix whose LevelHash is all 0xFF bytes
 * makes the reference fail with a NullPointerException (upperBound(levelHash) == null); here its upper bound is open and nothing is
 * pruned by the first rule.
 * mode BMQ_ROUTER_REFERENCE reproduces findCandidates rule for rule.  Those rules look at the LevelHash of a range's start / end key
 * only, as if a range never spanned two level counts: a range [key of 4 levels, key of 6 levels) is pruned for `a/#` when its start
 * key's hash sorts behind hash(a), although every 5-level topic under `a` lives in it (tests/test_router.py shows the case) -- retained
 * messages are then missing from the reply.  mode BMQ_ROUTER_EXACT keeps a range iff it meets one of the key intervals
 * [tenant | L | hash(prefix), tenant | L | upperBound(hash(prefix))), L >= levels, the filter can match in: never a range too few,
 * never one too many. */
#define BMQ_ROUTER_REFERENCE 0u
#define BMQ_ROUTER_EXACT 1u
int bmq_retain_range_lookup(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filters, const uint32_t* filter_off, uint32_t n_filters,
                            const uint8_t* range_flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end,
                            const uint32_t* end_off, uint32_t n_ranges, uint32_t mode, uint8_t* out_keep);

/* The same three lookups over a router OBJECT: the boundaries are copied, checked and indexed once and asked many times (the reference
 * keeps its TreeMap between calls as well and rebuilds it only when the range landscape changes).  n_ranges of bmq_router_retain_lookup's
 * out_keep rows = the n_ranges the router was created with.  A router is immutable: any number of threads may ask it. */
typedef struct bmq_router bmq_router;
int bmq_router_create(const uint8_t* range_flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end,
                      const uint32_t* end_off, uint32_t n_ranges, bmq_router** out);
void bmq_router_destroy(bmq_router* r);
int bmq_router_lookup_key(const bmq_router* r, const uint8_t* key, uint32_t key_len, int32_t* out_index);
int bmq_router_lookup_boundary(const bmq_router* r, uint8_t query_flags, const uint8_t* q_start, uint32_t q_start_len, const uint8_t* q_end,
                               uint32_t q_end_len, uint32_t* out_first, uint32_t* out_count);
int bmq_router_retain_lookup(const bmq_router* r, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filters,
                             const uint32_t* filter_off, uint32_t n_filters, uint32_t mode, uint8_t* out_keep);

/* ---- retain direction (RS/index/IRetainTopicIndex.java:27-35) -------------------------------------------- */
/* A retained-topic id is a STABLE handle, like a route id: a bulk load (bmq_retain_rebuild*, bmq_retain_compact) numbers the topics by
 * the rank of (tenant, level list) in byte order -- tenants in byte order of their ids, a tenant's topics level list by level list, so
 * that every subtree is one id range --; a topic added later by bmq_retain_apply* gets the next unused id; the id of a removed topic
 * goes dead and comes back to life when the topic is retained again.  No id changes or is reused until the next bulk load
 * (bmq_retain_info.generation counts them).
 * Load the retained-topic index: (tenant, topic) pairs.  Replaces the full-scan rebuild in RS/RetainStoreCoProc.java:134-137,279-296.
 * _ex: with what IRetainTopicIndex.add(tenantId, topic, timestamp, expirySeconds) carries (RS/index/IRetainTopicIndex.java:28):
 * timestamp_hlc[i] = Message.timestamp (an HLC: milliseconds << 16 | counter, base-hlc HLC.java:145-151), expiry_seconds[i] =
 * Message.expiryInterval; the message expires at (timestamp_hlc >> 16) + 1000 * expiry_seconds ms (RS/RetainStoreCoProc.java:298-304).
 * Both arrays NULL (or the plain entry points): the topics never expire. */
int bmq_retain_rebuild(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                       const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off,
                       uint32_t n_topics);
int bmq_retain_rebuild_ex(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                          const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                          const uint64_t* timestamp_hlc, const uint32_t* expiry_seconds);
/* IRetainTopicIndex.add / remove (RS/index/RetainTopicIndex.java:126-134; UTIL/index/TopicLevelTrie.java:49-182) as the post-commit
 * closures of RetainStoreCoProc.batchRetain / gc issue them (RS/RetainStoreCoProc.java:240-255,270-275); op[i]: 0 = add (an add of a
 * topic that is there replaces its timestamp / expiry: :246-249), 1 = remove (of an absent topic: a no-op).  The index is mutated
 * WHERE IT LIVES, by three kernels on the engine stream (bmq_retain_core.h): the batch lands behind every match batch launched before
 * it and in front of every later one; matching is never refused while a batch is applied, and the call returns when the device has
 * applied it.  Ops on the same topic inside one batch take effect in order (the last one decides).  A malformed op (op code, tenant
 * index) fails the batch with BMQ_E_INVAL before anything is changed.
 * _batch: ops of several tenants in one call -- op_tenant[i] indexes the tenant table (NULL: every op belongs to tenant 0);
 * out_topic_ids (may be NULL) [i] = the id of op i's topic, 0xFFFFFFFF for the removal of an absent topic or an op that a later op on
 * the same topic superseded. */
int bmq_retain_apply(bmq_engine* e, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics,
                     const uint32_t* topic_off, const uint8_t* op, uint32_t n);
int bmq_retain_apply_ex(bmq_engine* e, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics, const uint32_t* topic_off,
                        const uint8_t* op, const uint64_t* timestamp_hlc, const uint32_t* expiry_seconds, uint32_t n);
int bmq_retain_apply_batch(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* op_tenant,
                           const uint8_t* topics, const uint32_t* topic_off, const uint8_t* op, const uint64_t* timestamp_hlc,
                           const uint32_t* expiry_seconds, uint32_t n, uint32_t* out_topic_ids);
/* Maintenance: fold what bmq_retain_apply* changed into a fresh bulk load (removed topics and their ids go, added topics become ranks
 * again: every subtree one id range, the fast path of '+' and '#').  A new generation of ids. */
int bmq_retain_compact(bmq_engine* e);
/* The same generation change without the stall (round 6): TopicLevelTrie contracts tombed branches as it goes
 * (UTIL/index/TopicLevelTrie.java:257-384); bmq_retain_compact holds the engine for the whole load (2 s for 1 M topics).
 *   bmq_retain_compact_begin   snapshot of the live topics with their stamps (the engine lock for tens of milliseconds);
 *   bmq_retain_compact_build   loads them into an index of their own: the long part, NO engine lock held -- matching and bmq_retain_apply* go
 *                              on meanwhile, what they add / remove is logged;
 *   bmq_retain_compact_swap    uploads the new generation, replays the log in order and makes it the serving one (no batch may be in flight:
 *                              BMQ_E_STATE).  Topic ids are re-numbered: bmq_retain_info.generation + 1, as after bmq_retain_compact.
 *                              *out_carried = topics of the new bulk load, *out_replayed = logged ops replayed.  BMQ_E_NOSPACE: the log outgrew
 *                              1 GiB (abort and begin again; no mutation is ever refused).
 *   bmq_retain_compact_abort   drops the half-built generation.
 * One compaction call at a time; bmq_retain_rebuild* / bmq_retain_compact are refused (BMQ_E_STATE) between begin and swap / abort. */
int bmq_retain_compact_begin(bmq_engine* e);
int bmq_retain_compact_build(bmq_engine* e);
int bmq_retain_compact_swap(bmq_engine* e, uint64_t* out_carried /* may be NULL */, uint64_t* out_replayed /* may be NULL */);
int bmq_retain_compact_abort(bmq_engine* e);
typedef struct bmq_retain_info {
    uint64_t n_topics;        /* retained topics now */
    uint64_t n_tenants;       /* tenants of the last bulk load */
    uint64_t id_bound;        /* every id handed out in this generation is below */
    uint64_t loaded_topics;   /* ids below are ranks of the last bulk load ... */
    uint64_t loaded_removed;  /* ... of which that many have been removed since */
    uint64_t added_ids;       /* ids handed out to topics added since the bulk load */
    uint64_t overlay_nodes;   /* nodes of the overlay trie that holds them */
    uint64_t epoch;           /* +1 per rebuild / apply / compact */
    uint64_t generation;      /* +1 per bulk load (rebuild / compact): ids of different generations are unrelated */
} bmq_retain_info;
int bmq_retain_info_get(const bmq_engine* e, bmq_retain_info* out);
/* id -> the topic the id denotes in this generation (whether or not it is retained right now: bmq_retain_topic_info tells): out
 * receives the tenant id followed by the topic (no separator), *out_tenant_len bytes of tenant, *out_len bytes in total.
 * bmq_retain_topics: many ids at once (bulk-loaded ids from the host's copy of the load, added ones with ONE device gather):
 * out_off[n + 1] byte offsets into out, out_tenant_len[n]; an unknown id gives an empty string. */
int bmq_retain_topic(const bmq_engine* e, uint32_t topic_id, uint8_t* out, uint32_t cap, uint32_t* out_len,
                     uint32_t* out_tenant_len);
int bmq_retain_topics(const bmq_engine* e, const uint32_t* topic_ids, uint32_t n, uint8_t* out, uint64_t cap, uint64_t* out_off,
                      uint32_t* out_tenant_len);
/* id -> the rest of RetainedMsgInfo (RS/index/RetainedMsgInfo.java:29-36) + the expiry instant in ms (~0: never); any out may be NULL.
 * BMQ_E_INVAL: the id was never handed out, or its topic is not retained any more. */
int bmq_retain_topic_info(const bmq_engine* e, uint32_t topic_id, uint64_t* out_timestamp_hlc, uint32_t* out_expiry_seconds,
                          uint64_t* out_expire_at_ms);
/* IRetainTopicIndex.findAll() (RS/index/RetainTopicIndex.java:140-143): *out_n_topics = retained topics now, *out_epoch (may be
 * NULL) counts the retain mutations so far; bmq_retain_live_ids lists their ids, ascending -- of one tenant (every topic of it, '$'
 * ones too) or, tenant == NULL, of all.  Writes up to cap ids, *out_n = total; BMQ_E_NOSPACE if cap was too small. */
int bmq_retain_find_all(const bmq_engine* e, uint64_t* out_n_topics, uint64_t* out_epoch);
int bmq_retain_live_ids(const bmq_engine* e, const uint8_t* tenant, uint32_t tenant_len, uint32_t* out_ids, uint32_t cap, uint32_t* out_n);
/* The scan of RetainStoreCoProc's GC (RS/RetainStoreCoProc.java:257-277): ids (ascending) of the retained topics whose message has
 * expired at now_ms (expireTime <= now) -- of every tenant (tenant == NULL: the reference's findAll() branch), or of `tenant`: there the
 * reference scans index.match(tenantId, "#"), which does not reach topics whose first level starts with '$'
 * (RS/index/RetainTopicIndex.java:60,86,99), and neither does this.  override_expiry_seconds >= 0 replaces the stored expiry interval
 * as GCRequest.expirySeconds does.  One kernel over the ids + a device select.  Writes up to cap ids, *out_n = total; BMQ_E_NOSPACE
 * if cap was too small. */
int bmq_retain_expired(const bmq_engine* e, const uint8_t* tenant, uint32_t tenant_len, uint64_t now_ms,
                       int64_t override_expiry_seconds, uint32_t* out_ids, uint32_t cap, uint32_t* out_n);
/* Batch of IRetainTopicIndex.match(tenant, topicFilter) (RS/index/RetainTopicIndex.java:136-138; selector
 * :36-124; walk UTIL/index/TopicLevelTrie.java:190-249).  Output CSR of topic ids (ascending per row). */
int bmq_retain_match_batch(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                           const uint32_t* filter_tenant, const uint8_t* filters, const uint32_t* filter_off,
                           uint32_t n_filters, uint32_t* out_row_ptr, uint32_t* out_topic_ids,
                           uint64_t out_capacity, uint64_t* out_needed);
/* RetainStoreCoProc.match(tenant, filter, limit, now) (RS/RetainStoreCoProc.java:167-190; limit from RetainStoreCoProc.proto:63,
 * tenant default RetainMessageMatchLimit = 10, Setting.java:77): the reference walks the FULL match set and returns the first
 * `limit` messages that have not expired (expireAt > now).  The order it walks in is a HashSet's (UTIL/index/StrategySet.java:31),
 * i.e. unspecified; here it is ascending topic id: row i = the limit[i] SMALLEST ids among the matching topics that are still
 * live at now_ms (fewer if fewer are live; limit 0 -> empty row).  out_match_count[i] (may be NULL) = number of topics filter i
 * matches, expired ones included.  With every limit <= 64 nothing is expanded: the ids are picked from the matched id ranges.
 * Same buffer protocol as bmq_retain_match_batch. */
int bmq_retain_match_limited(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                             const uint32_t* filter_tenant, const uint8_t* filters, const uint32_t* filter_off,
                             uint32_t n_filters, const uint32_t* limit, uint64_t now_ms, uint32_t* out_row_ptr,
                             uint32_t* out_topic_ids, uint64_t out_capacity, uint64_t* out_needed, uint32_t* out_match_count);
int bmq_retain_match_batch_dev(bmq_engine* e, const uint8_t* d_tenants, const uint32_t* d_tenant_off,
                               uint32_t n_tenants, const uint32_t* d_filter_tenant, const uint8_t* d_filters,
                               const uint32_t* d_filter_off, uint32_t n_filters, uint32_t* d_out_row_ptr,
                               uint32_t* d_out_topic_ids, uint64_t out_capacity, uint64_t* d_out_total);

#ifdef __cplusplus
}
#endif
#endif /* BMQ_H */
