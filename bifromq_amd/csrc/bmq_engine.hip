// bmq_engine.hip -- the engine behind include/bmq.h: owns the HBM-resident index, the per-batch scratch and the
// HIP stream; launches the gfx950 kernels of bmq_dist_kernels.h / bmq_retain_kernels.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared (see __graft_entry__.build()).
//
// There is deliberately NO CPU implementation of the match path in this library: without a device every match
// entry point returns BMQ_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "../../include/bmq.h"
#include "bmq_codec.h"
#include "bmq_dist_index.h"
#include "bmq_dist_kernels.h"
#include "bmq_poll_kernel.h"
#include "bmq_exec_dev.h"
#include "bmq_exec_host.h"
#include "bmq_fanout.h"
#include "bmq_format_kernels.h"
#include "bmq_range_core.h"
#include "bmq_retain.h"
#include "bmq_retain_dyn.h"
#include "bmq_retain_kernels.h"

using namespace bmq;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // grow-only; contents are NOT preserved
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        release();
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            want = bytes;
            e = hipMalloc(&p, want);
            if (e != hipSuccess) {
                p = nullptr;
                return e;
            }
        }
        cap = want;
        return hipSuccess;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct RetainDevice { // the bulk-loaded part of the retained-topic index in HBM (the mutable part: RetainDyn, bmq_retain_dyn.h)
    DevBuf nodes, edges, posts, gps, tenants, dict, pool;
    RetainIndexView view{};
};
struct RetainLimit { // bmq_retain_match_limited in flight: select the first `limit` live ids from the ranges instead of expanding
    bool active = false;
    const uint32_t* d_limit = nullptr;
    uint64_t now = 0;
    uint32_t* d_tmp = nullptr;
    uint32_t* d_kept = nullptr;
    uint32_t* d_counts = nullptr;
};

} // namespace

struct bmq_engine {
    bmq_config cfg{};
    int device = -1;
    hipStream_t stream = nullptr;
    std::mutex mu;             // protects engine state inside one entry point
    mutable std::recursive_mutex api;  // held for the whole duration of the host-buffer entry points (they are multi-step)
    std::string err;

    // dist direction: the index lives in HBM and is built / mutated there by the builder kernels (bmq_exec_dev.h); a host-only
    // engine (device < 0) runs the same builder on host threads (bmq_exec_host.h) for inspection -- it can never match
    DevExec dx;
    DevExec dxi[2]; // the route index's executors: one under the serving generation (engine stream), the other free -- or, while a generation
                    // is being built beside it (bmq_compact_begin), under that one, on the build stream; the swap exchanges the roles
    std::unique_ptr<DistIndex<DevExec>> dix;
    HostExec hx;
    std::unique_ptr<DistIndex<HostExec>> hix;
    uint64_t epoch = 0;
    bool built = false;
    bool slow_on = false, sort_on = false; // the repair kernels are in the pipeline (see launch_dist)
    uint32_t slow_idle = 0, sort_idle = 0; // consecutive batches that ran them for nothing
    DevBuf dd_table;                       // in-batch de-duplication (bmq_dedup_kernels.h): one table for all batch slots (batches run in stream order)
    uint32_t dd_gen = 0;                   // generation of the last batch that used it (1..255; 0: fresh / just zeroed)
    uint32_t dedup_min = 0xFFFFFFFFu;      // batches of at least this many topics are de-duplicated (bmq_config.dedup_min_topics; default: never)
    int publish_mode = 1;                  // a batch's counters reach the host through k_publish: 0 never (hipMemcpyAsync), 1 launches of fewer than 4096 topics, 2 always (BMQ_PUBLISH_KERNEL)
    bool dedup_sorted = false;             // ... by comparing neighbours: the caller's batches are ordered by (tenant, topic) (bmq_config.dedup_sorted)
    bool mixed_on = false;                 // k_walk runs in its MIXED instantiation (batches are not grouped by tenant)
    uint32_t mixed_idle = 0;
    int walk_geom = 0;                     // LDS geometry of k_walk: 0 default, 2 smallest lists (bmq_config caps <= 128)
    double split_mult = EXPAND_SPLIT_MULT_X16 / 16.0; // heavy = this many mean blocks (finish_dist raises it while the list overflows)
    double blk_mean_ranges = 0, blk_mean_ids = 0; // per 64-row block of the large dist batches so far (k_expand's heavy blocks: launch_dist)
    bool kernel_events = false; // bmq_config.kernel_timing: HIP events around k_walk / k_expand of every dist batch (~4 us each)

    // Everything ONE batch in flight needs: per-batch scratch, staging of the host-buffer API, counters, events.  The asynchronous
    // host API (bmq_match_submit / bmq_match_wait) owns two slots of its own, so that the upload of batch i+1 and the download of
    // batch i-1 are in flight while the kernels of batch i run; every other entry point works on slot 0.
    struct BatchSlot {
        DevBuf b_subs, b_super, b_blk_stats, b_dbg_wave, b_rep, b_visit, b_heavy;
        // de-duplication of an ordered batch (bmq_dedup_adj_kernels.h): per-block sums, the dense batch of run heads and its per-row results
        DevBuf b_drow, b_adj_cnt, b_adj_mask, b_adj_super, b_c_topics, b_c_off, b_c_tenant, b_c_rep, b_c_pair_off, b_c_pair_cnt, b_c_route_cnt;
        uint64_t adj_cap = 0; // bytes of b_c_topics in use as the dense batch's topic bytes (grown on ST_NEED_ADJ)
        DevBuf b_pair_off, b_pair_cnt, b_route_cnt, b_pairs, b_spill, b_wave_sums, b_slow_list, b_scratch, b_sort_list, b_ctr,
            b_total;
        uint64_t pair_cap = 0, scratch_cap = 0, spill_cap = 0;
        uint32_t slow_cap = 0, sort_cap = 0;
        Counters* h_ctr = nullptr; // pinned
        Counters* d_h_ctr = nullptr; // ... as the device addresses it (k_publish)
        // staging for the host-buffer API
        DevBuf s_tenants, s_tenant_off, s_topic_tenant, s_topics, s_topic_off, s_row_ptr, s_ids, s_lim, s_lim_ids, s_lim_tmp;
        hipEvent_t ev[8]{};
        hipEvent_t ev_in = nullptr, ev_done = nullptr; // inputs uploaded / batch (kernels + counter read-back) complete
        // the batch in flight (for bmq_match_finish / bmq_match_wait)
        bool pending = false;
        bool clean = false; // counters / allocators / super sums are zero (k_reset ran behind the last batch of the slot)
        bool ran_slow = false, ran_sort = false; // k_walk_slow / k_sort_rows were part of this batch's launch
        bool ran_mixed = false;                  // k_walk ran in its MIXED instantiation
        bool ran_rdeep = false;                  // retain: k_retain_walk_deep was part of this batch's launch
        bool timed = false; // this batch was launched with the per-kernel events (bmq_config.kernel_timing)
        bool total_timed = false; // ... with the two events around the whole batch (bmq_stats.ms_total)
        int pending_kind = 0; // 0 dist, 1 retain
        bool submitted = false; // owned by a bmq_match_submit ticket
        bool api_held = false;  // the *_dev launch took the engine's api lock; bmq_match_finish gives it back
        uint64_t dev_cap = 0;   // ids the slot's own output buffer holds
        uint32_t n_rows = 0;
        BatchArgs last{};
        RetainArgs rlast{};
        // result formats of bmq_match_submit_fmt (bmq_formats.inc)
        int format = 0;                                 // BMQ_FMT_*
        bool devptr = false;                            // a bmq_match_submit_dev ticket: the caller's buffers, nothing staged
        DevBuf f_cnt, f_ptr, f_ranges, f_side, f_sums;  // RANGES: counts, their prefix sums, the ranges, the side ids, sizes / flags
        uint64_t range_cap = 0, side_cap = 0;
        unsigned long long* h_fsums = nullptr;          // pinned copy of f_sums
        hipEvent_t ev_fmt = nullptr;                    // the format kernels (and the read-back of their sizes) are complete
        DevBuf g_pairs, g_groups;                       // GROUPED: out_topic | out_route, group_off | group_rep
    };
    // slots[0] (== cur, never re-pointed) serves the blocking / *_dev entry points; slots[1 ..] are the BMQ_MAX_TICKETS tickets of
    // bmq_match_submit / bmq_match_wait, so that a ticket in flight is never staged over by another thread's blocking call.
    BatchSlot slots[1 + BMQ_MAX_TICKETS];
    BatchSlot* const cur = &slots[0];
    hipStream_t s_in = nullptr, s_out = nullptr; // copy streams of the asynchronous host API
    bmq_stats stats{};

    // retain direction
    RetainIndexHost rhost;
    RetainDevice rdev;
    std::unique_ptr<RetainDyn<DevExec>> drt;  // dead bitmap + overlay trie + per-id payload, mutated on the device
    std::unique_ptr<RetainDyn<HostExec>> hrt; // ... of a host-only engine (inspection, sanitizer fuzzers)
    RetainIndexView rhview{};                 // host-only engine: the bulk-loaded arrays where RetainIndexHost keeps them
    bool rbuilt = false;
    // bmq_retain_compact_begin / _build / _swap: the next generation of the retained-topic index, built beside the serving one (round 6)
    struct RetainCompaction {
        bool active = false;  // (guarded by mu, like the log; `next` and `items` belong to the one thread that drives the compaction)
        bool built = false;
        std::vector<RetainIndexHost::Item> items; // the live topics at begin, with their stamps
        RetainIndexHost next;                     // ... loaded into an index of their own, outside the engine lock
        struct Op {
            std::string tenant, topic;
            uint8_t op;
            bool has_ts;
            unsigned long long ts;
            uint32_t expiry;
        };
        std::vector<Op> log;                      // IRetainTopicIndex.add / remove since begin, in order: replayed before the swap
        uint64_t log_bytes = 0;
        bool log_overflow = false;
    } rcmp;
    uint64_t repoch = 0;      // +1 per retain rebuild / apply / compact
    uint64_t rgeneration = 0; // +1 per retain rebuild / compact: topic ids of different generations are unrelated
    RetainLimit rlim;
    DevBuf r_scratch, r_deep_list, r_deep_levels, r_arena;
    DevBuf r_ov_off, r_ov_cnt, r_ov_route, r_ov_pairs; // k_retain_overlay's range list
    uint64_t ov_pair_cap = 0;
    hipStream_t s_side = nullptr;                      // ... runs beside k_retain_walk
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool rdeep_on = false;   // k_retain_walk_deep is in the pipeline (batches hold filters of more than R_MAXL levels)
    uint32_t rdeep_idle = 0;
    DevBuf range_buf; // staging of bmq_range_lookup
    // fan-out grouping (bmq_fanout.h): group table + per-route cache, created by the first call
    std::unique_ptr<Fanout<DevExec>> dfo;
    std::unique_ptr<Fanout<HostExec>> hfo;
    DevBuf fo_buf; // staging of bmq_fanout_group
    // multi-GPU exchange inside the library (bmq_exchange.inc): RCCL communicator of this rank, its own stream
    void* comm = nullptr;
    int comm_world = 0, comm_rank = 0;
    hipStream_t s_ex = nullptr;
    hipEvent_t ev_ex = nullptr;
    DevBuf ex_buf;
    DevBuf part_buf; // bmq_partition_batch_dev: flags, lengths and their prefix sums
    uint32_t rgcap = 0;
    uint32_t rwcap = 0;      // k_retain_walk: entries per frontier list in the waves' arena (grown on ST_RETAIN_LIST)
    uint32_t rw_waves = 0;   // k_retain_walk: resident waves (its persistent grid)
    // bmq_compact_begin / _poll / _swap: the next generation of the route index, built beside the serving one from its live keys
    struct Compaction {
        bool active = false; // (guarded by mu, like the log; the rest by cmp_mu)
        DevExec* bx = nullptr; // the executor under the generation being built
        std::unique_ptr<DistIndex<DevExec>> next_d; // the generation being built: on the device,
        std::unique_ptr<DistIndex<HostExec>> next_h; // or (host-only engine) in host memory, like the serving one
        uint32_t cursor = 0, n_ids = 0; // ids of the serving generation handed over so far / to hand over
        uint64_t carried = 0;           // live keys handed over
        std::vector<uint8_t> log_keys;  // what was mutated meanwhile: replayed into `next` before the swap
        std::vector<uint32_t> log_off{0};
        std::vector<uint8_t> log_op;
        // the batch whose outcome is not known yet (between apply_begin and complete_apply): it enters the log only once apply_end has
        // accepted it -- a batch refused for a malformed key, or one that fails on the device, must not be replayed
        std::vector<uint8_t> pend_keys, pend_op;
        std::vector<uint32_t> pend_off;
        bool log_overflow = false;      // the log outgrew LOG_CAP_BYTES: this compaction cannot be swapped any more (abort + begin again)
        static constexpr uint64_t LOG_CAP_BYTES = 1ull << 30;
    } cmp;
    std::mutex cmp_mu;                // one compaction call at a time (taken BEFORE mu)
    hipStream_t s_build = nullptr;    // the stream the next generation is built on: lowest priority, beside the match batches
    hipEvent_t ev_serving = nullptr;  // "what the serving generation was told so far": the build stream waits for it before a snapshot
    bool apply_open = false; // bmq_routes_apply_async: the batch's outcome has not been fetched yet (complete_apply)
    // the persistent matcher of the batching front (bmq_poll_kernel.h, bmq_poller.inc)
    struct Poller {
        bool enabled = true;   // bmq_batcher_config.persistent_matcher
        bool running = false;  // k_poll was launched and has not been waited for (guarded by mu, like everything here but the ring words)
        bool broken = false;   // a generation timed out: never started again on this engine
        uint32_t yield_every = 64; // the waiting leader yields its CPU after this many polls of the completion word (0: never; BMQ_POLL_YIELD: experiments): with four
                                   // callers per CPU the followers of other generations get the core -- 64 blocked callers 0.76 -> 0.80 M calls/s (medians of four runs), 16 unchanged
        hipStream_t stream = nullptr;
        hipEvent_t ev_index = nullptr; // "the index is complete": recorded on the engine stream, waited for by the poller's
        PollDesc* desc = nullptr; // page-locked, POLL_SLOTS
        PollDone* done = nullptr;
        PollCtl* ctl = nullptr;
        uint8_t *blob_in = nullptr, *blob_out = nullptr; // page-locked: one input and one output blob per slot (pinned before any launch)
        DevBuf scratch;           // every per-wave array of PollArgs, one allocation
        PollArgs args{};
        bool busy[POLL_SLOTS] = {};
        uint32_t rr = 0;          // slot hand-out: round robin over the waves
        uint64_t n_served = 0, n_fallback = 0, n_timeouts = 0, n_starts = 0, n_unserved = 0;
        std::atomic<uint64_t> n_bad_input{0}; // generations a wave refused to touch (offsets / tenant indices out of range: never seen outside bring-up)
    } pol;
};

static int retain_finish(bmq_engine* e, uint64_t* out_total);
static void poller_stop_locked(bmq_engine* e);
extern "C" void bmq_comm_destroy(bmq_engine* e);

static int complete_apply(bmq_engine* e); // (defined behind the index helpers below)
namespace {

#define HIPCHK(e, expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _err = (expr);                                                                        \
        if (_err != hipSuccess) {                                                                        \
            (e)->err = std::string(#expr) + ": " + hipGetErrorString(_err);                              \
            return BMQ_E_HIP;                                                                            \
        }                                                                                                \
    } while (0)

int set_err(bmq_engine* e, int code, const std::string& msg) {
    e->err = msg;
    return code;
}

int upload(bmq_engine* e, DevBuf& b, const void* src, size_t bytes) {
    HIPCHK(e, b.ensure(bytes ? bytes : 16));
    if (bytes) HIPCHK(e, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, e->stream));
    return BMQ_OK;
}

// ---- dist batch ------------------------------------------------------------------------------------------------
// topics per k_walk / k_expand wave = 2^shift: small batches are spread over more, shorter waves (see k_walk)
uint32_t tpw_shift_for(uint32_t n_topics) {
    if (const char* v = bmq_env("BMQ_TPW_SHIFT")) return (uint32_t)std::min(6, std::max(0, atoi(v))); // profiling experiments
    return n_topics >= 131072 ? 6u : (n_topics >= 16384 ? 4u : 2u); // measured: profiles/r02/extras/tpw_sweep.txt (10 k topics: 0.117 / 0.092 / 0.084 ms)
}

int ensure_batch_scratch(bmq_engine* e, bmq_engine::BatchSlot& S, uint32_t n_tenants, uint32_t n_topics) {
    const uint32_t sh = tpw_shift_for(n_topics);
    const uint32_t n_blocks = (n_topics + (1u << sh) - 1) >> sh;
    if (S.pair_cap == 0) S.pair_cap = 1u << 16;
    S.pair_cap = std::max<uint64_t>(S.pair_cap, (uint64_t)n_topics * 4);
    if (S.pair_cap >= 0xFFFFFFFFull) return set_err(e, BMQ_E_RANGE, "matched-range buffer exceeds 2^32 entries");
    if (S.slow_cap == 0) S.slow_cap = 1024;
    S.slow_cap = std::max<uint32_t>(S.slow_cap, n_topics / 16);
    if (S.sort_cap == 0) S.sort_cap = 1024;
    S.sort_cap = std::max<uint32_t>(S.sort_cap, n_topics / 64);
    if (S.scratch_cap == 0) S.scratch_cap = (uint64_t)(e->cfg.slow_scratch_mb ? e->cfg.slow_scratch_mb : 64) * (1u << 20) / 4;
    HIPCHK(e, S.b_pair_off.ensure(sizeof(uint32_t) * std::max(n_topics, 1u)));
    HIPCHK(e, S.b_pair_cnt.ensure(sizeof(uint32_t) * std::max(n_topics, 1u)));
    HIPCHK(e, S.b_route_cnt.ensure(sizeof(uint32_t) * std::max(n_topics, 1u)));
    HIPCHK(e, S.b_pairs.ensure(sizeof(MatchRange) * S.pair_cap));
    const void *p_subs = S.b_subs.p, *p_super = S.b_super.p, *p_ctr = S.b_ctr.p;
    HIPCHK(e, S.b_subs.ensure(sizeof(SubAlloc) * 2 * N_SUB));
    HIPCHK(e, S.b_super.ensure(sizeof(unsigned long long) * SUPER_STRIDE * ((n_blocks >> SUPER_SHIFT) + 2)));
    if (S.spill_cap == 0) S.spill_cap = 1u << 16;
    S.spill_cap = std::max<uint64_t>(S.spill_cap, (uint64_t)n_topics * 2);
    HIPCHK(e, S.b_blk_stats.ensure(sizeof(uint4) * std::max(n_blocks, 1u)));
    HIPCHK(e, S.b_spill.ensure(sizeof(uint4) * S.spill_cap));
    HIPCHK(e, S.b_wave_sums.ensure(sizeof(unsigned long long) * std::max(n_blocks, 1u)));
    HIPCHK(e, S.b_slow_list.ensure(sizeof(uint32_t) * S.slow_cap));
    HIPCHK(e, S.b_sort_list.ensure(sizeof(uint32_t) * S.sort_cap));
    HIPCHK(e, S.b_scratch.ensure(sizeof(uint32_t) * S.scratch_cap));
    HIPCHK(e, S.b_ctr.ensure(sizeof(Counters)));
    HIPCHK(e, S.b_total.ensure(sizeof(unsigned long long)));
    if (p_subs != S.b_subs.p || p_super != S.b_super.p || p_ctr != S.b_ctr.p) S.clean = false; // fresh memory
    return BMQ_OK;
}

constexpr uint32_t REPAIR_IDLE_BATCHES = 32;

// zero state of a batch slot's counters, range allocators and super-block sums (k_reset), whole capacity
void reset_slot(bmq_engine* e, bmq_engine::BatchSlot& S, hipStream_t s) {
    const uint32_t n_super = (uint32_t)(S.b_super.cap / (sizeof(unsigned long long) * SUPER_STRIDE));
    const uint32_t items = std::max<uint32_t>(n_super, (uint32_t)(sizeof(SubAlloc) * 2 * N_SUB / 8));
    hipLaunchKernelGGL(k_reset, dim3((items + 63) / 64), dim3(64), 0, s, S.b_ctr.as<Counters>(), S.b_subs.as<SubAlloc>(),
                       S.b_super.as<unsigned long long>(), n_super);
    S.clean = true;
}

// RANGES format: k_fmt_count -> two prefix sums -> k_fmt_emit on the engine stream (bmq_format_kernels.h); sizes into S.h_fsums
int enqueue_ranges(bmq_engine* e, bmq_engine::BatchSlot& S, const BatchArgs& a) {
    const size_t n1 = (size_t)a.n_topics + 1;
    HIPCHK(e, S.f_cnt.ensure(sizeof(uint32_t) * 2 * n1));
    HIPCHK(e, S.f_ptr.ensure(sizeof(uint32_t) * 2 * n1));
    HIPCHK(e, S.f_sums.ensure(4 * sizeof(unsigned long long)));
    if (S.range_cap < (uint64_t)a.n_topics * 8 + 1024) S.range_cap = (uint64_t)a.n_topics * 8 + 1024;
    if (S.side_cap < 65536) S.side_cap = 65536;
    HIPCHK(e, S.f_ranges.ensure(sizeof(MatchRange) * S.range_cap));
    HIPCHK(e, S.f_side.ensure(sizeof(uint32_t) * S.side_cap));
    FmtArgs f{};
    f.ix = a.ix;
    f.pair_off = a.pair_off;
    f.pair_cnt = a.pair_cnt;
    f.pairs = a.pairs;
    f.ctr = a.ctr;
    f.n_topics = a.n_topics;
    f.cnt_r = S.f_cnt.as<uint32_t>();
    f.cnt_s = f.cnt_r + n1;
    f.range_ptr = S.f_ptr.as<uint32_t>();
    f.side_ptr = f.range_ptr + n1;
    f.out_ranges = S.f_ranges.as<MatchRange>();
    f.range_cap = S.range_cap;
    f.out_side = S.f_side.as<uint32_t>();
    f.side_cap = S.side_cap;
    f.sums = S.f_sums.as<unsigned long long>();
    hipStream_t s = e->stream;
    const dim3 grid((unsigned)((n1 + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_fmt_count, grid, block, 0, s, f);
    size_t bytes = 0;
    HIPCHK(e, hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, f.cnt_r, f.range_ptr, (int)n1, s));
    if (!e->dx.ensure_tmp(bytes)) return set_err(e, BMQ_E_NOMEM, e->dx.err);
    HIPCHK(e, hipcub::DeviceScan::ExclusiveSum(e->dx.tmp, bytes, f.cnt_r, f.range_ptr, (int)n1, s));
    HIPCHK(e, hipcub::DeviceScan::ExclusiveSum(e->dx.tmp, bytes, f.cnt_s, f.side_ptr, (int)n1, s));
    hipLaunchKernelGGL(k_fmt_emit, grid, block, 0, s, f);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipMemcpyAsync(S.h_fsums, f.sums, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipEventRecord(S.ev_fmt, s));
    return BMQ_OK;
}

int launch_dist(bmq_engine* e, bmq_engine::BatchSlot& S, BatchArgs& a) {
    if (int rc = complete_apply(e)) return rc; // (an apply batch still open: the launch needs the index as it leaves it)
    a.ix = e->dix->view();
    a.tpw_shift = tpw_shift_for(a.n_topics);
    a.n_blocks = (a.n_topics + (1u << a.tpw_shift) - 1) >> a.tpw_shift;
    a.pair_off = S.b_pair_off.as<uint32_t>();
    a.pair_cnt = S.b_pair_cnt.as<uint32_t>();
    a.route_cnt = S.b_route_cnt.as<uint32_t>();
    a.pairs = S.b_pairs.as<MatchRange>();
    a.pair_cap = S.pair_cap;
    a.subs = S.b_subs.as<SubAlloc>();
    a.super_sums = S.b_super.as<unsigned long long>();
    a.blk_stats = S.b_blk_stats.as<uint4>();
    a.spill = S.b_spill.as<uint4>();
    a.spill_cap = S.spill_cap;
    a.wave_sums = S.b_wave_sums.as<unsigned long long>();
    a.slow_list = S.b_slow_list.as<uint32_t>();
    a.slow_cap = S.slow_cap;
    a.scratch = S.b_scratch.as<uint32_t>();
    a.scratch_cap = S.scratch_cap;
    a.sort_list = S.b_sort_list.as<uint32_t>();
    a.sort_cap = S.sort_cap;
    a.ctr = S.b_ctr.as<Counters>();
    {
        const char* dbg = bmq_env("BMQ_DEBUG");
        a.debug_flags = (BMQ_EXPERIMENTS && dbg) ? (uint32_t)atoi(dbg) : 0u; // (the kernels' experiments exist in -DBMQ_EXPERIMENTS=1 builds only)
        a.dbg_wave = nullptr;
        if (a.debug_flags & 30u) {
            HIPCHK(e, S.b_dbg_wave.ensure(sizeof(uint4) * 2 * std::max(a.n_blocks, 1u)));
            a.dbg_wave = S.b_dbg_wave.as<uint4>();
        }
    }
    a.qcap = e->cfg.wave_queue_cap;
    a.pcap = e->cfg.wave_pair_cap;
    a.rep = nullptr, a.visit_cnt = nullptr, a.dd_table = nullptr, a.dd_mask = 0, a.dd_gen = 0;
    // k_expand's heavy blocks (bmq_batch_args.h): listed by whoever writes the per-block sums, expanded by four waves each.  Large batches only:
    // a small one has no tail to speak of, and its waves own fewer than 64 rows.
    a.heavy_list = nullptr, a.heavy_cap = 0, a.split_ranges = a.split_ids = 0xFFFFFFFFu;
    if (a.tpw_shift == 6 && a.n_blocks >= 1024) {
        a.heavy_cap = a.n_blocks / EXPAND_HEAVY_DIV;
        // heavy = EXPAND_SPLIT_MULT_X16 / 16 x the mean block of the large batches so far (finish_dist keeps the means)
        a.split_ranges = e->blk_mean_ranges > 0 ? (uint32_t)std::min(4e9, std::max(64.0, e->blk_mean_ranges * e->split_mult)) : EXPAND_SPLIT_RANGES;
        a.split_ids = e->blk_mean_ids > 0 ? (uint32_t)std::min(4e9, std::max(256.0, e->blk_mean_ids * e->split_mult)) : EXPAND_SPLIT_IDS;
        HIPCHK(e, S.b_heavy.ensure(sizeof(uint32_t) * a.heavy_cap));
        a.heavy_list = S.b_heavy.as<uint32_t>();
    }
    const bool adj = a.n_topics >= e->dedup_min && e->dedup_sorted; // an ordered batch: equal rows are neighbours (bmq_dedup_adj_kernels.h)
    BatchArgs a2{};  // adj: the dense batch of run heads the walk kernels run on
    AdjArgs g{};
    AdjFill gf{};
    if (adj) {
        const size_t n = a.n_topics, nb = a.n_blocks, n_super = ((nb - 1) >> SUPER_SHIFT) + 1;
        if (S.adj_cap == 0) S.adj_cap = 48 * n + 4096; // (a first guess: the scatter kernel asks for more -- ST_NEED_ADJ -- and the batch runs again)
        HIPCHK(e, S.b_visit.ensure(4 * n));
        HIPCHK(e, S.b_drow.ensure(4 * n));
        HIPCHK(e, S.b_adj_cnt.ensure(8 * nb));
        HIPCHK(e, S.b_adj_mask.ensure(8 * nb));
        HIPCHK(e, S.b_adj_super.ensure(8 * SUPER_STRIDE * n_super));
        HIPCHK(e, S.b_c_topics.ensure(S.adj_cap + 64));
        HIPCHK(e, S.b_c_off.ensure(4 * (n + 1)));
        HIPCHK(e, S.b_c_tenant.ensure(4 * n));
        HIPCHK(e, S.b_c_rep.ensure(4 * n));
        HIPCHK(e, S.b_c_pair_off.ensure(4 * n));
        HIPCHK(e, S.b_c_pair_cnt.ensure(4 * n));
        HIPCHK(e, S.b_c_route_cnt.ensure(4 * n));
        g.topics = a.topics, g.topic_off = a.topic_off, g.topic_tenant = a.topic_tenant;
        g.n_topics = a.n_topics, g.n_blocks = a.n_blocks, g.tpw_shift = a.tpw_shift;
        g.drow = S.b_drow.as<uint32_t>();
        g.blk_cnt = S.b_adj_cnt.as<unsigned long long>(), g.blk_mask = S.b_adj_mask.as<unsigned long long>(), g.super_cnt = S.b_adj_super.as<unsigned long long>();
        g.c_topics = S.b_c_topics.as<uint8_t>(), g.c_cap = S.adj_cap;
        g.c_off = S.b_c_off.as<uint32_t>(), g.c_tenant = S.b_c_tenant.as<uint32_t>(), g.c_rep = S.b_c_rep.as<uint32_t>();
        g.ctr = a.ctr;
        a.rep = g.drow; // (only its being there matters to the kernels behind the walk: a de-duplicated batch, k_fill_adj writes the per-block sums)
        a.visit_cnt = S.b_visit.as<uint32_t>();
        a2 = a;
        a2.topics = g.c_topics, a2.topic_off = g.c_off, a2.topic_tenant = g.c_tenant, a2.rep = g.c_rep;
        a2.pair_off = S.b_c_pair_off.as<uint32_t>(), a2.pair_cnt = S.b_c_pair_cnt.as<uint32_t>(), a2.route_cnt = S.b_c_route_cnt.as<uint32_t>();
        gf.drow = g.drow, gf.c_pair_off = a2.pair_off, gf.c_pair_cnt = a2.pair_cnt, gf.c_route_cnt = a2.route_cnt, gf.c_visit = a2.visit_cnt;
    } else if (a.n_topics >= e->dedup_min) { // identical (tenant, topic) rows are walked once
        uint32_t cap = 1024;
        while (cap < 2 * a.n_topics && cap < (1u << 31)) cap <<= 1;
        if (e->dd_table.cap < sizeof(unsigned long long) * cap) e->dd_gen = 0; // (a new table: to be zeroed)
        HIPCHK(e, e->dd_table.ensure(sizeof(unsigned long long) * cap));
        cap = 1024; // the table in use = the largest power of two the buffer holds (it only grows)
        while ((size_t)cap * 2 * sizeof(unsigned long long) <= e->dd_table.cap && cap < (1u << 31)) cap <<= 1;
        if (e->dd_gen == 0 || e->dd_gen == 255) { // entries of other generations count as free; the 8-bit generation wrapped: start over
            HIPCHK(e, hipMemsetAsync(e->dd_table.p, 0, sizeof(unsigned long long) * cap, e->stream));
            e->dd_gen = 0;
        }
        e->dd_gen++;
        HIPCHK(e, S.b_rep.ensure(sizeof(uint32_t) * a.n_topics));
        HIPCHK(e, S.b_visit.ensure(sizeof(uint32_t) * a.n_topics));
        a.rep = S.b_rep.as<uint32_t>(), a.visit_cnt = S.b_visit.as<uint32_t>();
        a.dd_table = e->dd_table.as<unsigned long long>(), a.dd_mask = cap - 1, a.dd_gen = e->dd_gen;
    }
    const BatchArgs& w = adj ? a2 : a; // what the walk kernels run on
    hipStream_t s = e->stream;
    S.timed = e->kernel_events;
    if (!S.clean) reset_slot(e, S, s); // first batch of the slot, buffers regrown, or a retain batch ran on it
    // ms_total: two more events per batch -- ~8 us of host time, which a launch of a few hundred topics (the batching front's) feels
    S.total_timed = e->kernel_events || a.n_topics >= 4096;
    // Per-kernel times (bmq_config.kernel_timing) need no event of their own where a neighbouring one marks the same instant: ev[5] stands
    // behind k_expand unless a repair kernel runs in between, ev[2] in front of it -- four events per batch instead of six (~4 us each on
    // the stream, measured).  k_walk keeps its own start event BEHIND ev[0]: the first packet after an idle stream is stamped before the
    // queue has woken up, and ev[0] -> ev[2] read 6 us more than the kernel's own duration (rocprofv3), ev[1] -> ev[2] agrees with it.
    if (S.total_timed) HIPCHK(e, hipEventRecord(S.ev[0], s));
    if (adj) { // (inside ms_total, in front of ms_walk)
        HIPCHK(e, hipMemsetAsync(S.b_adj_super.p, 0, 8 * SUPER_STRIDE * ((((size_t)a.n_blocks - 1) >> SUPER_SHIFT) + 1), s));
        hipLaunchKernelGGL(k_dd_adj_heads, dim3(a.n_blocks), dim3(64), 0, s, g);
        hipLaunchKernelGGL(k_dd_adj_scatter, dim3(a.n_blocks), dim3(64), 0, s, g);
    } else if (a.rep) hipLaunchKernelGGL(k_dedup, dim3(a.n_blocks), dim3(64), 0, s, a);
    if (e->kernel_events) HIPCHK(e, hipEventRecord(S.ev[1], s));
    {
        // k_walk<token table entries, stack items, range entries, MIXED>: the LDS geometry is a compile-time property (bmq_walk_kernel.h)
        const dim3 grid((a.debug_flags & 32u) ? ((a.n_blocks + 7u) / 8u) * 8u : a.n_blocks), block(64);
        S.ran_mixed = e->mixed_on;
        const int g = e->walk_geom;
#define BMQ_WALK_LAUNCH(...) BMQ_WALK_LAUNCH3(__VA_ARGS__)
#define BMQ_WALK_LAUNCH3(TC, QC, PC)                                                       \
    do {                                                                                   \
        if (e->mixed_on) hipLaunchKernelGGL((k_walk<TC, QC, PC, true>), grid, block, 0, s, w); \
        else hipLaunchKernelGGL((k_walk<TC, QC, PC, false>), grid, block, 0, s, w);            \
    } while (0)
#if BMQ_EXPERIMENTS
        if (a.debug_flags & 16u) hipLaunchKernelGGL(k_occ_probe, grid, block, 0, s, a); // (experiments: its census replaces k_walk's)
#endif
        if (g == 2) BMQ_WALK_LAUNCH(BMQ_WALK_GEOM_SMALLEST); // smallest lists: every overflow path runs all the time (tests)
        else BMQ_WALK_LAUNCH(BMQ_WALK_GEOM_DEFAULT);
#undef BMQ_WALK_LAUNCH
#undef BMQ_WALK_LAUNCH3
    }
    if (e->kernel_events) HIPCHK(e, hipEventRecord(S.ev[2], s));
    // The two repair kernels run only while batches need them (finish_dist turns them on -- and completes the batch that found
    // out -- and off again): topics deeper than FAST_LEVELS (the broker rejects them: Setting.MaxTopicLevels = 16) and rows whose
    // ranges do not come out in ascending id order.  A launch that finds nothing to do still costs its slot in the stream.
    S.ran_slow = e->slow_on;
    S.ran_sort = e->sort_on;
    if (e->slow_on) {
        hipLaunchKernelGGL(k_walk_slow, dim3(256), dim3(64), 0, s, w);
        if (e->kernel_events) HIPCHK(e, hipEventRecord(S.ev[3], s));
    }
    if (adj) hipLaunchKernelGGL(k_fill_adj, dim3(a.n_blocks), dim3(64), 0, s, a, gf); // (inside ms_expand)
    else if (a.rep) hipLaunchKernelGGL(k_fill, dim3(a.n_blocks), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_expand, dim3(a.n_blocks + (EXPAND_PARTS - 1u) * a.heavy_cap), dim3(64), 0, s, a); // (helper waves of the heavy blocks first)
    if (e->sort_on) {
        if (e->kernel_events) HIPCHK(e, hipEventRecord(S.ev[4], s));
        hipLaunchKernelGGL(k_sort_rows, dim3(1024), dim3(256), 0, s, a);
    }
    if (S.total_timed) HIPCHK(e, hipEventRecord(S.ev[5], s));
    if (S.format == BMQ_FMT_RANGES) { // the compact range lists, while the slot's counters still say whether the batch is complete
        const int frc = enqueue_ranges(e, S, a);
        if (frc) return frc;
    }
    // the counters' way to the host: a kernel for small launches (k_publish: stays in the compute queue), the copy engine otherwise
    if (S.d_h_ctr && (e->publish_mode == 2 || (e->publish_mode == 1 && a.n_topics < 4096))) hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, a.ctr, S.d_h_ctr);
    else HIPCHK(e, hipMemcpyAsync(S.h_ctr, a.ctr, sizeof(Counters), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipEventRecord(S.ev_done, s));
    reset_slot(e, S, s); // behind the batch: the counters / allocators are clean again when the next batch arrives
    HIPCHK(e, hipGetLastError());
    S.last = a;
    S.pending = true;
    S.pending_kind = 0;
    return BMQ_OK;
}

// waits; grows internal buffers and re-runs when a kernel asked for it
// BMQ_DEBUG=2: per-wave phase clocks of the last k_walk launch (profiling experiments only)
static void print_wave_debug(bmq_engine* e, bmq_engine::BatchSlot& S) {
    const BatchArgs& a = S.last;
    if (!a.dbg_wave || !a.n_blocks) return;
    std::vector<uint4> h(2 * (size_t)a.n_blocks);
    if (hipMemcpy(h.data(), a.dbg_wave, sizeof(uint4) * 2 * a.n_blocks, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (a.debug_flags & 24u) { // k_walk residency census (8) / the probe kernel's (16): {start lo, start hi, duration, HW_ID | XCC_ID << 16} per wave
        std::vector<std::pair<unsigned long long, int>> ev;
        unsigned long long t0 = ~0ull, t1 = 0, busy = 0;
        std::map<uint32_t, std::vector<std::pair<unsigned long long, int>>> per_simd;
        for (uint32_t i = 0; i < a.n_blocks; i++) {
            const unsigned long long st = ((unsigned long long)h[i].y << 32) | h[i].x, en = st + h[i].z;
            t0 = std::min(t0, st), t1 = std::max(t1, en), busy += h[i].z;
            ev.push_back({st, 1}), ev.push_back({en, -1});
            // HW_ID: wave_id [3:0] simd_id [5:4] pipe [7:6] cu_id [11:8] sh_id [12] se_id [15:13]; + XCC_ID
            const uint32_t simd_key = (h[i].w >> 4) & 0xFFFFFu & ~0xCu; // simd + cu + sh + se + xcc (pipe bits dropped)
            per_simd[simd_key].push_back({st, 1}), per_simd[simd_key].push_back({en, -1});
        }
        if (const char* fn = bmq_env("BMQ_CENSUS_FILE")) { // raw records for offline analysis (tools/census.py)
            if (FILE* f = fopen(fn, "wb")) {
                fwrite(h.data(), sizeof(uint4), a.n_blocks, f);
                fclose(f);
            }
        }
        std::sort(ev.begin(), ev.end());
        int cur = 0, peak = 0;
        for (auto& x : ev) cur += x.second, peak = std::max(peak, cur);
        int simd_peak = 0;
        double simd_peak_sum = 0;
        for (auto& kv : per_simd) {
            std::sort(kv.second.begin(), kv.second.end());
            int c = 0, pk = 0;
            for (auto& x : kv.second) c += x.second, pk = std::max(pk, c);
            simd_peak = std::max(simd_peak, pk), simd_peak_sum += pk;
        }
        fprintf(stderr, "[bmq] k_walk census: %u waves, span %llu ticks, mean wave %.0f ticks, mean resident %.0f (peak %d) waves; %zu distinct SIMD keys, peak waves on one SIMD %d (mean of peaks %.1f)\n",
                a.n_blocks, t1 - t0, (double)busy / a.n_blocks, (double)busy / (double)(t1 - t0), peak, per_simd.size(), simd_peak,
                simd_peak_sum / std::max<size_t>(per_simd.size(), 1));
        return;
    }
    if (a.debug_flags & 4u) { // k_expand: head (row pointers, wave base) | plan + range load + order | prefix | id generation
        if (!BMQ_EXP_CLOCKS) {
            fprintf(stderr, "[bmq] k_expand phase clocks need a build with -DBMQ_EXP_CLOCKS=1 (tools/build_variant.sh)\n");
            return;
        }
        double p[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < a.n_blocks; i++) p[0] += h[i].x, p[1] += h[i].y, p[2] += h[i].z, p[3] += h[i].w;
        fprintf(stderr, "[bmq] k_expand waves=%u clocks/wave: head %.0f load+order %.0f prefix %.0f generate %.0f\n", a.n_blocks,
                p[0] / a.n_blocks, p[1] / a.n_blocks, p[2] / a.n_blocks, p[3] / a.n_blocks);
        { // the waves' lifetimes, and where in the dispatch order the long ones sit (the last tenth of the blocks is the launch's tail)
            std::vector<unsigned long long> tot(a.n_blocks);
            unsigned long long mx_tail = 0;
            for (uint32_t i = 0; i < a.n_blocks; i++) {
                tot[i] = (unsigned long long)h[i].x + h[i].y + h[i].z + h[i].w;
                if (i >= a.n_blocks - a.n_blocks / 10) mx_tail = std::max(mx_tail, tot[i]);
            }
            std::sort(tot.begin(), tot.end());
            fprintf(stderr, "[bmq] k_expand wave lifetimes (clocks): p50 %llu p90 %llu p99 %llu max %llu; longest among the last tenth of the blocks %llu\n",
                    tot[a.n_blocks / 2], tot[(size_t)(a.n_blocks * 0.9)], tot[(size_t)(a.n_blocks * 0.99)], tot.back(), mx_tail);
        }
        return;
    }
    double s1 = 0, s2 = 0, s3 = 0, sr = 0, si = 0;
    std::vector<uint32_t> r(a.n_blocks), c2(a.n_blocks);
    for (uint32_t i = 0; i < a.n_blocks; i++) {
        s1 += h[i].x, s2 += h[i].y, s3 += h[i].z, sr += h[i].w & 255u, si += h[i].w >> 8;
        r[i] = h[i].w & 255u, c2[i] = h[i].y;
    }
    std::sort(r.begin(), r.end());
    std::sort(c2.begin(), c2.end());
    const double n = a.n_blocks;
    {
        double pa = 0, pb = 0, pc = 0;
        for (uint32_t i = 0; i < a.n_blocks; i++) pa += h[a.n_blocks + i].x, pb += h[a.n_blocks + i].y, pc += h[a.n_blocks + i].z;
        fprintf(stderr, "[bmq] k_walk tokenise in detail: stage bytes %.0f | offsets + tenant %.0f | levels + dictionary %.0f\n", pa / n, pb / n, pc / n);
    }
    fprintf(stderr, "[bmq] k_walk waves=%u clocks/wave: tokenise %.0f walk %.0f (p50 %u p99 %u) write %.0f | rounds mean %.1f p50 %u p99 %u max %u | items/wave %.0f\n",
            a.n_blocks, s1 / n, s2 / n, c2[a.n_blocks / 2], c2[(size_t)(a.n_blocks * 0.99)], s3 / n, sr / n, r[a.n_blocks / 2],
            r[(size_t)(a.n_blocks * 0.99)], r.back(), si / n);
}

int finish_dist(bmq_engine* e, bmq_engine::BatchSlot& S, uint64_t* out_total) {
    for (int attempt = 0; attempt < 8; attempt++) {
        HIPCHK(e, hipEventSynchronize(S.ev_done)); // this batch only: a later batch may already be running behind it
        if (S.last.debug_flags & 30u) print_wave_debug(e, S);
        const Counters c = *S.h_ctr;
        const uint32_t grow = c.status & (ST_RERUN | ST_NEED_SORTLIST | ST_NEED_ADJ);
        if (grow) {
            if (grow & ST_NEED_PAIRS) {
                S.pair_cap = S.pair_cap * 2; // slices fill unevenly: double until every sub-allocator fits
                if (S.pair_cap >= 0xFFFFFFFFull) return set_err(e, BMQ_E_RANGE, "matched-range buffer exceeds 2^32 entries");
                HIPCHK(e, S.b_pairs.ensure(sizeof(MatchRange) * S.pair_cap));
            }
            if (grow & ST_NEED_SPILL) {
                S.spill_cap = S.spill_cap * 2;
                if (S.spill_cap >= 0xFFFFFFFFull) return set_err(e, BMQ_E_RANGE, "range spill buffer exceeds 2^32 records");
                HIPCHK(e, S.b_spill.ensure(sizeof(uint4) * S.spill_cap));
            }
            if (grow & ST_NEED_SLOW) {
                S.slow_cap = std::max<uint32_t>(S.slow_cap * 2, c.slow_count);
                HIPCHK(e, S.b_slow_list.ensure(sizeof(uint32_t) * S.slow_cap));
            }
            if (grow & ST_NEED_SCRATCH) {
                S.scratch_cap = std::max<uint64_t>(S.scratch_cap * 2, c.scratch_alloc + c.scratch_alloc / 8);
                HIPCHK(e, S.b_scratch.ensure(sizeof(uint32_t) * S.scratch_cap));
            }
            if (grow & ST_NEED_ADJ) { // the dense batch of an ordered, de-duplicated batch: its topic bytes are known now
                S.adj_cap = (uint64_t)c.adj_bytes + c.adj_bytes / 8 + 4096;
                HIPCHK(e, S.b_c_topics.ensure(S.adj_cap + 64));
            }
            if (grow & ST_NEED_SORTLIST) {
                S.sort_cap = std::max<uint32_t>(S.sort_cap * 2, c.sort_count);
                HIPCHK(e, S.b_sort_list.ensure(sizeof(uint32_t) * S.sort_cap));
            }
            if (c.sort_count) e->sort_on = true, e->sort_idle = 0;
            if (c.slow_count) e->slow_on = true, e->slow_idle = 0;
            if (c.status & ST_WANT_MIXED) e->mixed_on = true, e->mixed_idle = 0;
            BatchArgs a = S.last;
            int rc = launch_dist(e, S, a);
            if (rc) return rc;
            continue;
        }
        if ((c.status & ST_WANT_MIXED) && !S.ran_mixed) { // waves held many tenants each: once more through the instantiation for that
            e->mixed_on = true, e->mixed_idle = 0;
            BatchArgs a = S.last;
            int rc = launch_dist(e, S, a);
            if (rc) return rc;
            continue;
        }
        if (c.slow_count && !S.ran_slow) { // deep topics and k_walk_slow was not in the pipeline: the batch runs again with it
            e->slow_on = true, e->slow_idle = 0;
            BatchArgs a = S.last;
            int rc = launch_dist(e, S, a);
            if (rc) return rc;
            continue;
        }
        if (c.sort_count && !S.ran_sort && !(c.status & (ST_RANGE | ST_NOSPACE))) { // rows to order: only that kernel
            e->sort_on = true, e->sort_idle = 0;
            // the slot's counters were reset behind the batch: k_sort_rows reads the row count from them
            HIPCHK(e, hipMemcpyAsync(S.last.ctr, S.h_ctr, sizeof(Counters), hipMemcpyHostToDevice, e->stream));
            hipLaunchKernelGGL(k_sort_rows, dim3(1024), dim3(256), 0, e->stream, S.last);
            reset_slot(e, S, e->stream);
            HIPCHK(e, hipGetLastError());
            HIPCHK(e, hipStreamSynchronize(e->stream));
        }
        if (S.ran_slow && (c.slow_count ? (e->slow_idle = 0) : ++e->slow_idle) >= REPAIR_IDLE_BATCHES) e->slow_on = false;
        if (S.ran_sort && (c.sort_count ? (e->sort_idle = 0) : ++e->sort_idle) >= REPAIR_IDLE_BATCHES) e->sort_on = false;
        // (the MIXED instantiation cannot tell whether the grouped one would do: it is dropped after a while and comes back if asked for)
        if (S.ran_mixed && ++e->mixed_idle >= 8 * REPAIR_IDLE_BATCHES) e->mixed_on = false;
        S.pending = false;
        bmq_stats& st = e->stats;
        st = bmq_stats{};
        st.n_topics = S.last.n_topics;
        st.n_visit = c.n_visit;
        st.n_match = c.total_ids;
        st.n_ranges = c.n_ranges;
        st.n_slow_topics = c.slow_count;
        st.n_sorted_rows = c.sort_count;
        st.topic_bytes = c.topic_bytes;
        st.n_walked = c.n_walked ? c.n_walked : (S.last.rep ? 0u : S.last.n_topics);
        st.n_split_blocks = S.last.heavy_list ? std::min(c.heavy_count, S.last.heavy_cap) : 0u;
        if (S.last.heavy_list) { // what a block of the coming batches is measured against (launch_dist): a moving mean over the large batches
            const double nb = S.last.n_blocks, mr = c.n_ranges / nb, mi = c.total_ids / nb;
            e->blk_mean_ranges = e->blk_mean_ranges > 0 ? 0.75 * e->blk_mean_ranges + 0.25 * mr : mr;
            e->blk_mean_ids = e->blk_mean_ids > 0 ? 0.75 * e->blk_mean_ids + 0.25 * mi : mi;
            // a list that overflowed held a random part of the heavy blocks (no gain): the bar goes up until the list holds them, and comes back slowly
            constexpr double base = EXPAND_SPLIT_MULT_X16 / 16.0;
            if (c.heavy_count > S.last.heavy_cap) e->split_mult *= 1.25;
            else if (c.heavy_count < S.last.heavy_cap / 4) e->split_mult = std::max(base, e->split_mult / 1.05);
            if (BMQ_EXPERIMENTS && bmq_env("BMQ_DEBUG_HEAVY"))
                fprintf(stderr, "[bmq] k_expand: %u of %u blocks listed as heavy (room for %u; thresholds %u ranges / %u ids; mean block %.0f / %.0f)\n", c.heavy_count,
                        S.last.n_blocks, S.last.heavy_cap, S.last.split_ranges, S.last.split_ids, mr, mi);
        }
        if (S.total_timed) (void)hipEventElapsedTime(&st.ms_total, S.ev[0], S.ev[5]);
        if (S.timed) {
            (void)hipEventElapsedTime(&st.ms_walk, S.ev[1], S.ev[2]);
            (void)hipEventElapsedTime(&st.ms_expand, S.ev[S.ran_slow ? 3 : 2], S.ev[S.ran_sort ? 4 : 5]);
        }
        if (out_total) *out_total = c.total_ids;
        if (c.status & ST_RANGE) return set_err(e, BMQ_E_RANGE, "batch produced >= 2^32 route ids");
        if (c.status & ST_NOSPACE) return set_err(e, BMQ_E_NOSPACE, "output buffer too small");
        return BMQ_OK;
    }
    return set_err(e, BMQ_E_NOMEM, "scratch growth did not converge");
}

int check_dist_ready(bmq_engine* e) {
    if (!e) return BMQ_E_INVAL;
    if (e->device < 0) return set_err(e, BMQ_E_NODEVICE, "engine is host-only: matching requires a gfx950 device");
    if (!e->built || !e->dix || !e->dix->built) return set_err(e, BMQ_E_STATE, "bmq_rebuild has not been called");
    return BMQ_OK;
}

} // namespace

// run `f` on whichever index the engine has (HBM-resident, or host memory for a host-only engine)
template <class F> static auto with_index(bmq_engine* e, F&& f) { return e->dix ? f(*e->dix) : f(*e->hix); }
// (serving generation, generation being built) of whichever executor the engine has
template <class F> static auto with_generations(bmq_engine* e, F&& f) { return e->dix ? f(*e->dix, *e->cmp.next_d) : f(*e->hix, *e->cmp.next_h); }
// (`also`: the executor of a generation being built, when the caller is the thread that drives it -- bmq_compact_poll works on it with
// e->mu released, so nobody else may read its error string: ADVICE r5)
static int index_error(bmq_engine* e, const std::string& msg, bool invalid_input, DevExec* also = nullptr) {
    if (e->dix)
        for (DevExec* x : {&e->dx, &e->dix->x, also})
            if (x && !x->err.empty() && msg == x->err) return set_err(e, BMQ_E_HIP, msg);
    if (msg.find("out of") == 0) return set_err(e, BMQ_E_NOMEM, msg);
    return set_err(e, invalid_input ? BMQ_E_INVAL : BMQ_E_STATE, msg);
}

// A batch handed over with bmq_routes_apply_async: waits for its counters, lets the index finish what the gate stopped, does the
// bookkeeping.  Called (under e->mu) by bmq_routes_apply_wait and in front of everything that reads or changes the route index -- a match
// launch needs the index as the batch left it (regions may have moved), a mutation the ids it handed out.  The batch's error, if any, is
// the caller's: a match launched behind a failed apply fails with it.
static int complete_apply(bmq_engine* e) {
    if (!e->apply_open) return BMQ_OK;
    e->apply_open = false;
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    bool bad_input = false;
    const bool ok = with_index(e, [&](auto& ix) {
        const bool r = ix.apply_end();
        if (!r) {
            e->err = ix.error;
            bad_input = !ix.broken && ix.error.find("malformed") != std::string::npos;
        }
        return r;
    });
    bmq_engine::Compaction& c = e->cmp;
    if (!ok) {
        c.pend_keys.clear(), c.pend_off.clear(), c.pend_op.clear(); // never applied: never replayed
        return index_error(e, e->err, bad_input);
    }
    if (c.active && !c.pend_op.empty()) { // accepted: the generation being built gets it too, before the swap
        if (c.log_keys.size() + c.pend_keys.size() > bmq_engine::Compaction::LOG_CAP_BYTES) {
            c.log_overflow = true; // (the mutation itself is applied all the same: what gives way is the compaction)
            c.log_keys.clear(), c.log_keys.shrink_to_fit(), c.log_off.assign(1, 0u), c.log_op.clear();
        } else if (!c.log_overflow) {
            const uint32_t base = c.log_off.back();
            c.log_keys.insert(c.log_keys.end(), c.pend_keys.begin(), c.pend_keys.end());
            for (size_t i = 1; i < c.pend_off.size(); i++) c.log_off.push_back(base + c.pend_off[i]);
            c.log_op.insert(c.log_op.end(), c.pend_op.begin(), c.pend_op.end());
        }
    }
    c.pend_keys.clear(), c.pend_off.clear(), c.pend_op.clear();
    e->epoch++;
    e->built = true;
    return BMQ_OK;
}

// ====================================================================================================================
// C ABI
// ====================================================================================================================
extern "C" {

const char* bmq_version(void) { return "bifromq_amd 0.2 (gfx950)"; }

int bmq_engine_create(const bmq_config* cfg, bmq_engine** out) {
    if (!out) return BMQ_E_INVAL;
    *out = nullptr;
    bmq_config c{};
    c.device = 0;
    if (cfg) {
        if (cfg->struct_size < 8 || cfg->struct_size > sizeof(bmq_config)) return BMQ_E_INVAL;
        memcpy(&c, cfg, cfg->struct_size);
    }
    // The LDS geometry of k_walk is a compile-time property of its instantiations (bmq_walk_kernel.h): the two caps SELECT one -- 128 in
    // either the smallest lists (192 tokens / 128 items / 128 ranges: tests force the overflow paths with them), 0 the default (512 / 176 /
    // 152: 5.0 KB of LDS per one-wave workgroup, 8 waves per SIMD).  Other values used to run the default silently: refused.
    // BMQ_WALK_GEOM picks an instantiation by number (profiling experiments).
    if (const char* v = bmq_env("BMQ_QCAP")) c.wave_queue_cap = (uint32_t)atoi(v); // profiling experiments
    if (const char* v = bmq_env("BMQ_PCAP")) c.wave_pair_cap = (uint32_t)atoi(v);
    if ((c.wave_queue_cap != 0 && c.wave_queue_cap != 128) || (c.wave_pair_cap != 0 && c.wave_pair_cap != 128)) return BMQ_E_INVAL;
    if (c.region_slack > 64) return BMQ_E_INVAL;
    const bool smallest = c.wave_queue_cap == 128 || c.wave_pair_cap == 128;
    c.wave_queue_cap = smallest ? WALK_QC_SMALLEST : WALK_QC_DEFAULT; // (what bmq_config reports back / BatchArgs carries: the geometry in use)
    c.wave_pair_cap = smallest ? WALK_PC_SMALLEST : WALK_PC_DEFAULT;
    auto e = std::make_unique<bmq_engine>();
    e->cfg = c;
    e->device = c.device;
    e->walk_geom = smallest ? 2 : 0;
    if (c.dedup_min_topics) e->dedup_min = c.dedup_min_topics;
    if (const char* v = bmq_env("BMQ_DEDUP_MIN")) e->dedup_min = (uint32_t)strtoul(v, nullptr, 10); // profiling experiments (4294967295: never)
    e->dedup_sorted = c.dedup_sorted != 0;
    if (const char* v = bmq_env("BMQ_PUBLISH_KERNEL")) e->publish_mode = atoi(v); // profiling experiments
    if (const char* v = bmq_env("BMQ_WALK_GEOM")) e->walk_geom = atoi(v);
    if (const char* v = bmq_env("BMQ_WALK_MIXED")) e->mixed_on = atoi(v) != 0; // profiling experiments
    e->kernel_events = c.kernel_timing != 0;
    if (const char* v = bmq_env("BMQ_KERNEL_EVENTS")) e->kernel_events = atoi(v) != 0; // profiling experiments
    if (c.device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || c.device >= n) return BMQ_E_NODEVICE;
        if (hipSetDevice(c.device) != hipSuccess) return BMQ_E_NODEVICE;
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return BMQ_E_HIP;
        if (hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking) != hipSuccess)
            return BMQ_E_HIP;
        e->dx.upload_stream = e->dxi[0].upload_stream = e->s_in; // an apply batch's ops are uploaded beside the batch the engine stream still runs
        for (auto& sl : e->slots) {
            for (auto& ev : sl.ev)
                if (hipEventCreate(&ev) != hipSuccess) return BMQ_E_HIP;
            if (hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) != hipSuccess)
                return BMQ_E_HIP;
            if (hipHostMalloc((void**)&sl.h_ctr, sizeof(Counters), hipHostMallocDefault) != hipSuccess) return BMQ_E_NOMEM;
            memset(sl.h_ctr, 0, sizeof(Counters));
            if (hipHostGetDevicePointer((void**)&sl.d_h_ctr, sl.h_ctr, 0) != hipSuccess) sl.d_h_ctr = nullptr; // (then: the copy engine, always)
            if (hipEventCreateWithFlags(&sl.ev_fmt, hipEventDisableTiming) != hipSuccess) return BMQ_E_HIP;
            if (hipHostMalloc((void**)&sl.h_fsums, 4 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) return BMQ_E_NOMEM;
            memset(sl.h_fsums, 0, 4 * sizeof(unsigned long long));
        }
        e->dx.device = e->dxi[0].device = e->dxi[1].device = c.device;
        e->dx.stream = e->dxi[0].stream = e->stream;
        e->dix = std::make_unique<DistIndex<DevExec>>(e->dxi[0]);
        if (c.region_slack) e->dix->slack_num = c.region_slack;
        e->drt = std::make_unique<RetainDyn<DevExec>>(e->dx);
    } else {
        e->hix = std::make_unique<DistIndex<HostExec>>(e->hx);
        if (c.region_slack) e->hix->slack_num = c.region_slack;
        e->hrt = std::make_unique<RetainDyn<HostExec>>(e->hx);
    }
    *out = e.release();
    return BMQ_OK;
}

void bmq_engine_destroy(bmq_engine* e) {
    if (!e) return;
    if (e->device >= 0) {
        (void)hipSetDevice(e->device);
        poller_stop_locked(e);
        if (e->pol.stream) (void)hipStreamDestroy(e->pol.stream);
        if (e->pol.ev_index) (void)hipEventDestroy(e->pol.ev_index);
        if (e->pol.desc) (void)hipHostFree(e->pol.desc);
        if (e->pol.done) (void)hipHostFree(e->pol.done);
        if (e->pol.ctl) (void)hipHostFree(e->pol.ctl);
        if (e->pol.blob_in) (void)hipHostFree(e->pol.blob_in);
        if (e->pol.blob_out) (void)hipHostFree(e->pol.blob_out);
        e->pol.scratch.release();
        if (e->stream) (void)hipStreamSynchronize(e->stream);
        if (e->s_in) (void)hipStreamSynchronize(e->s_in);
        if (e->s_out) (void)hipStreamSynchronize(e->s_out);
        for (auto& sl : e->slots) {
            for (auto& ev : sl.ev)
                if (ev) (void)hipEventDestroy(ev);
            if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
            if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
            if (sl.h_ctr) (void)hipHostFree(sl.h_ctr);
            if (sl.ev_fmt) (void)hipEventDestroy(sl.ev_fmt);
            if (sl.h_fsums) (void)hipHostFree(sl.h_fsums);
        }
        if (e->s_in) (void)hipStreamDestroy(e->s_in);
        if (e->s_out) (void)hipStreamDestroy(e->s_out);
        bmq_comm_destroy(e);
        if (e->ev_ex) (void)hipEventDestroy(e->ev_ex);
        if (e->s_ex) (void)hipStreamDestroy(e->s_ex);
        if (e->s_side) (void)hipStreamDestroy(e->s_side);
        if (e->s_build) (void)hipStreamSynchronize(e->s_build);
        e->cmp = bmq_engine::Compaction{}; // (a generation left half-built)
        if (e->s_build) (void)hipStreamDestroy(e->s_build);
        if (e->ev_serving) (void)hipEventDestroy(e->ev_serving);
        if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
        if (e->ev_join) (void)hipEventDestroy(e->ev_join);
        e->dfo.reset();
        e->drt.reset();
        e->dix.reset(); // frees the HBM arrays while the stream still exists
        e->dx.release(e->dx.tmp);
        e->dx.tmp = nullptr;
        if (e->stream) (void)hipStreamDestroy(e->stream);
    }
    delete e;
}

const char* bmq_last_error(const bmq_engine* e) { return e ? e->err.c_str() : "null engine"; }
void* bmq_stream(const bmq_engine* e) { return e ? (void*)e->stream : nullptr; }

int bmq_rebuild(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, uint32_t n_keys) {
    std::unique_lock<std::recursive_mutex> api_lock;
    if (e) api_lock = std::unique_lock<std::recursive_mutex>(e->api);
    if (!e || (n_keys && (!keys || !key_off))) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (int rc_open = complete_apply(e)) return rc_open; // (a batch handed over with bmq_routes_apply_async first)
    if (e->cmp.active) return set_err(e, BMQ_E_STATE, "a compaction is running: bmq_compact_swap or bmq_compact_abort first");
    if (e->cur->pending) return set_err(e, BMQ_E_STATE, "a batch is in flight: call bmq_match_finish first");
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    poller_stop_locked(e); // (the persistent matcher reads the index: it leaves before the index changes)
    static const uint32_t zero_off[1] = {0};
    static const uint8_t no_bytes[16] = {0};
    e->built = false;
    const bool ok = with_index(e, [&](auto& ix) {
        const bool r = ix.rebuild(n_keys ? keys : no_bytes, n_keys ? key_off : zero_off, n_keys);
        if (!r) e->err = ix.error;
        return r;
    });
    if (!ok) return index_error(e, e->err, e->err.find("malformed") != std::string::npos);
    e->epoch++;
    e->built = true;
    return BMQ_OK;
}

int bmq_compact(bmq_engine* e) {
    std::unique_lock<std::recursive_mutex> api_lock;
    if (e) api_lock = std::unique_lock<std::recursive_mutex>(e->api);
    if (!e) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (int rc_open = complete_apply(e)) return rc_open; // (a batch handed over with bmq_routes_apply_async first)
    if (e->cmp.active) return set_err(e, BMQ_E_STATE, "a compaction is running: bmq_compact_swap or bmq_compact_abort first");
    for (auto& sl : e->slots)
        if (sl.pending) return set_err(e, BMQ_E_STATE, "a batch is in flight: finish / wait for it first");
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    poller_stop_locked(e);
    const bool ok = with_index(e, [&](auto& ix) {
        const bool r = ix.compact();
        if (!r) e->err = ix.error;
        return r;
    });
    if (!ok) return index_error(e, e->err, false);
    e->epoch++;
    return BMQ_OK;
}

static int routes_apply_common(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n, bool async) {
    std::unique_lock<std::recursive_mutex> api_lock;
    if (e) api_lock = std::unique_lock<std::recursive_mutex>(e->api);
    if (!e || (n && (!keys || !key_off || !op))) return BMQ_E_INVAL;
    // The builder kernels run on the engine stream, in order with the match batches: a mutation costs the matcher threads the
    // duration of its kernels (well under a millisecond for 100 k ops), not a host-side rebuild.
    std::lock_guard<std::mutex> g(e->mu);
    if (int rc = complete_apply(e)) return rc; // the batch before this one
    if (n == 0) return BMQ_OK;
    // a batch handed over with bmq_match_submit is simply in front of the builder kernels on the engine stream; only the
    // caller-driven *_dev protocol (results read by the caller between launch and finish) excludes a mutation in between
    if (e->cur->pending) return set_err(e, BMQ_E_STATE, "a batch is in flight: call bmq_match_finish first");
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    poller_stop_locked(e); // (the persistent matcher reads the index: it leaves before the builder kernels are enqueued)
    bool bad_input = false;
    const bool ok = with_index(e, [&](auto& ix) {
        const bool r = ix.apply_begin(keys, key_off, op, n);
        if (!r) {
            e->err = ix.error;
            bad_input = !ix.broken && ix.error.find("malformed") != std::string::npos;
        }
        return r;
    });
    if (!ok) return index_error(e, e->err, bad_input);
    if (e->cmp.active && !e->cmp.log_overflow) { // a generation is being built beside this one: remembered until the batch's outcome is known (complete_apply)
        bmq_engine::Compaction& c = e->cmp;
        c.pend_keys.assign(keys, keys + key_off[n]);
        c.pend_off.assign(key_off, key_off + n + 1);
        c.pend_op.assign(op, op + n);
    }
    e->apply_open = true;
    return async ? BMQ_OK : complete_apply(e);
}
int bmq_routes_apply(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n) {
    return routes_apply_common(e, keys, key_off, op, n, false);
}
int bmq_routes_apply_async(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n) {
    return routes_apply_common(e, keys, key_off, op, n, true);
}
int bmq_routes_apply_wait(bmq_engine* e) {
    if (!e) return BMQ_E_INVAL;
    std::lock_guard<std::recursive_mutex> api_lock(e->api);
    std::lock_guard<std::mutex> g(e->mu);
    return complete_apply(e);
}

// ---- compaction without a stall: the next generation is built beside the serving one -------------------------------------------------
// (TopicLevelTrie contracts as it goes, UTIL/index/TopicLevelTrie.java:257-384; here deleted routes leave garbage -- abandoned id lists,
// dead trie nodes, dictionary tokens nobody uses, key bytes -- until a generation change.)  The key bytes never leave HBM: a chunk of the
// serving generation's live keys is gathered into its staging buffer and handed to the next generation's builder as it is (puts through
// the apply path: no order required, so no sort); what is mutated meanwhile is logged and replayed before the swap.
static int replay_log(bmq_engine* e, size_t from) {
    bmq_engine::Compaction& c = e->cmp;
    const size_t n = c.log_op.size();
    std::vector<uint32_t> off;
    for (size_t lo = from; lo < n;) {
        const size_t hi = std::min(n, lo + 65536);
        off.resize(hi - lo + 1);
        for (size_t i = lo; i <= hi; i++) off[i - lo] = c.log_off[i] - c.log_off[lo];
        std::string msg;
        if (!with_generations(e, [&](auto&, auto& next) {
                const bool r = next.apply(c.log_keys.data() + c.log_off[lo], off.data(), c.log_op.data() + lo, (uint32_t)(hi - lo));
                if (!r) msg = next.error;
                return r;
            }))
            return index_error(e, msg, false, c.bx);
        lo = hi;
    }
    return BMQ_OK;
}
int bmq_compact_begin(bmq_engine* e) {
    if (!e) return BMQ_E_INVAL;
    std::lock_guard<std::recursive_mutex> api_lock(e->api);
    std::lock_guard<std::mutex> gc(e->cmp_mu);
    std::lock_guard<std::mutex> g(e->mu);
    if (int rc = complete_apply(e)) return rc;
    if (e->cmp.active) return set_err(e, BMQ_E_STATE, "a compaction is running: bmq_compact_swap or bmq_compact_abort first");
    if (!e->built) return set_err(e, BMQ_E_STATE, "no index");
    e->cmp = bmq_engine::Compaction{};
    if (e->dix) {
        HIPCHK(e, hipSetDevice(e->device));
        if (!e->s_build) {
            int least = 0, greatest = 0;
            HIPCHK(e, hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(e, hipStreamCreateWithPriority(&e->s_build, hipStreamNonBlocking, least)); // a match batch's waves go first
            HIPCHK(e, hipEventCreateWithFlags(&e->ev_serving, hipEventDisableTiming));
        }
        DevExec* bx = &e->dix->x == &e->dxi[0] ? &e->dxi[1] : &e->dxi[0];
        bx->stream = e->s_build;
        bx->upload_stream = nullptr;
        e->cmp.bx = bx;
        e->cmp.next_d = std::make_unique<DistIndex<DevExec>>(*bx);
        e->cmp.next_d->slack_num = e->dix->slack_num;
        // the sizes below are read through the serving generation's executor: behind what the engine stream holds
    } else {
        e->cmp.next_h = std::make_unique<DistIndex<HostExec>>(e->hx);
        e->cmp.next_h->slack_num = e->hix->slack_num;
    }
    std::string msg;
    if (!with_generations(e, [&](auto& cur, auto& next) {
            e->cmp.n_ids = cur.id_bound();
            const bool r = next.reserve_like(cur) && next.reserve_import(65536, 65536ull * 192); // regions, pools and tables at their final size, the chunk buffers too: the carry-over allocates nothing (chunks of up to 65536 ids)
            if (!r) msg = next.error;
            cur.defer_release = r; // what the serving generation outgrows meanwhile stays readable for the builder
            return r;
        })) {
        e->cmp = bmq_engine::Compaction{};
        return index_error(e, msg, false);
    }
    e->cmp.active = true;
    return BMQ_OK;
}
int bmq_compact_poll(bmq_engine* e, uint32_t max_ids, uint32_t* out_done_permille) {
    if (!e) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> gc(e->cmp_mu);
    bmq_engine::Compaction& c = e->cmp;
    std::unique_lock<std::mutex> g(e->mu);
    if (!c.active) return set_err(e, BMQ_E_STATE, "no compaction is running");
    if (c.log_overflow) return set_err(e, BMQ_E_NOSPACE, "the mutation log of this compaction outgrew its cap (1 GiB of route keys): bmq_compact_abort, then begin again");
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    if (c.cursor < c.n_ids) {
        const uint32_t hi = (uint32_t)std::min<uint64_t>(c.n_ids, (uint64_t)c.cursor + std::max(max_ids, 1u));
        std::string msg;
        // (1) under the engine lock, nothing waited for: the build stream is put behind what the serving generation was told so far and
        // copies its key references of the chunk.  A mutation that lands later is in the log (routes_apply_common), whichever side of the
        // copy its effect on a reference falls.
        if (e->dix) {
            HIPCHK(e, hipEventRecord(e->ev_serving, e->stream));
            HIPCHK(e, hipStreamWaitEvent(e->s_build, e->ev_serving, 0));
        }
        bool ok = with_generations(e, [&](auto& cur, auto& next) {
            const bool r = next.import_snapshot(cur, c.cursor, hi);
            if (!r) msg = next.error;
            return r;
        });
        // (2) the rest touches the generation being built only, on its own stream: matching and mutations go on meanwhile
        if (ok && e->dix) g.unlock();
        uint32_t n_live = 0;
        if (ok) {
            auto finish = [&](auto& next) {
                const bool r = next.import_apply(n_live);
                if (!r) msg = next.error;
                return r;
            };
            ok = c.next_d ? finish(*c.next_d) : finish(*c.next_h);
        }
        if (!g.owns_lock()) g.lock();
        if (!ok) return index_error(e, msg, false, c.bx);
        c.carried += n_live;
        c.cursor = hi;
    }
    if (out_done_permille) *out_done_permille = c.n_ids ? (uint32_t)((uint64_t)c.cursor * 1000 / c.n_ids) : 1000u;
    return BMQ_OK;
}
int bmq_compact_swap(bmq_engine* e, uint64_t* out_carried, uint64_t* out_replayed) {
    if (!e) return BMQ_E_INVAL;
    std::lock_guard<std::recursive_mutex> api_lock(e->api);
    std::lock_guard<std::mutex> gc(e->cmp_mu);
    std::lock_guard<std::mutex> g(e->mu);
    bmq_engine::Compaction& c = e->cmp;
    if (!c.active) return set_err(e, BMQ_E_STATE, "no compaction is running");
    if (c.cursor < c.n_ids) return set_err(e, BMQ_E_STATE, "the next generation is not complete: bmq_compact_poll until it reports 1000");
    if (int rc = complete_apply(e)) return rc;
    if (c.log_overflow) return set_err(e, BMQ_E_NOSPACE, "the mutation log of this compaction outgrew its cap (1 GiB of route keys): bmq_compact_abort, then begin again");
    for (auto& sl : e->slots)
        if (sl.pending) return set_err(e, BMQ_E_STATE, "a batch is in flight: finish / wait for it first");
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    poller_stop_locked(e); // (the generations change places below)
    if (int rc = replay_log(e, 0)) return rc; // what the serving generation was told since bmq_compact_begin, in order
    if (e->dix) {
        HIPCHK(e, hipStreamSynchronize(e->s_build));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        c.bx->stream = e->stream; // the executor under the new serving generation works on the engine stream from here on
        c.bx->upload_stream = e->s_in;
    }
    if (out_carried) *out_carried = c.carried;
    if (out_replayed) *out_replayed = c.log_op.size();
    with_generations(e, [&](auto& cur, auto& next) {
        next.generation = cur.generation + 1; // route ids of the two generations are unrelated
        return true;
    });
    e->dfo.reset(); // (the fan-out grouping state belongs to the index it was built over)
    if (e->dix) e->dix.swap(c.next_d);
    else e->hix.swap(c.next_h);
    e->cmp = bmq_engine::Compaction{}; // frees the old generation (and the blocks it outgrew meanwhile)
    e->epoch++;
    return BMQ_OK;
}
int bmq_compact_abort(bmq_engine* e) {
    if (!e) return BMQ_E_INVAL;
    std::lock_guard<std::recursive_mutex> api_lock(e->api);
    std::lock_guard<std::mutex> gc(e->cmp_mu);
    std::lock_guard<std::mutex> g(e->mu);
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    if (e->cmp.active) {
        if (e->dix) {
            HIPCHK(e, hipStreamSynchronize(e->s_build));
            HIPCHK(e, hipStreamSynchronize(e->stream));
        }
        with_index(e, [&](auto& cur) {
            cur.defer_release = false;
            cur.release_deferred();
            return true;
        });
    }
    e->cmp = bmq_engine::Compaction{};
    return BMQ_OK;
}

int bmq_index_info_get(const bmq_engine* ce, bmq_index_info* out) {
    bmq_engine* e = const_cast<bmq_engine*>(ce);
    if (!e || !out) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (int rc_open = complete_apply(e)) return rc_open; // (a batch handed over with bmq_routes_apply_async first)
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    DistIndexStats st;
    uint64_t generation = 0;
    const bool ok = with_index(e, [&](auto& ix) {
        generation = ix.generation;
        const bool r = ix.stats(st);
        if (!r) e->err = ix.error;
        return r;
    });
    if (!ok) return index_error(e, e->err, false);
    memset(out, 0, sizeof(*out));
    out->n_routes = st.n_routes;
    out->n_tenants = st.n_tenants;
    out->n_nodes = st.n_nodes;
    out->n_tokens = st.n_tokens;
    out->trie_slots = st.trie_slots;
    out->dict_slots = st.dict_slots;
    out->device_bytes = e->dix ? st.bytes : 0;
    out->epoch = e->epoch;
    out->generation = generation;
    out->next_route_id = st.next_id;
    out->garbage_bytes = st.trie_garbage_slots * sizeof(TrieSlot) + st.id_list_garbage * 4;
    return BMQ_OK;
}

int bmq_route_key(const bmq_engine* ce, uint32_t route_id, uint8_t* out, uint32_t cap, uint32_t* out_len) {
    bmq_engine* e = const_cast<bmq_engine*>(ce);
    if (!e || !out_len) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (int rc_open = complete_apply(e)) return rc_open; // (a batch handed over with bmq_routes_apply_async first)
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    std::string k;
    bool hard = false;
    const bool ok = with_index(e, [&](auto& ix) {
        const bool r = ix.route_key(route_id, k);
        if (!r && !ix.error.empty()) {
            hard = true;
            e->err = ix.error;
        }
        return r;
    });
    if (!ok) return hard ? index_error(e, e->err, false) : BMQ_E_INVAL; // no such route (never existed, or deleted)
    *out_len = (uint32_t)k.size();
    if (k.size() > cap) return BMQ_E_NOSPACE;
    if (out && !k.empty()) memcpy(out, k.data(), k.size());
    return BMQ_OK;
}

int bmq_route_keys(const bmq_engine* ce, const uint32_t* route_ids, uint32_t n, uint8_t* out, uint64_t cap, uint64_t* out_off) {
    bmq_engine* e = const_cast<bmq_engine*>(ce);
    if (!e || !out_off || (n && !route_ids)) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (int rc_open = complete_apply(e)) return rc_open; // (a batch handed over with bmq_routes_apply_async first)
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off;
    const bool ok = with_index(e, [&](auto& ix) {
        const bool r = ix.route_keys(route_ids, n, bytes, off);
        if (!r) e->err = ix.error;
        return r;
    });
    if (!ok) return index_error(e, e->err, false);
    memcpy(out_off, off.data(), sizeof(uint64_t) * ((size_t)n + 1));
    if (off[n] > cap || (off[n] && !out)) return BMQ_E_NOSPACE;
    if (off[n]) memcpy(out, bytes.data(), off[n]);
    return BMQ_OK;
}

int bmq_index_find(const bmq_engine* ce, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filter,
                   uint32_t filter_len, uint32_t* out_ids, uint32_t cap, uint32_t* out_n) {
    bmq_engine* e = const_cast<bmq_engine*>(ce);
    if (!e || !out_n) return BMQ_E_INVAL;
    *out_n = 0;
    std::lock_guard<std::mutex> g(e->mu);
    if (int rc_open = complete_apply(e)) return rc_open; // (a batch handed over with bmq_routes_apply_async first)
    if (!e->built) return BMQ_E_STATE;
    if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
    std::vector<uint32_t> ids;
    const bool ok = with_index(e, [&](auto& ix) {
        const bool r = ix.find(std::string_view((const char*)tenant, tenant_len), std::string_view((const char*)filter, filter_len), ids);
        if (!r) e->err = ix.error;
        return r;
    });
    if (!ok) return index_error(e, e->err, false);
    *out_n = (uint32_t)ids.size();
    for (uint32_t i = 0; i < ids.size() && i < cap; i++) out_ids[i] = ids[i];
    return BMQ_OK;
}

int bmq_match_batch_dev(bmq_engine* e, const uint8_t* d_tenants, const uint32_t* d_tenant_off, uint32_t n_tenants,
                        const uint32_t* d_topic_tenant, const uint8_t* d_topics, const uint32_t* d_topic_off,
                        uint32_t n_topics, uint32_t* d_out_row_ptr, uint32_t* d_out_route_ids, uint64_t out_capacity,
                        uint64_t* d_out_total) {
    int rc = check_dist_ready(e);
    if (rc) return rc;
    if (n_topics == 0 || !d_out_row_ptr || !d_topic_off || !d_topics || !d_topic_tenant || !d_out_total)
        return set_err(e, BMQ_E_INVAL, "null pointer or empty batch");
    if ((uintptr_t)d_topics & 15) return set_err(e, BMQ_E_INVAL, "the topic byte buffer must be 16-byte aligned");
    // The launch .. finish window belongs to ONE caller: the api lock is taken here and given back by bmq_match_finish (same
    // thread), so a host-buffer call or a batcher launch of another thread cannot consume or overwrite this batch.
    if (!e->api.try_lock()) return set_err(e, BMQ_E_STATE, "the engine is busy with another thread's call");
    struct Unlock {
        bmq_engine* e;
        bool keep = false;
        ~Unlock() {
            if (!keep) e->api.unlock();
        }
    } api_guard{e};
    std::lock_guard<std::mutex> g(e->mu);
    if (e->cur->pending) return set_err(e, BMQ_E_STATE, "a batch is in flight: call bmq_match_finish first");
    HIPCHK(e, hipSetDevice(e->device));
    if ((rc = ensure_batch_scratch(e, *e->cur, n_tenants, n_topics))) return rc;
    BatchArgs a{};
    a.tenants = d_tenants;
    a.tenant_off = d_tenant_off;
    a.n_tenants = n_tenants;
    a.topic_tenant = d_topic_tenant;
    a.topics = d_topics;
    a.topic_off = d_topic_off;
    a.n_topics = n_topics;
    a.out_row_ptr = d_out_row_ptr;
    a.out_ids = d_out_route_ids;
    a.out_capacity = d_out_route_ids ? out_capacity : 0;
    a.out_total = (unsigned long long*)d_out_total;
    rc = launch_dist(e, *e->cur, a);
    if (rc == BMQ_OK) {
        api_guard.keep = true;
        e->cur->api_held = true;
    }
    return rc;
}

int bmq_match_finish(bmq_engine* e, uint64_t* out_total) {
    if (!e) return BMQ_E_INVAL;
    if (e->device < 0) return set_err(e, BMQ_E_NODEVICE, "engine is host-only");
    int rc;
    bool give_back = false;
    {
        std::lock_guard<std::mutex> g(e->mu);
        if (!e->cur->pending) return set_err(e, BMQ_E_STATE, "no batch in flight");
        HIPCHK(e, hipSetDevice(e->device));
        rc = e->cur->pending_kind == 0 ? finish_dist(e, *e->cur, out_total) : retain_finish(e, out_total);
        if (rc != BMQ_OK) e->cur->pending = false; // a failed batch is over too: the next launch starts clean
        give_back = e->cur->api_held;
        e->cur->api_held = false;
    }
    if (give_back) e->api.unlock();
    return rc;
}

int bmq_sync(bmq_engine* e) {
    if (!e) return BMQ_E_INVAL;
    if (e->device < 0) return BMQ_E_NODEVICE;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return BMQ_OK;
}

int bmq_set_kernel_timing(bmq_engine* e, int on) {
    if (!e) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> g(e->mu);
    e->kernel_events = on != 0;
    return BMQ_OK;
}

int bmq_stats_get(const bmq_engine* e, bmq_stats* out) {
    if (!e || !out) return BMQ_E_INVAL;
    *out = e->stats;
    return BMQ_OK;
}

int bmq_match_batch(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                    const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                    uint32_t* out_row_ptr, uint32_t* out_route_ids, uint64_t out_capacity, uint64_t* out_needed) {
    std::unique_lock<std::recursive_mutex> api_lock;
    if (e) api_lock = std::unique_lock<std::recursive_mutex>(e->api);
    int rc = check_dist_ready(e);
    if (rc) return rc;
    if (!out_row_ptr || !out_needed) return set_err(e, BMQ_E_INVAL, "null output pointer");
    *out_needed = 0;
    if (n_topics == 0) {
        out_row_ptr[0] = 0;
        return BMQ_OK;
    }
    if (!topics || !topic_off || !topic_tenant || (n_tenants && (!tenants || !tenant_off)))
        return set_err(e, BMQ_E_INVAL, "null input pointer");
    uint64_t dev_cap;
    {
        std::lock_guard<std::mutex> g(e->mu);
        HIPCHK(e, hipSetDevice(e->device));
        const size_t tb = n_tenants ? tenant_off[n_tenants] : 0, pb = topic_off[n_topics];
        HIPCHK(e, e->cur->s_tenants.ensure(tb + 16));
        HIPCHK(e, e->cur->s_topics.ensure(pb + 16));
        if ((rc = upload(e, e->cur->s_tenant_off, tenant_off, sizeof(uint32_t) * (n_tenants ? n_tenants + 1 : 0)))) return rc;
        if (tb) HIPCHK(e, hipMemcpyAsync(e->cur->s_tenants.p, tenants, tb, hipMemcpyHostToDevice, e->stream));
        if ((rc = upload(e, e->cur->s_topic_tenant, topic_tenant, sizeof(uint32_t) * n_topics))) return rc;
        if (pb) HIPCHK(e, hipMemcpyAsync(e->cur->s_topics.p, topics, pb, hipMemcpyHostToDevice, e->stream));
        if ((rc = upload(e, e->cur->s_topic_off, topic_off, sizeof(uint32_t) * (n_topics + 1)))) return rc;
        HIPCHK(e, e->cur->s_row_ptr.ensure(sizeof(uint32_t) * (n_topics + 1)));
        HIPCHK(e, e->cur->b_total.ensure(sizeof(unsigned long long)));
        dev_cap = std::max<uint64_t>(e->cur->s_ids.cap / 4, std::max<uint64_t>((uint64_t)n_topics * 4, 1024));
        HIPCHK(e, e->cur->s_ids.ensure(dev_cap * 4));
    }
    for (int attempt = 0; attempt < 3; attempt++) {
        rc = bmq_match_batch_dev(e, e->cur->s_tenants.as<uint8_t>(), e->cur->s_tenant_off.as<uint32_t>(), n_tenants,
                                 e->cur->s_topic_tenant.as<uint32_t>(), e->cur->s_topics.as<uint8_t>(), e->cur->s_topic_off.as<uint32_t>(),
                                 n_topics, e->cur->s_row_ptr.as<uint32_t>(), e->cur->s_ids.as<uint32_t>(), dev_cap,
                                 e->cur->b_total.as<uint64_t>());
        if (rc) return rc;
        uint64_t total = 0;
        rc = bmq_match_finish(e, &total);
        *out_needed = total;
        if (rc == BMQ_E_NOSPACE) {
            if (total > out_capacity || !out_route_ids) return rc; // the caller's buffer is the problem
            std::lock_guard<std::mutex> g(e->mu);
            dev_cap = total;
            HIPCHK(e, e->cur->s_ids.ensure(dev_cap * 4));
            continue;
        }
        if (rc) return rc;
        std::lock_guard<std::mutex> g(e->mu);
        HIPCHK(e, hipMemcpy(out_row_ptr, e->cur->s_row_ptr.p, sizeof(uint32_t) * (n_topics + 1), hipMemcpyDeviceToHost));
        if (total > out_capacity || (total && !out_route_ids)) return set_err(e, BMQ_E_NOSPACE, "output buffer too small");
        if (total) HIPCHK(e, hipMemcpy(out_route_ids, e->cur->s_ids.p, sizeof(uint32_t) * total, hipMemcpyDeviceToHost));
        return BMQ_OK;
    }
    return set_err(e, BMQ_E_NOMEM, "device output buffer growth did not converge");
}

// ---- asynchronous host-buffer match: two batches in flight ---------------------------------------------------------------
void* bmq_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void bmq_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int bmq_match_submit(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                     const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, int* out_ticket) {
    return bmq_match_submit_fmt(e, tenants, tenant_off, n_tenants, topic_tenant, topics, topic_off, n_topics, BMQ_FMT_IDS, out_ticket);
}

int bmq_match_submit_fmt(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                         const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, int format, int* out_ticket) {
    int rc = check_dist_ready(e);
    if (rc) return rc;
    if (format < BMQ_FMT_IDS || format > BMQ_FMT_GROUPED) return set_err(e, BMQ_E_INVAL, "unknown result format");
    if (!out_ticket || n_topics == 0 || !topics || !topic_off || !topic_tenant || (n_tenants && (!tenants || !tenant_off)))
        return set_err(e, BMQ_E_INVAL, "null pointer or empty batch");
    std::lock_guard<std::mutex> g(e->mu);
    HIPCHK(e, hipSetDevice(e->device));
    int k = -1;
    for (int i = 0; i < BMQ_MAX_TICKETS; i++)
        if (!e->slots[1 + i].pending && !e->slots[1 + i].submitted) {
            k = i;
            break;
        }
    if (k < 0) return set_err(e, BMQ_E_STATE, "every ticket is in flight: call bmq_match_wait first");
    bmq_engine::BatchSlot& S = e->slots[1 + k]; // ticket k: never the slot the blocking entry points stage into
    const size_t tb = n_tenants ? tenant_off[n_tenants] : 0, pb = topic_off[n_topics];
    HIPCHK(e, S.s_tenants.ensure(tb + 16));
    HIPCHK(e, S.s_topics.ensure(pb + 16));
    HIPCHK(e, S.s_tenant_off.ensure(sizeof(uint32_t) * ((size_t)n_tenants + 1)));
    HIPCHK(e, S.s_topic_tenant.ensure(sizeof(uint32_t) * (size_t)n_topics));
    HIPCHK(e, S.s_topic_off.ensure(sizeof(uint32_t) * ((size_t)n_topics + 1)));
    HIPCHK(e, S.s_row_ptr.ensure(sizeof(uint32_t) * ((size_t)n_topics + 1)));
    HIPCHK(e, S.b_total.ensure(sizeof(unsigned long long)));
    const bool with_ids = format == BMQ_FMT_IDS || format == BMQ_FMT_GROUPED;
    if (with_ids) {
        S.dev_cap = std::max<uint64_t>(S.s_ids.cap / 4, std::max<uint64_t>((uint64_t)n_topics * 24, 1024));
        HIPCHK(e, S.s_ids.ensure(S.dev_cap * 4));
    }
    if ((rc = ensure_batch_scratch(e, S, n_tenants, n_topics))) return rc;
    // upload on the copy-in stream: it overlaps the kernels of the batch submitted before (pinned sources: bmq_host_alloc)
    if (n_tenants) {
        HIPCHK(e, hipMemcpyAsync(S.s_tenant_off.p, tenant_off, sizeof(uint32_t) * ((size_t)n_tenants + 1), hipMemcpyHostToDevice, e->s_in));
        if (tb) HIPCHK(e, hipMemcpyAsync(S.s_tenants.p, tenants, tb, hipMemcpyHostToDevice, e->s_in));
    }
    HIPCHK(e, hipMemcpyAsync(S.s_topic_tenant.p, topic_tenant, sizeof(uint32_t) * (size_t)n_topics, hipMemcpyHostToDevice, e->s_in));
    if (pb) HIPCHK(e, hipMemcpyAsync(S.s_topics.p, topics, pb, hipMemcpyHostToDevice, e->s_in));
    HIPCHK(e, hipMemcpyAsync(S.s_topic_off.p, topic_off, sizeof(uint32_t) * ((size_t)n_topics + 1), hipMemcpyHostToDevice, e->s_in));
    HIPCHK(e, hipEventRecord(S.ev_in, e->s_in));
    HIPCHK(e, hipStreamWaitEvent(e->stream, S.ev_in, 0));
    BatchArgs a{};
    a.tenants = S.s_tenants.as<uint8_t>();
    a.tenant_off = S.s_tenant_off.as<uint32_t>();
    a.n_tenants = n_tenants;
    a.topic_tenant = S.s_topic_tenant.as<uint32_t>();
    a.topics = S.s_topics.as<uint8_t>();
    a.topic_off = S.s_topic_off.as<uint32_t>();
    a.n_topics = n_topics;
    a.out_row_ptr = S.s_row_ptr.as<uint32_t>();
    // COUNTS / RANGES: no id is written -- k_expand still lays down the row pointers (= the fan-out of every topic) and the totals,
    // sees that nothing fits a buffer of 0 ids, and leaves (ST_NOSPACE, which these formats expect)
    a.out_ids = with_ids ? S.s_ids.as<uint32_t>() : nullptr;
    a.out_capacity = with_ids ? S.dev_cap : 0;
    a.out_total = S.b_total.as<unsigned long long>();
    S.format = format;
    if ((rc = launch_dist(e, S, a))) {
        S.format = BMQ_FMT_IDS;
        return rc;
    }
    S.submitted = true;
    S.n_rows = n_topics;
    *out_ticket = k;
    return BMQ_OK;
}

int bmq_match_wait(bmq_engine* e, int ticket, uint32_t* out_row_ptr, uint32_t* out_route_ids, uint64_t out_capacity, uint64_t* out_needed) {
    if (!e || ticket < 0 || ticket >= BMQ_MAX_TICKETS || !out_row_ptr || !out_needed) return BMQ_E_INVAL;
    if (e->device < 0) return set_err(e, BMQ_E_NODEVICE, "engine is host-only");
    bmq_engine::BatchSlot& S = e->slots[1 + ticket];
    uint64_t total = 0;
    {
        std::lock_guard<std::mutex> g(e->mu);
        if (!S.submitted) return set_err(e, BMQ_E_STATE, "no such ticket in flight");
        if (S.format != BMQ_FMT_IDS || S.devptr) return set_err(e, BMQ_E_STATE, "the ticket was submitted with another result format");
        HIPCHK(e, hipSetDevice(e->device));
    }
    (void)hipEventSynchronize(S.ev_done); // outside the lock: other threads may submit / apply meanwhile
    std::unique_lock<std::mutex> g(e->mu);
    int rc = BMQ_OK;
    {
        for (int attempt = 0; attempt < 3; attempt++) {
            rc = finish_dist(e, S, &total); // grows internal scratch and re-runs if a kernel asked for it
            if (rc != BMQ_E_NOSPACE || total <= S.dev_cap) break;
            S.dev_cap = total; // the slot's own id buffer was too small: it knows the size now
            if (S.s_ids.ensure(S.dev_cap * 4) != hipSuccess) {
                rc = set_err(e, BMQ_E_NOMEM, "out of device memory (result buffer)");
                break;
            }
            BatchArgs a = S.last;
            a.out_ids = S.s_ids.as<uint32_t>();
            a.out_capacity = S.dev_cap;
            if ((rc = launch_dist(e, S, a))) break;
        }
    }
    *out_needed = total;
    const bool fits = total <= out_capacity && (total == 0 || out_route_ids);
    hipError_t he = hipSuccess;
    if (rc == BMQ_OK) {
        // download on the copy-out stream: it overlaps the kernels of the batch submitted after this one.  The slot stays taken
        // until the copy has finished; the engine lock is not held meanwhile.
        he = hipMemcpyAsync(out_row_ptr, S.s_row_ptr.p, sizeof(uint32_t) * ((size_t)S.n_rows + 1), hipMemcpyDeviceToHost, e->s_out);
        if (he == hipSuccess && fits && total)
            he = hipMemcpyAsync(out_route_ids, S.s_ids.p, sizeof(uint32_t) * total, hipMemcpyDeviceToHost, e->s_out);
        g.unlock();
        if (he == hipSuccess) he = hipStreamSynchronize(e->s_out);
        g.lock();
    }
    S.submitted = false;
    S.pending = false;
    if (rc) return rc;
    if (he != hipSuccess) return set_err(e, BMQ_E_HIP, std::string("result download: ") + hipGetErrorString(he));
    return fits ? BMQ_OK : set_err(e, BMQ_E_NOSPACE, "output buffer too small");
}

// Tickets over device-accessible buffers (include/bmq.h): nothing is staged and nothing copied -- the kernels read the caller's
// inputs and write the caller's outputs in place (HBM, or page-locked host memory over PCIe).
int bmq_match_submit_dev(bmq_engine* e, const uint8_t* d_tenants, const uint32_t* d_tenant_off, uint32_t n_tenants, const uint32_t* d_topic_tenant,
                         const uint8_t* d_topics, const uint32_t* d_topic_off, uint32_t n_topics, uint32_t* d_out_row_ptr, uint32_t* d_out_route_ids,
                         uint64_t out_capacity, uint64_t* d_out_total, int* out_ticket) {
    int rc = check_dist_ready(e);
    if (rc) return rc;
    if (!out_ticket || n_topics == 0 || !d_out_row_ptr || !d_topic_off || !d_topics || !d_topic_tenant || !d_out_total || (n_tenants && (!d_tenants || !d_tenant_off)))
        return set_err(e, BMQ_E_INVAL, "null pointer or empty batch");
    if ((uintptr_t)d_topics & 15) return set_err(e, BMQ_E_INVAL, "the topic byte buffer must be 16-byte aligned");
    std::lock_guard<std::mutex> g(e->mu);
    HIPCHK(e, hipSetDevice(e->device));
    int k = -1;
    for (int i = 0; i < BMQ_MAX_TICKETS; i++)
        if (!e->slots[1 + i].pending && !e->slots[1 + i].submitted) {
            k = i;
            break;
        }
    if (k < 0) return set_err(e, BMQ_E_STATE, "every ticket is in flight: call bmq_match_wait first");
    bmq_engine::BatchSlot& S = e->slots[1 + k];
    if ((rc = ensure_batch_scratch(e, S, n_tenants, n_topics))) return rc;
    BatchArgs a{};
    a.tenants = d_tenants;
    a.tenant_off = d_tenant_off;
    a.n_tenants = n_tenants;
    a.topic_tenant = d_topic_tenant;
    a.topics = d_topics;
    a.topic_off = d_topic_off;
    a.n_topics = n_topics;
    a.out_row_ptr = d_out_row_ptr;
    a.out_ids = d_out_route_ids;
    a.out_capacity = d_out_route_ids ? out_capacity : 0;
    a.out_total = (unsigned long long*)d_out_total;
    S.format = BMQ_FMT_IDS;
    if ((rc = launch_dist(e, S, a))) return rc;
    S.submitted = true;
    S.devptr = true;
    S.n_rows = n_topics;
    *out_ticket = k;
    return BMQ_OK;
}

int bmq_match_wait_dev(bmq_engine* e, int ticket, uint64_t* out_total) {
    if (!e || ticket < 0 || ticket >= BMQ_MAX_TICKETS) return BMQ_E_INVAL;
    if (e->device < 0) return set_err(e, BMQ_E_NODEVICE, "engine is host-only");
    bmq_engine::BatchSlot& S = e->slots[1 + ticket];
    {
        std::lock_guard<std::mutex> g(e->mu);
        if (!S.submitted || !S.devptr) return set_err(e, BMQ_E_STATE, "no such device-buffer ticket in flight");
        HIPCHK(e, hipSetDevice(e->device));
    }
    (void)hipEventSynchronize(S.ev_done); // outside the lock: other threads may submit / apply meanwhile
    std::lock_guard<std::mutex> g(e->mu);
    uint64_t total = 0;
    const int rc = finish_dist(e, S, &total); // grows internal scratch and re-runs if a kernel asked for it
    if (out_total) *out_total = total;
    S.submitted = false;
    S.pending = false;
    S.devptr = false;
    return rc;
}

// ---- MatchedRoutes caps (DW/cache/MatchedRoutes.java:87-141) over rows of route ids ----------------------------------------
// One row = what matchAll found for one topic.  Persistent (subBrokerId == 1) normal routes and group routes are admitted
// first-come in KV KEY order up to their caps; every rejected route is a throttle event.  After a rebuild ids are key ranks; routes
// added since carry later ids, so whenever a cap can bind the row is ordered by key BYTES first.  Keys come from the HBM key store
// in one gather for all rows.  A row whose length does not exceed either cap cannot be throttled and is not even classified
// (classified == false: its class counts are unknown, at most its length).
namespace {
struct CapEvent {
    int32_t type; // 0 PersistentFanoutThrottled, 1 GroupFanoutThrottled
    uint32_t id;
};
struct CappedRow {
    std::vector<uint32_t> kept; // ascending ids
    std::vector<CapEvent> events;
    uint32_t n_persistent = 0, n_group = 0; // among kept (only if classified)
    bool classified = false;
};
int cap_rows(bmq_engine* e, const uint32_t* rp, const uint32_t* ids, uint32_t n_rows, int32_t max_pf, int32_t max_gf, std::vector<CappedRow>& out) {
    out.assign(n_rows, CappedRow{});
    const int64_t lim = std::min<int64_t>(max_pf, max_gf);
    std::vector<uint32_t> need_row, g_ids;
    std::vector<uint64_t> g_rp{0};
    for (uint32_t r = 0; r < n_rows; r++) {
        const uint32_t n = rp[r + 1] - rp[r];
        if ((int64_t)n <= lim) {
            out[r].kept.assign(ids + rp[r], ids + rp[r + 1]);
            continue;
        }
        need_row.push_back(r);
        g_ids.insert(g_ids.end(), ids + rp[r], ids + rp[r + 1]);
        g_rp.push_back(g_ids.size());
    }
    if (need_row.empty()) return BMQ_OK;
    std::vector<uint8_t> kb;
    std::vector<uint64_t> ko;
    {
        std::lock_guard<std::mutex> g(e->mu);
        if (int rc_open = complete_apply(e)) return rc_open; // (a batch handed over with bmq_routes_apply_async first: its bookkeeping is read below)
        if (e->device >= 0) HIPCHK(e, hipSetDevice(e->device));
        const bool ok = with_index(e, [&](auto& ix) {
            const bool r = ix.route_keys(g_ids.data(), (uint32_t)g_ids.size(), kb, ko);
            if (!r) e->err = ix.error;
            return r;
        });
        if (!ok) return index_error(e, e->err, false);
    }
    struct Ent {
        uint32_t id;
        uint64_t k;
        uint8_t cls; // 0 transient normal, 1 persistent normal, 2 group
    };
    std::vector<Ent> row;
    for (uint32_t q = 0; q < need_row.size(); q++) {
        CappedRow& o = out[need_row[q]];
        row.clear();
        int64_t n_pers = 0, n_grp = 0;
        for (uint64_t k = g_rp[q]; k < g_rp[q + 1]; k++) {
            const std::string_view key((const char*)kb.data() + ko[k], (size_t)(ko[k + 1] - ko[k]));
            if (key.empty()) continue; // unsubscribed between the match and this lookup: the route no longer exists
            RouteKeyParts kp;
            if (!decode_route_key(key, kp)) return set_err(e, BMQ_E_INVAL, "corrupt key store");
            uint8_t cls = 2;
            if (kp.flag == 1) {
                // subBrokerId = integer prefix of "<brokerId>\0<receiverId>\0<delivererKey>" (SCHEMA/KVSchemaUtil.java:56-58)
                long broker = 0;
                size_t p = 0;
                while (p < kp.receiver.size() && kp.receiver[p] >= '0' && kp.receiver[p] <= '9') broker = broker * 10 + (kp.receiver[p++] - '0');
                cls = (broker == 1 && p > 0) ? 1 : 0;
            }
            n_pers += cls == 1;
            n_grp += cls == 2;
            row.push_back({g_ids[k], k, cls});
        }
        if (n_pers > (int64_t)max_pf || n_grp > (int64_t)max_gf)
            std::sort(row.begin(), row.end(), [&](const Ent& x, const Ent& y) {
                const std::string_view kx((const char*)kb.data() + ko[x.k], (size_t)(ko[x.k + 1] - ko[x.k]));
                const std::string_view ky((const char*)kb.data() + ko[y.k], (size_t)(ko[y.k + 1] - ko[y.k]));
                return kx < ky; // std::string_view compares as unsigned bytes, a proper prefix first: KV order
            });
        int64_t persistent = 0, groups = 0;
        for (const Ent& en : row) {
            int ev_type = -1;
            if (en.cls == 1) {
                if (persistent < (int64_t)max_pf) persistent++;
                else ev_type = 0;
            } else if (en.cls == 2) {
                if (groups + 1 <= (int64_t)max_gf) groups++;
                else ev_type = 1;
            }
            if (ev_type < 0) o.kept.push_back(en.id);
            else o.events.push_back({ev_type, en.id});
        }
        o.classified = true;
        o.n_persistent = (uint32_t)persistent;
        o.n_group = (uint32_t)groups;
        std::sort(o.kept.begin(), o.kept.end());
    }
    return BMQ_OK;
}
} // namespace

// ---- host-side mirror of MatchedRoutes (DW/cache/MatchedRoutes.java:87-141) -------------------------------------------
int bmq_match_all(bmq_engine* e, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics, const uint32_t* topic_off,
                  uint32_t n_topics, int32_t max_pf, int32_t max_gf, uint32_t* out_row_ptr, uint32_t* out_route_ids,
                  uint64_t out_capacity, uint64_t* out_needed, int32_t* out_events, uint32_t events_cap,
                  uint32_t* out_n_events) {
    std::unique_lock<std::recursive_mutex> api_lock;
    if (e) api_lock = std::unique_lock<std::recursive_mutex>(e->api);
    int rc = check_dist_ready(e);
    if (rc) return rc;
    if (!out_row_ptr || !out_needed) return set_err(e, BMQ_E_INVAL, "null output pointer");
    if (out_n_events) *out_n_events = 0;
    *out_needed = 0;
    if (n_topics == 0) {
        out_row_ptr[0] = 0;
        return BMQ_OK;
    }
    // matchAll takes a Set<String>: identical topics share one MatchedRoutes (TenantRouteMatcher.java:73-78)
    std::map<std::string_view, uint32_t> first;
    std::vector<uint32_t> canon(n_topics);
    for (uint32_t i = 0; i < n_topics; i++) {
        const std::string_view tp((const char*)topics + topic_off[i], topic_off[i + 1] - topic_off[i]);
        canon[i] = first.emplace(tp, i).first->second;
    }
    const uint32_t toff[2] = {0, tenant_len};
    std::vector<uint32_t> tt(n_topics, 0), rp(n_topics + 1);
    std::vector<uint32_t> ids(std::max<uint64_t>(1024, (uint64_t)n_topics * 8));
    uint64_t need = 0;
    rc = bmq_match_batch(e, tenant, toff, 1, tt.data(), topics, topic_off, n_topics, rp.data(), ids.data(), ids.size(), &need);
    if (rc == BMQ_E_NOSPACE) {
        ids.resize(need);
        rc = bmq_match_batch(e, tenant, toff, 1, tt.data(), topics, topic_off, n_topics, rp.data(), ids.data(), ids.size(), &need);
    }
    if (rc) return rc;
    // MatchedRoutes applies its caps first-come in KV KEY order (DW/cache/MatchedRoutes.java:87-141): cap_rows
    std::vector<std::vector<uint32_t>> kept(n_topics);
    uint32_t n_ev = 0;
    {
        std::vector<uint32_t> u_rp{0}, u_ids, u_topic; // the distinct topics' rows
        for (uint32_t i = 0; i < n_topics; i++)
            if (canon[i] == i) {
                u_ids.insert(u_ids.end(), ids.begin() + rp[i], ids.begin() + rp[i + 1]);
                u_rp.push_back((uint32_t)u_ids.size());
                u_topic.push_back(i);
            }
        std::vector<CappedRow> rows;
        if ((rc = cap_rows(e, u_rp.data(), u_ids.data(), (uint32_t)u_topic.size(), max_pf, max_gf, rows))) return rc;
        for (uint32_t u = 0; u < u_topic.size(); u++) {
            kept[u_topic[u]] = std::move(rows[u].kept);
            for (const auto& ev : rows[u].events) {
                if (out_events && n_ev < events_cap) {
                    int32_t* o = out_events + 4 * (size_t)n_ev;
                    o[0] = ev.type; o[1] = (int32_t)u_topic[u]; o[2] = (int32_t)ev.id; o[3] = ev.type == 0 ? max_pf : max_gf;
                }
                n_ev++;
            }
        }
    }
    if (out_n_events) *out_n_events = n_ev;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_topics; i++) {
        out_row_ptr[i] = (uint32_t)total;
        total += kept[canon[i]].size();
    }
    out_row_ptr[n_topics] = (uint32_t)total;
    *out_needed = total;
    if (total > out_capacity || (total && !out_route_ids)) return set_err(e, BMQ_E_NOSPACE, "output buffer too small");
    for (uint32_t i = 0; i < n_topics; i++) {
        const auto& v = kept[canon[i]];
        if (!v.empty()) memcpy(out_route_ids + out_row_ptr[i], v.data(), v.size() * 4);
    }
    return BMQ_OK;
}

int bmq_routes_cap(bmq_engine* e, const uint32_t* row_ptr, const uint32_t* route_ids, uint32_t n_rows, int32_t max_pf, int32_t max_gf,
                   uint32_t* out_row_ptr, uint32_t* out_route_ids, uint32_t* out_class_counts, int32_t* out_events, uint32_t events_cap,
                   uint32_t* out_n_events) {
    if (!e || !row_ptr || !out_row_ptr || (row_ptr[n_rows] && (!route_ids || !out_route_ids))) return BMQ_E_INVAL;
    if (out_n_events) *out_n_events = 0;
    if (!e->built) return set_err(e, BMQ_E_STATE, "bmq_rebuild has not been called");
    std::vector<CappedRow> rows;
    const int rc = cap_rows(e, row_ptr, route_ids, n_rows, max_pf, max_gf, rows);
    if (rc) return rc;
    uint32_t total = 0, n_ev = 0;
    for (uint32_t r = 0; r < n_rows; r++) {
        out_row_ptr[r] = total;
        if (!rows[r].kept.empty()) memcpy(out_route_ids + total, rows[r].kept.data(), rows[r].kept.size() * 4); // never longer than the input row
        total += (uint32_t)rows[r].kept.size();
        if (out_class_counts) {
            out_class_counts[2 * r] = rows[r].classified ? rows[r].n_persistent : 0xFFFFFFFFu;
            out_class_counts[2 * r + 1] = rows[r].classified ? rows[r].n_group : 0xFFFFFFFFu;
        }
        for (const auto& ev : rows[r].events) {
            if (out_events && n_ev < events_cap) {
                int32_t* o = out_events + 4 * (size_t)n_ev;
                o[0] = ev.type; o[1] = (int32_t)r; o[2] = (int32_t)ev.id; o[3] = ev.type == 0 ? max_pf : max_gf;
            }
            n_ev++;
        }
    }
    out_row_ptr[n_rows] = total;
    if (out_n_events) *out_n_events = n_ev;
    return BMQ_OK;
}

// ---- codec ------------------------------------------------------------------------------------------------------------
uint32_t bmq_route_key_encode(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filter, uint32_t filter_len,
                              uint8_t flag, const uint8_t* receiver, uint32_t receiver_len, uint8_t* out, uint32_t cap) {
    const std::string k = encode_route_key(std::string_view((const char*)tenant, tenant_len),
                                           std::string_view((const char*)filter, filter_len), flag,
                                           std::string_view((const char*)receiver, receiver_len));
    if (out && k.size() <= cap) memcpy(out, k.data(), k.size());
    return (uint32_t)k.size();
}

int bmq_route_key_decode(const uint8_t* key, uint32_t key_len, uint32_t spans[6]) {
    RouteKeyParts kp;
    const std::string_view k((const char*)key, key_len);
    if (!key || !spans || !decode_route_key(k, kp)) return BMQ_E_INVAL;
    spans[0] = (uint32_t)(kp.tenant.data() - k.data());
    spans[1] = (uint32_t)kp.tenant.size();
    spans[2] = (uint32_t)(kp.esc_filter.data() - k.data());
    spans[3] = (uint32_t)kp.esc_filter.size();
    spans[4] = (uint32_t)(kp.receiver.data() - k.data());
    spans[5] = (uint32_t)kp.receiver.size();
    return kp.flag;
}

// ---- retain store key schema ---------------------------------------------------------------------------------
uint32_t bmq_retain_message_key(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len, uint8_t* out, uint32_t cap) {
    const std::string k = retain_message_key(std::string_view((const char*)tenant, tenant_len), std::string_view((const char*)topic, topic_len));
    if (out && k.size() <= cap) memcpy(out, k.data(), k.size());
    return (uint32_t)k.size();
}
int bmq_retain_filter_route(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filter, uint32_t filter_len, uint8_t* out_key_prefix,
                            uint32_t cap, uint32_t* out_key_len, uint8_t* out_level_hash, uint32_t hash_cap, uint32_t* out_hash_len,
                            uint32_t* out_levels) {
    if (!out_key_len || !out_hash_len || !out_levels) return BMQ_E_INVAL;
    const RetainFilterRoute r = retain_filter_route(std::string_view((const char*)tenant, tenant_len), std::string_view((const char*)filter, filter_len));
    *out_key_len = (uint32_t)r.key_prefix.size();
    *out_hash_len = (uint32_t)r.level_hash.size();
    *out_levels = r.levels;
    if (r.key_prefix.size() > cap || r.level_hash.size() > hash_cap) return BMQ_E_NOSPACE;
    if (out_key_prefix && !r.key_prefix.empty()) memcpy(out_key_prefix, r.key_prefix.data(), r.key_prefix.size());
    if (out_level_hash && !r.level_hash.empty()) memcpy(out_level_hash, r.level_hash.data(), r.level_hash.size());
    return (r.wildcard ? 1 : 0) | (r.multi ? 2 : 0);
}

int32_t bmq_java_string_hash(const uint8_t* utf8, uint32_t len) { return java_string_hash(std::string_view((const char*)utf8, len)); }

} // extern "C"

#include "bmq_retain_engine.inc"
#include "bmq_poller.inc"
#include "bmq_batcher.inc"
#include "bmq_range_engine.inc"
#include "bmq_exchange.inc"
#include "bmq_fanout_engine.inc"
#include "bmq_formats.inc"
