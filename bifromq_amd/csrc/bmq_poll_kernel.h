// bmq_poll_kernel.h -- k_poll: the PERSISTENT matcher of the batching front (bmq_batcher.inc).
//
// Production asks for one topic per call (TenantRouteCache.java:180-193: one matchAll(singleton(topic)) per cache miss from the
// matchExecutor pool, DW/DistWorkerCoProcFactory.java:74-88); the batching front collects what is missing right now into a generation of a
// few dozen topics.  Up to round 5 every generation was a LAUNCH (k_walk + k_expand + the counters' way back + an event to wait for):
// 58-67 us for 1-256 topics, of which the kernels are a few -- the rest is launch and completion latency.  k_poll takes the launch out:
// a handful of one-wave workgroups stay resident and poll a ring of request descriptors in page-locked host memory; the leader of a
// generation fills the slot's input blob (in place, as before), writes the descriptor and rings the doorbell (a sequence number); the wave
// that owns the slot runs walk_wave<MIXED> and expand_wave -- the very code of k_walk and k_expand -- on it, writes the CSR in place into the
// slot's page-locked output blob and publishes the sequence number in the slot's completion word, which the leader spins on.
//
// What keeps this safe:
//   * the index is IMMUTABLE while the poller runs.  Every entry point that changes index memory (bmq_rebuild, bmq_routes_apply[_async],
//     bmq_compact*, destroy) stops the poller first (exit flag + stream synchronisation, bmq_poller.inc) and it is started again by the next
//     generation that wants it: kernel boundaries stay the only fences between index writers and readers, as everywhere in this engine;
//   * its lifetime is bounded three ways: an exit flag the host raises, an idle limit (no doorbell for POLL_IDLE_TICKS) and a hard limit
//     (POLL_LIFE_TICKS) -- a poller nobody talks to leaves the GPU by itself, a wedged one cannot hold it (the host falls back to launches);
//   * a generation the wave cannot finish by itself (a topic deeper than FAST_LEVELS, a buffer of the wave too small, rows that need the
//     sort fix-up) is handed back with POLL_FALLBACK: the leader launches it the old way.  Rare by construction (generations are small).
#pragma once

namespace bmq {

constexpr uint32_t POLL_SLOTS = 16;       // ring slots = generations that can be in the ring at once
#ifndef BMQ_POLL_WAVES
#define BMQ_POLL_WAVES 8
#endif
constexpr uint32_t POLL_WAVES = BMQ_POLL_WAVES; // resident one-wave workgroups; wave w serves the slots s with s % POLL_WAVES == w
constexpr uint32_t POLL_MAX_TOPICS = 64;  // one wave's worth (larger generations are launches: the chip is the better place for them)
constexpr unsigned long long POLL_TICK_HZ = 100000000ull;          // s_memrealtime counts the 100 MHz reference clock (s_memtime: shader clocks)
constexpr unsigned long long POLL_IDLE_TICKS = POLL_TICK_HZ / 50;  // 20 ms without a doorbell: leave (the next generation starts a new one)
constexpr unsigned long long POLL_LIFE_TICKS = POLL_TICK_HZ * 2;   // 2 s: leave whatever happens
enum : uint32_t { POLL_OK = 0, POLL_FALLBACK = 1, POLL_BAD_INPUT = 2 };

struct alignas(64) PollDesc { // host -> device, one per slot, page-locked host memory
    // the slot's input blob: topics at 0 | tenants | tenant_off | topic_tenant | topic_off (byte offsets), in_bytes in all
    uint32_t o_tenants, o_toff, o_tt, o_poff, in_bytes;
    uint32_t n_tenants, n_topics;
    uint32_t seq;  // the doorbell, written LAST: served + 1
    uint32_t pad[8];
};
struct alignas(64) PollDone { // device -> host, one per slot, page-locked host memory
    unsigned long long total, n_visit, n_ranges, topic_bytes;
    uint32_t status;      // POLL_OK / POLL_FALLBACK
    uint32_t batch_status; // Counters.status of the generation (ST_NOSPACE, ST_RANGE: the leader's business)
    uint32_t seq;         // written LAST: the generation with this sequence number is complete
    uint32_t pad[5];
};
struct alignas(64) PollCtl { // page-locked host memory
    uint32_t exit;        // host -> device: leave after the generation in hand
    uint32_t ignore;      // host -> device (test hook): doorbells are not answered -- the leader's time-out path
    uint32_t exited;      // device -> host: waves that have left (a poller with exited != 0 is not handed anything any more)
    uint32_t pad;
};

struct PollArgs {
    DistIndexView ix;
    const PollDesc* desc;
    PollDone* done;
    PollCtl* ctl;
    const uint8_t* blob_in;  // page-locked: POLL_SLOTS input blobs of in_stride bytes
    uint8_t* blob_out;       // page-locked: POLL_SLOTS output blobs of out_stride bytes: total (8 B) | pad | row_ptr at 16 | ids at out_ids_off
    uint32_t in_stride, out_stride, out_ids_off, out_ids_cap;
    uint8_t* dev_in;         // device memory: one copy of an input blob per wave (in_stride bytes each)
    // per-wave scratch in device memory, [POLL_WAVES] pieces each
    uint32_t* pair_off;   // [64] per wave
    uint32_t* pair_cnt;
    uint32_t* route_cnt;
    MatchRange* pairs;    // pair_cap per wave
    unsigned long long pair_cap;
    SubAlloc* subs;       // 2 * N_SUB per wave
    unsigned long long* super_sums; // SUPER_STRIDE per wave
    uint4* blk_stats;     // 1 per wave
    uint4* spill;         // spill_cap per wave
    unsigned long long spill_cap;
    unsigned long long* wave_sums; // 1 per wave
    uint32_t* slow_list;  // 64 per wave
    uint32_t* sort_list;  // 64 per wave
    Counters* ctr;        // 1 per wave
    unsigned long long* t_work; // device memory: when ANY wave last had a doorbell (the idle limit is the poller's, not a wave's)
};

#ifndef BMQ_WAVE_EMU
// host memory, system scope, around every cache: the doorbells and the control words
__device__ __forceinline__ uint32_t poll_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void poll_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

using PollGeom = WalkLds<BMQ_WALK_GEOM_DEFAULT, true>; // the MIXED instantiation: a generation holds requests of any tenants, in arrival order

__global__ __launch_bounds__(64) void k_poll(PollArgs p) {
    __shared__ __align__(16) uint32_t lds[PollGeom::BYTES / 4];
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    constexpr uint32_t OWN = POLL_SLOTS / POLL_WAVES;
    static_assert(POLL_SLOTS % POLL_WAVES == 0, "every wave owns the same number of slots");
    uint32_t served[OWN];
#pragma unroll
    for (uint32_t k = 0; k < OWN; k++) served[k] = sgpr(poll_load(&p.done[w + k * POLL_WAVES].seq)); // where the last launch left off
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) atomicMax(p.t_work, t0);
    Counters* const ctr = p.ctr + w;
    SubAlloc* const subs = p.subs + (size_t)w * 2 * N_SUB;
    for (;;) {
        bool any = false;
#pragma unroll
        for (uint32_t k = 0; k < OWN; k++) {
            const uint32_t s = w + k * POLL_WAVES;
            const PollDesc* const d = p.desc + s;
            const uint32_t seq = sgpr(poll_load(&d->seq));
            const uint32_t hook = seq == served[k] + 1u ? sgpr(poll_load(&p.ctl->ignore)) : 0u;
            if (hook == 1u) any = true; // (the test hook plays a wedged wave: it does not answer and does not leave by itself either)
            if (seq != served[k] + 1u || hook == 1u) continue;
            any = true;
            // ---- the generation's argument block: what launch_dist fills for a launch, from the descriptor and this wave's scratch ----
            // The batch behind the doorbell is COPIED from the slot's page-locked blob into this wave's device buffer with system-scope loads,
            // and the walk runs on the copy.  (First version: the walk read the blob in host memory with ordinary loads -- and got lines of
            // the slot's PREVIOUS generation out of the GPU's caches now and then: offsets of an old batch read as offsets of the new one,
            // a level scan running off into unmapped memory as soon as two waves were busy at once, and -- worse -- silently wrong rows
            // where nothing faulted.  A launch gets clean caches from the command processor; a resident wave reads host memory around them.)
            BatchArgs a{};
            a.ix = p.ix;
            uint32_t dw[8];
            {
                const unsigned long long* q = reinterpret_cast<const unsigned long long*>(d);
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const unsigned long long x = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    dw[2 * i] = sgpr((uint32_t)x), dw[2 * i + 1] = sgpr((uint32_t)(x >> 32));
                }
            }
            const uint32_t o_tenants = dw[0], o_toff = dw[1], o_tt = dw[2], o_poff = dw[3], in_bytes = min(dw[4], p.in_stride);
            a.n_tenants = dw[5];
            a.n_topics = min(dw[6], POLL_MAX_TOPICS);
            uint8_t* const din = p.dev_in + (size_t)w * p.in_stride;
            {
                const unsigned long long* src = reinterpret_cast<const unsigned long long*>(p.blob_in + (size_t)s * p.in_stride);
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(din);
                const uint32_t n8 = (in_bytes + 7u) >> 3;
                for (uint32_t i0 = 0; i0 < n8; i0 += 256u) { // four requests per lane in flight: 2 KB per trip over the link
                    unsigned long long v[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) {
                        const uint32_t i = i0 + j * 64u + lane;
                        v[j] = i < n8 ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
                    }
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) {
                        const uint32_t i = i0 + j * 64u + lane;
                        if (i < n8) dst[i] = v[j];
                    }
                }
            }
            a.topics = din;
            a.tenants = din + min(o_tenants, p.in_stride);
            a.tenant_off = reinterpret_cast<const uint32_t*>(din + (min(o_toff, p.in_stride) & ~3u));
            a.topic_tenant = reinterpret_cast<const uint32_t*>(din + (min(o_tt, p.in_stride) & ~3u));
            a.topic_off = reinterpret_cast<const uint32_t*>(din + (min(o_poff, p.in_stride) & ~3u));
            {
                uint8_t* const ob = p.blob_out + (size_t)s * p.out_stride;
                a.out_total = reinterpret_cast<unsigned long long*>(ob);
                a.out_row_ptr = reinterpret_cast<uint32_t*>(ob + 16);
                a.out_ids = reinterpret_cast<uint32_t*>(ob + p.out_ids_off);
                a.out_capacity = p.out_ids_cap;
            }
            // this wave's own stores (the copy above, its scratch of a generation ago) are read back below: around this CU's vector L1 and
            // the scalar cache (the walk reads the offsets of its first and last topic with s_load)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __builtin_amdgcn_s_dcache_inv();
            a.pair_off = p.pair_off + w * 64u, a.pair_cnt = p.pair_cnt + w * 64u, a.route_cnt = p.route_cnt + w * 64u;
            a.pairs = p.pairs + (size_t)w * p.pair_cap, a.pair_cap = p.pair_cap;
            a.subs = subs;
            a.super_sums = p.super_sums + (size_t)w * SUPER_STRIDE;
            a.blk_stats = p.blk_stats + w;
            a.spill = p.spill + (size_t)w * p.spill_cap, a.spill_cap = p.spill_cap;
            a.wave_sums = p.wave_sums + w;
            a.n_blocks = 1, a.tpw_shift = 6;
            a.slow_list = p.slow_list + w * 64u, a.slow_cap = 64;
            a.scratch = nullptr, a.scratch_cap = 0;
            a.sort_list = p.sort_list + w * 64u, a.sort_cap = 64;
            a.ctr = ctr;
            a.qcap = WALK_QC_DEFAULT, a.pcap = WALK_PC_DEFAULT;
            // What the wave is about to trust, looked at first: every piece inside the copy, offsets ascending, tenant indices inside the
            // batch's tenant table.  A generation that fails is handed back untouched (POLL_BAD_INPUT): nothing a host thread wrote can make a
            // resident wave run off into unmapped memory.
            {
                const uint32_t end_toff = o_toff + 4u * (a.n_tenants + 1u), end_tt = o_tt + 4u * a.n_topics, end_poff = o_poff + 4u * (a.n_topics + 1u);
                bool bad = a.n_tenants > 64u || o_tenants > o_toff || end_toff > o_tt || end_tt > o_poff || end_poff + 16u > in_bytes || ((o_toff | o_tt | o_poff) & 3u);
                if (!bad && lane < a.n_topics) {
                    const uint32_t o0 = a.topic_off[lane], o1 = a.topic_off[lane + 1], tt = a.topic_tenant[lane];
                    bad = o0 > o1 || o1 + 16u > o_tenants || o1 - o0 > 65535u || tt >= a.n_tenants;
                }
                if (!bad && lane <= a.n_tenants) {
                    const uint32_t o0 = a.tenant_off[lane];
                    bad = o_tenants + o0 + 16u > o_toff || (lane > 0 && o0 < a.tenant_off[lane - 1]);
                }
                if (ballot64(bad) != 0ull) {
                    PollDone* const dn = p.done + s;
                    if (lane == 0) dn->total = 0ull, dn->status = POLL_BAD_INPUT, dn->batch_status = 0u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                    if (lane == 0) poll_store(&dn->seq, seq);
                    served[k] = seq;
                    continue;
                }
            }
            // k_reset's work for this wave's one-block batch
            if (lane < sizeof(Counters) / 8) reinterpret_cast<unsigned long long*>(ctr)[lane] = 0ull;
            if (lane == 0) subs[0].used = 0ull, subs[N_SUB].used = 0ull, a.super_sums[0] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#if BMQ_EXPERIMENTS
            if (hook == 2u) { // (bring-up) answer without touching the blobs
                if (lane == 0) poll_store(&p.done[s].seq, seq);
                served[k] = seq;
                continue;
            }
            if (hook == 3u) { // (bring-up) read the input blob, touch the output blob, answer
                uint32_t acc = lane < a.n_topics ? a.topic_off[lane + 1] + a.topic_tenant[lane] : 0u;
                acc += a.tenant_off[0] + a.topics[0] + a.tenants[0];
                if (lane <= a.n_topics) a.out_row_ptr[lane] = 0u;
                if (lane == 0) *a.out_total = acc & 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                if (lane == 0) p.done[s].total = 0ull, p.done[s].status = POLL_OK, p.done[s].batch_status = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                if (lane == 0) poll_store(&p.done[s].seq, seq);
                served[k] = seq;
                continue;
            }
#endif
            walk_wave<BMQ_WALK_GEOM_DEFAULT, true>(a, 0u, lds);
#if BMQ_EXPERIMENTS
            if (hook == 4u) { // (bring-up) the walk only
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                if (lane <= a.n_topics) a.out_row_ptr[lane] = 0u;
                if (lane == 0) p.done[s].total = 0ull, p.done[s].status = POLL_OK, p.done[s].batch_status = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                if (lane == 0) poll_store(&p.done[s].seq, seq);
                served[k] = seq;
                continue;
            }
#endif
            // what the walk left in this wave's scratch is read back below (status, the ranges): around this CU's L1
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            uint32_t st = sgpr(__hip_atomic_load(&ctr->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const uint32_t n_slow = sgpr(__hip_atomic_load(&ctr->slow_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            uint32_t verdict = ((st & (ST_RERUN | ST_WANT_MIXED)) != 0 || n_slow != 0) ? (uint32_t)POLL_FALLBACK : (uint32_t)POLL_OK;
            if (verdict == POLL_OK) {
                expand_wave(a, 0u);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                st = sgpr(__hip_atomic_load(&ctr->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (sgpr(__hip_atomic_load(&ctr->sort_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) verdict = POLL_FALLBACK; // rows for k_sort_rows
            }
            PollDone* const dn = p.done + s;
            if (lane == 0) {
                dn->total = __hip_atomic_load(&ctr->total_ids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dn->n_visit = __hip_atomic_load(&ctr->n_visit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dn->n_ranges = __hip_atomic_load(&ctr->n_ranges, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dn->topic_bytes = __hip_atomic_load(&ctr->topic_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dn->status = verdict;
                dn->batch_status = st;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // every lane's stores into the output blob and lane 0's above ...
            if (lane == 0) poll_store(&dn->seq, seq);                 // ... are in host memory before the completion word
            served[k] = seq;
        }
        // The idle limit is the POLLER's: a wave whose slots are quiet stays while others are busy -- a poller that has lost some of its
        // waves would leave their slots unanswered, and the leaders that rang them waiting for the rest to end.
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (any && lane == 0) atomicMax(p.t_work, now);
        const unsigned long long tw = __hip_atomic_load(p.t_work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t idle = now > tw && now - tw > POLL_IDLE_TICKS ? 1u : 0u;
        if (sgpr(poll_load(&p.ctl->exit)) != 0u || now - t0 > POLL_LIFE_TICKS || sgpr(idle) != 0u) break;
        if (!any) __builtin_amdgcn_s_sleep(8);
    }
    // Leaving: say so FIRST, then look at the doorbells once more -- a leader that rang after this wave's last look either sees `exited`
    // (and waits for the kernel to end before it decides) or is served here.
    if (lane == 0) {
        __hip_atomic_fetch_add(&p.ctl->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
}
#endif // BMQ_WAVE_EMU

} // namespace bmq
