// bmq_dist_kernels.h -- gfx950 (wave64) kernels of the dist-direction match path.
//
// Replaces, for a whole batch at once, what TenantRouteMatcher.matchAll does per topic on a
// matcher thread (DW/cache/TenantRouteMatcher.java:67-161): resolve every publish topic to the ids
// of all routes whose filter matches it.  Semantics follow SURVEY.md 8a-0 (TRIE/TopicTrieNode.java:150-152
// for the '$' rule, TRIE/NTopicFilterTrieNode.java:143-146 for '#' matching the parent level).
//
// Included only by bmq_engine.hip (hipcc --offload-arch=gfx950).  Work decomposition:
//   (no prologue)     : the batch's tenants are resolved by the walk kernels themselves (a wave's topics usually share one
//                       tenant: the directory lookup then runs on the scalar unit); the batch counters are zeroed BEHIND the
//                       previous batch of the slot (k_reset) -- a batch starts on its walk kernel.
//   k_walk<LV,QC,PC>  : (bmq_walk_kernel.h) one wave = one workgroup per 64 topics.
//                       Phase 1: tenant directory entry; the wave stages its topics' bytes in LDS with coalesced 16-byte loads, then walks
//                       them level by level: every lane scans its own next level (LDS only), then ALL lanes look
//                       their level up in the dictionary together -- one memory latency per level, not per lane.
//                       Phase 2: the wave drains a depth-first LDS work stack of (node, topic, level, kind) items,
//                       one item per lane per round; an item reads one 64-byte bucket: the home bucket of the edge
//                       (node id, token of the topic's level) or (node id, '+') -- the child's header comes with it.
//                       Pushes and matches are compacted with ballot + mbcnt.
//                       Phase 3: matched (begin,count) ranges are counting-sorted by topic and written out.
//   k_walk_slow       : per-lane DFS with global scratch for topics of more than FAST_LEVELS levels.  Launched only while
//                       batches have such topics (the finish step re-runs the batch that found out: bmq_engine.hip).
//   (no scan kernel)  : k_walk leaves per-wave id counts + their sums per 256 waves; every k_expand wave adds up the <= 61 + 255
//                       values in front of it (five coalesced loads per lane) instead of waiting for a single-workgroup scan.
//   k_expand          : per 64 rows: order each row's ranges by first id in LDS, lay them out in output order and fill the
//                       CSR ids with fully coalesced stores (short ranges located through a start bitmap, long ranges streamed);
//                       the last wave of every 256 sums the batch statistics k_walk left per wave.
//   k_sort_rows       : bitonic fix-up of rows whose range list was too long to order in k_expand; launched only while batches
//                       have such rows.
#pragma once
#ifndef BMQ_WAVE_EMU // (tools/emu/walk_emu.cpp compiles this file with g++ against the wave64 emulator: the device-only pieces step aside)
#include <hip/hip_runtime.h>
#endif

#include <type_traits>

#include "bmq_layout.h"
#include "bmq_batch_args.h"

namespace bmq {

// ------------------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------------------
// Waves of one workgroup are independent here (each owns a slice of LDS).  LDS operations of ONE wave execute in issue
// order, so lanes of a wave see each other's LDS writes without a hardware barrier; this only stops the compiler from
// moving LDS accesses across the hand-off point.
#ifndef BMQ_WAVE_EMU
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t rank_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
#endif
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += o;
    }
    total = __shfl(inc, 63);
    return inc - v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// trie access
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ TrieSlot unpack_slot(const uint4& a, const uint4& b) {
    TrieSlot s;
    s.parent = a.x; s.token = a.y; s.own_begin = a.z; s.own_count = a.w;
    s.hash_begin = b.x; s.hash_count = b.y; s.node = b.z; s.lit_bloom = b.w;
    return s;
}
// One aligned 64-byte line = one trie bucket (two TrieSlots) or one dictionary group (two DictSlots), requested with
// four 16-byte loads issued back to back and ONE wait.  Written as asm because the compiler otherwise splits the
// line into dependent pieces (first the 8 key bytes of slot 0, wait, then those of slot 1, wait, then the payload):
// three serial cache round trips per visited node instead of one.  The index is never written while a batch runs.
struct Line64 {
    uint4 a0, a1, b0, b1;
};
static_assert(DICT_GROUP == 2, "a dictionary group is one Line64");
#ifndef BMQ_WAVE_EMU
__device__ __forceinline__ void load_line64(const void* p, Line64& r) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                 "global_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\t"
                 "global_load_dwordx4 %3, %4, off offset:48\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(r.a0), "=&v"(r.a1), "=&v"(r.b0), "=&v"(r.b1)
                 : "v"(p));
}
// ... with a wave-uniform base (SGPR pair) and a 32-bit byte offset per lane: no 64-bit address arithmetic on the vector unit
__device__ __forceinline__ void load_line64_s(const void* sbase, uint32_t voff, Line64& r) {
    asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                 "global_load_dwordx4 %1, %4, %5 offset:16\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:32\n\t"
                 "global_load_dwordx4 %3, %4, %5 offset:48\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(r.a0), "=&v"(r.a1), "=&v"(r.b0), "=&v"(r.b1)
                 : "v"(voff), "s"(sbase));
}
#else // (emulator: the same bytes through memcpy)
inline void load_line64(const void* p, Line64& r) { memcpy(&r, p, 64); }
inline void load_line64_s(const void* sbase, uint32_t voff, Line64& r) { memcpy(&r, reinterpret_cast<const uint8_t*>(sbase) + voff, 64); }
#endif
// ------------------------------------------------------------------------------------------------------------
// dictionary lookup (exact: tag + length + bytes).  ONE latency: the whole home group (one 64-byte line) is requested together.
// byte_at(i) returns byte i of the string buffer the level lives in (LDS-staged or global).
// ------------------------------------------------------------------------------------------------------------
template <class ByteAt>
__device__ __forceinline__ uint32_t dict_lookup(const DistIndexView& ix, const LevelHash& h, uint32_t len,
                                                const uint32_t inl[4], uint32_t start, ByteAt&& byte_at) {
    const uint32_t tag = level_hash_tag(h);
    uint32_t g = level_hash_slot(h, len) & ix.dict_group_mask;
    auto tail_eq = [&](uint32_t pool_off) { // bytes beyond the 16 inline ones (rare: levels are short)
        bool eq = true;
        for (uint32_t i = 16; i < len && eq; i++) eq = ix.pool[pool_off + i] == byte_at(start + i);
        return eq;
    };
    for (uint32_t probes = 0; probes <= ix.dict_group_mask; probes++) { // bounded: a damaged table must not hang the GPU
        Line64 ln;
        load_line64(ix.dict + DICT_GROUP * (size_t)g, ln);
        const bool h0 = ln.a0.x == tag && ln.a0.z == len && ln.a1.x == inl[0] && ln.a1.y == inl[1] && ln.a1.z == inl[2] &&
                        ln.a1.w == inl[3];
        const bool h1 = ln.b0.x == tag && ln.b0.z == len && ln.b1.x == inl[0] && ln.b1.y == inl[1] && ln.b1.z == inl[2] &&
                        ln.b1.w == inl[3];
        if (h0 && (len <= 16 || tail_eq(ln.a0.w))) return ln.a0.y;
        if (h1 && (len <= 16 || tail_eq(ln.b0.w))) return ln.b0.y;
        if (ln.a0.x == 0 || ln.b0.x == 0) return TOK_UNKNOWN;
        g = (g + 1) & ix.dict_group_mask;
    }
    return TOK_UNKNOWN;
}

// The same lookup, one 32-byte SLOT at a time: the builder fills a group's slot 0 before its slot 1, and at load factor <= 1/4 most
// strings sit in slot 0 -- the second slot is requested only by the lanes that need it.  Half the bytes through the L1 / L2 path per lookup
// for a dependent second request now and then: right for k_walk, whose tokeniser runs under the other waves' walk rounds while the kernel
// as a whole sits at the memory system's rate of 64-byte lines (profiles/r04/k_walk_experiments.md).
#ifndef BMQ_WAVE_EMU
__device__ __forceinline__ void load_slot32(const void* p, uint4& a, uint4& b) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b)
                 : "v"(p));
}
#else
inline void load_slot32(const void* p, uint4& a, uint4& b) {
    memcpy(&a, p, 16);
    memcpy(&b, reinterpret_cast<const uint8_t*>(p) + 16, 16);
}
#endif
template <class ByteAt>
__device__ __forceinline__ uint32_t dict_lookup_by_slot(const DistIndexView& ix, const LevelHash& h, uint32_t len, const uint32_t inl[4], uint32_t start,
                                                        ByteAt&& byte_at) {
    const uint32_t tag = level_hash_tag(h);
    uint32_t g = level_hash_slot(h, len) & ix.dict_group_mask;
    auto tail_eq = [&](uint32_t pool_off) { // bytes beyond the 16 inline ones (rare: levels are short)
        bool eq = true;
        for (uint32_t i = 16; i < len && eq; i++) eq = ix.pool[pool_off + i] == byte_at(start + i);
        return eq;
    };
    for (uint32_t probes = 0; probes <= ix.dict_group_mask; probes++) { // bounded: a damaged table must not hang the GPU
        const DictSlot* grp = ix.dict + DICT_GROUP * (size_t)g;
#pragma unroll
        for (uint32_t k = 0; k < DICT_GROUP; k++) {
            uint4 hd, in;
            load_slot32(grp + k, hd, in);
            if (hd.x == tag && hd.z == len && in.x == inl[0] && in.y == inl[1] && in.z == inl[2] && in.w == inl[3] && (len <= 16 || tail_eq(hd.w))) return hd.y;
            if (hd.x == 0) return TOK_UNKNOWN; // a free slot ends the probe sequence (slots of a group fill in order)
        }
        g = (g + 1) & ix.dict_group_mask;
    }
    return TOK_UNKNOWN;
}

// Scans one level starting at pos: bytes up to the next '/' (split) or to `end`, FOUR BYTES PER STEP.
// word_at(p) returns the 4 bytes p..p+3 as a little-endian word (bytes at or beyond `end` may be garbage: masked here).
template <class WordAt>
__device__ __forceinline__ void scan_level(uint32_t& pos, uint32_t end, bool split, WordAt&& word_at, LevelHash& h,
                                           uint32_t inl[4], uint32_t& len, bool& last) {
    h = level_hash_init();
    inl[0] = inl[1] = inl[2] = inl[3] = 0;
    len = 0;
    for (;;) {
        const uint32_t remaining = end - pos;
        if (remaining == 0) break;
        const uint32_t w = word_at(pos);
        uint32_t nb = 4;
        if (split) { // first '/' among the four bytes (exact for the lowest hit, which is all that is used)
            const uint32_t x = w ^ 0x2F2F2F2Fu;
            const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
            if (z) nb = (uint32_t)(__ffs((int)z) - 1) >> 3;
        }
        nb = min(nb, remaining);
        if (nb) {
            const uint32_t wm = nb == 4 ? w : (w & ((1u << (8u * nb)) - 1u));
            level_hash_word(h, wm);
            if (len < 16) inl[len >> 2] = wm;
            len += nb;
            pos += nb;
        }
        if (nb < 4) break;
    }
    if (pos < end) {
        pos++; // skip the '/': another (possibly empty) level follows (UTIL/TopicUtil.java:206-225)
        last = false;
    } else {
        last = true;
    }
}

// 4 bytes at byte offset p of a global buffer (any alignment): two aligned dwords + byte align.  The buffer must be
// readable up to the next multiple of 4 after its last byte plus 4 (all packed inputs are padded by 16 bytes).
__device__ __forceinline__ uint32_t global_word_at(const uint8_t* base, uint32_t p) {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(base + (p & ~3u));
    return __builtin_amdgcn_alignbyte(a[1], a[0], p & 3u);
}

// ------------------------------------------------------------------------------------------------------------
// batch slot reset, tenant directory lookup
// ------------------------------------------------------------------------------------------------------------
constexpr TenantSlot EMPTY_TENANT{0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, {0, 0, 0}, NONE};
__device__ __forceinline__ bool tenant_known(const TenantSlot& t) { return (t.hash_lo | t.hash_hi) != 0; }

// k_reset: zeroes the batch counters / sub-allocators / super sums of a batch slot.  Enqueued BEHIND a batch (after its counters
// were copied out), so that the next batch on the slot starts on its walk kernel: in front of the batch the same work was measured
// at 19-40 us of a 0.42 ms step (profiles/r02, k_prologue), a chain of cold round trips that nothing overlapped.
__global__ __launch_bounds__(64) void k_reset(Counters* ctr, SubAlloc* subs, unsigned long long* super_sums, uint32_t n_super) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < sizeof(Counters) / 8) reinterpret_cast<unsigned long long*>(ctr)[i] = 0ull;
    if (i < sizeof(SubAlloc) * 2 * N_SUB / 8) reinterpret_cast<unsigned long long*>(subs)[i] = 0ull;
    if (i < n_super) super_sums[(size_t)i * SUPER_STRIDE] = 0ull;
}

// k_publish: the batch's counters -> the slot's page-locked host copy, behind the batch's last kernel.  Where it stands a 336-byte
// hipMemcpyAsync stood: that copy runs on an SDMA engine, i.e. on another hardware queue that has to be signalled and signals back -- for the
// small launches of the batching front (tens of topics, ~60 us in all) a kernel that stays in the compute queue is the shorter way
// (engine: publish_mode; the stores are visible to the host when the event behind the kernel has completed, like the results such
// launches write in place into page-locked memory).
static_assert(sizeof(Counters) % 16 == 0 && sizeof(Counters) / 16 <= 64, "one wave copies the counters, 16 bytes per lane");
__global__ __launch_bounds__(64) void k_publish(const Counters* ctr, Counters* host) {
    const uint32_t i = threadIdx.x;
    if (i < sizeof(Counters) / 16) reinterpret_cast<uint4*>(host)[i] = reinterpret_cast<const uint4*>(ctr)[i];
}

// The directory entry of a batch tenant (EMPTY_TENANT: no such tenant).  Resolved by the walk kernels themselves, per topic: the
// lanes of a wave mostly share one tenant (batches arrive grouped by tenant) and every wave of the launch keeps the few lines
// involved hot in L2 -- as a kernel of its own in front of the walk the same lookups were cold, dependent round trips.
// Three dependent requests (offsets -> id bytes -> directory slot); k_walk issues them one per tokeniser iteration, where each
// lands under the dictionary lookup's wait.
struct TenantQuery {
    uint32_t beg = 0, len = 0;    // the id's bytes in the batch's tenant table
    uint32_t raw[5] = {0, 0, 0, 0, 0}; // stage 0: the two offsets; stage 1: the aligned dwords around the first 16 bytes
    uint32_t w[4] = {0, 0, 0, 0}; // the first 16 bytes, zero padded
    uint32_t lo = 0, hi = 0, d = 0;
    TenantSlot slot = EMPTY_TENANT; // directory slot d
};
// A stage only REQUESTS its data and works on what the previous stage requested: nothing waits inside the stage it was asked in.
__device__ __forceinline__ void tenant_stage(const BatchArgs& a, uint32_t ti, TenantQuery& q, uint32_t st) {
    if (st == 0) {
        q.raw[0] = a.tenant_off[ti];
        q.raw[1] = a.tenant_off[ti + 1];
    } else if (st == 1) { // the first 16 bytes are requested together (a byte-wise loop is one memory latency per byte)
        q.beg = q.raw[0];
        q.len = q.raw[1] - q.raw[0];
        const uint32_t* p = reinterpret_cast<const uint32_t*>(a.tenants + (q.beg & ~3u));
#pragma unroll
        for (uint32_t k = 0; k < 5; k++) q.raw[k] = 4 * k < q.len + 3 ? p[k] : 0u; // (packed inputs are padded by 16 bytes)
    } else {
        const uint32_t sh = q.beg & 3u;
        uint64_t h = TENANT_HASH_INIT;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) { // (constant indices: the words stay in registers)
            const uint32_t w = __builtin_amdgcn_alignbyte(q.raw[k + 1], q.raw[k], sh);
            const uint32_t nb = q.len > 4 * k ? min(4u, q.len - 4 * k) : 0u;
            q.w[k] = nb >= 4 ? w : (w & ((1u << (8 * nb)) - 1u)); // zero padded like TenantSlot.name12
            for (uint32_t j = 0; j < 4; j++)
                if (4 * k + j < q.len) h = tenant_hash_step(h, (q.w[k] >> (8 * j)) & 0xFFu);
        }
        for (uint32_t k = 16; k < q.len; k += 4) {
            const uint32_t w = global_word_at(a.tenants, q.beg + k), nb = min(4u, q.len - k);
            for (uint32_t j = 0; j < nb; j++) h = tenant_hash_step(h, (w >> (8 * j)) & 0xFFu);
        }
        h = tenant_hash_final(h);
        q.lo = (uint32_t)h, q.hi = (uint32_t)(h >> 32);
        q.d = (q.lo ^ q.hi) & a.ix.tenant_mask;
        q.slot = a.ix.tenants[q.d];
    }
}
// after the three stages: the id's bytes decide; a displaced entry costs further directory slots
__device__ __forceinline__ TenantSlot tenant_verdict(const BatchArgs& a, const TenantQuery& q) {
    TenantSlot t = q.slot;
    uint32_t d = q.d;
    for (uint32_t probes = 0; probes <= a.ix.tenant_mask; probes++) {
        if (!tenant_known(t)) break;
        if (t.hash_lo == q.lo && t.hash_hi == q.hi && t.name_len == q.len) {
            // both sides are zero padded: the first three words compare whole
            bool eq = t.name12[0] == q.w[0] && t.name12[1] == q.w[1] && t.name12[2] == q.w[2];
            for (uint32_t k = 12; k < q.len && eq; k += 4) {
                const uint32_t nb = min(4u, q.len - k), m = nb == 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
                eq = ((global_word_at(a.ix.tenant_names, t.name_off + k) ^ global_word_at(a.tenants, q.beg + k)) & m) == 0;
            }
            if (eq) return t;
        }
        d = (d + 1) & a.ix.tenant_mask;
        t = a.ix.tenants[d];
    }
    return EMPTY_TENANT;
}
// The same lookup for a wave-uniform tenant, on the scalar unit: the loads go through the constant address space (s_load, scalar
// cache), which is legal because neither the batch's tenant table nor the index is written while a batch runs.
#ifndef BMQ_WAVE_EMU
typedef const uint32_t __attribute__((address_space(4))) * ScalarWords;
#else
typedef const uint32_t* ScalarWords;
#endif
__device__ __forceinline__ ScalarWords scalar_words(const void* p) { return (ScalarWords)(uintptr_t)p; }
__device__ __forceinline__ uint32_t scalar_word_at(const void* base, uint32_t p) { // 4 bytes at byte offset p, any alignment
    ScalarWords q = scalar_words(base) + (p >> 2);
    return __builtin_amdgcn_alignbyte(q[1], q[0], p & 3u);
}
__device__ __forceinline__ TenantSlot resolve_tenant_uniform(const BatchArgs& a, uint32_t ti) {
    ScalarWords off = scalar_words(a.tenant_off);
    const uint32_t beg = off[ti], len = off[ti + 1] - beg;
    uint32_t w[4];
    uint64_t h = TENANT_HASH_INIT;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t x = 4 * k < len ? scalar_word_at(a.tenants, beg + 4 * k) : 0u;
        const uint32_t nb = len > 4 * k ? min(4u, len - 4 * k) : 0u;
        w[k] = nb >= 4 ? x : (x & ((1u << (8 * nb)) - 1u));
        for (uint32_t j = 0; j < 4; j++)
            if (4 * k + j < len) h = tenant_hash_step(h, (w[k] >> (8 * j)) & 0xFFu);
    }
    for (uint32_t k = 16; k < len; k += 4) {
        const uint32_t x = scalar_word_at(a.tenants, beg + k), nb = min(4u, len - k);
        for (uint32_t j = 0; j < nb; j++) h = tenant_hash_step(h, (x >> (8 * j)) & 0xFFu);
    }
    h = tenant_hash_final(h);
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
    uint32_t d = (lo ^ hi) & a.ix.tenant_mask;
    for (uint32_t probes = 0; probes <= a.ix.tenant_mask; probes++) {
        ScalarWords sp = scalar_words(a.ix.tenants + d);
        TenantSlot t;
        t.hash_lo = sp[0], t.hash_hi = sp[1], t.name_off = sp[2], t.name_len = sp[3];
        t.base = sp[4], t.buckets = sp[5], t.n_nodes = sp[6], t.n_routes = sp[7];
        t.root_hash_begin = sp[8], t.root_hash_count = sp[9], t.root_lit_bloom = sp[10], t.pending = sp[11];
        t.name12[0] = sp[12], t.name12[1] = sp[13], t.name12[2] = sp[14], t.root_plus = sp[15];
        if (!tenant_known(t)) break;
        if (t.hash_lo == lo && t.hash_hi == hi && t.name_len == len) {
            bool eq = t.name12[0] == w[0] && t.name12[1] == w[1] && t.name12[2] == w[2];
            for (uint32_t k = 12; k < len && eq; k += 4) {
                const uint32_t nb = min(4u, len - k), m = nb == 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
                eq = ((scalar_word_at(a.ix.tenant_names, t.name_off + k) ^ scalar_word_at(a.tenants, beg + k)) & m) == 0;
            }
            if (eq) return t;
        }
        d = (d + 1) & a.ix.tenant_mask;
    }
    return EMPTY_TENANT;
}
__device__ __forceinline__ TenantSlot resolve_tenant(const BatchArgs& a, uint32_t ti) {
    TenantQuery q;
    tenant_stage(a, ti, q, 0);
    tenant_stage(a, ti, q, 1);
    tenant_stage(a, ti, q, 2);
    return tenant_verdict(a, q);
}

// ------------------------------------------------------------------------------------------------------------
// the per-item step shared by the LDS path and the slow path
// ------------------------------------------------------------------------------------------------------------
// item = (node id, meta); meta: bits 0-5 topic-local index, bits 6-29 level, bit 31 kind
//   kind L (0): probe the child of `node` labelled with the topic's token at `level`
//   kind P (1): probe the '+' child of `node` (it consumes the topic's level `level` whatever its token)
constexpr uint32_t KIND_P = 0x80000000u;
// (k_walk_slow only) kind P with KIND_DIRECT: the item's first word is not the parent's node id but the SLOT (relative to the region) of the
// '+' child itself -- found beside its parent (layout v3) or through TenantSlot.root_plus --: the slot is read, nothing is searched
constexpr uint32_t KIND_DIRECT = 0x40000000u;
__device__ __forceinline__ uint32_t make_meta(uint32_t tl, uint32_t level, uint32_t kind) { return tl | (level << 6) | kind; }
__device__ __forceinline__ uint32_t meta_level(uint32_t m) { return (m >> 6) & 0xFFFFFFu; } // (bits 6-29)

struct StepOut {
    bool found;      // a node was discovered
    bool emit_own, emit_hash, push_l, push_h;
    uint32_t idx;    // its node id
    uint32_t dl;     // levels consumed on arrival
    uint32_t plus_slot; // push_h: the slot (relative to the region) of the node's '+' child if it lies beside the node, else NONE (hashed home)
    TrieSlot s;
};

// Both item kinds read ONE bucket line: the home bucket of edge (node, token) -- which holds the child's complete slot.
// ln = bucket `bk` of the item (already loaded).  tok = the edge label: the topic's token at `level`, or TOK_PLUS.
// tok_at(level): the topic's token at that level; nlev: level count; sys: first level starts with '$'
template <class TokAt>
__device__ __forceinline__ void resolve_item(const DistIndexView& ix, Line64 ln, bool live, uint32_t node,
                                             uint32_t tok, uint32_t bk, uint32_t rbase, uint32_t rbuckets, uint32_t level,
                                             uint32_t nlev, bool sys, TokAt&& tok_at, StepOut& o) {
    bool m0 = ln.a0.x == node && ln.a0.y == tok;
    bool m1 = ln.b0.x == node && ln.b0.y == tok;
    bool more = live && !m0 && !m1 && ln.a0.x != NONE && ln.b0.x != NONE;
    // home bucket full of other edges (rare at load factor 1/2): first-free probing continues.  Bounded by the region size so
    // that not even a damaged image can hang the GPU.
    for (uint32_t probes = 1; more && probes < rbuckets; probes++) {
        bk = (bk + 1 == rbuckets) ? 0 : bk + 1;
        load_line64(ix.trie + rbase + 2 * bk, ln);
        m0 = ln.a0.x == node && ln.a0.y == tok;
        m1 = ln.b0.x == node && ln.b0.y == tok;
        more = !m0 && !m1 && ln.a0.x != NONE && ln.b0.x != NONE;
    }
    o.found = live && (m0 || m1);
    o.s = m1 ? unpack_slot(ln.b0, ln.b1) : unpack_slot(ln.a0, ln.a1);
    o.idx = o.s.node; // the child's node id: the parent key of its own children
    o.dl = level + 1; // >= 1: the '$' rule (wildcards in filter position 0) is the caller's business at the root only
    o.emit_own = o.found && (o.dl == nlev) && o.s.own_count != 0;
    o.emit_hash = o.found && o.s.hash_count != 0; // "<path>/#" matches whatever follows, also nothing
    o.push_l = o.push_h = false;
    o.plus_slot = NONE;
    if (o.found && o.dl < nlev) {
        const uint32_t t = tok_at(o.dl);
        o.push_l = t != TOK_UNKNOWN && ((o.s.lit_bloom >> bloom_bit(t)) & 1u);
        o.push_h = (o.s.lit_bloom & BLOOM_PLUS) != 0;
        const uint4& other = m1 ? ln.a0 : ln.b0; // layout v3: the '+' child lies in the other slot of this line if that was free when it came into being
        if (o.push_h && other.x == o.s.node && other.y == TOK_PLUS) o.plus_slot = 2 * bk + (m1 ? 0u : 1u);
    }
    (void)sys;
}
// ... of an item that names the '+' child's slot itself (KIND_DIRECT)
template <class TokAt>
__device__ __forceinline__ void step_direct(const DistIndexView& ix, const TenantSlot& rg, uint32_t slot_rel, uint32_t level, uint32_t nlev, TokAt&& tok_at, StepOut& o) {
    Line64 ln;
    load_line64(ix.trie + (size_t)rg.base + (slot_rel & ~1u), ln);
    const bool odd = (slot_rel & 1u) != 0;
    const uint4& hd = odd ? ln.b0 : ln.a0;
    o.found = hd.x != NONE && hd.y == TOK_PLUS;
    o.s = odd ? unpack_slot(ln.b0, ln.b1) : unpack_slot(ln.a0, ln.a1);
    o.idx = o.s.node;
    o.dl = level + 1;
    o.emit_own = o.found && (o.dl == nlev) && o.s.own_count != 0;
    o.emit_hash = o.found && o.s.hash_count != 0;
    o.push_l = o.push_h = false;
    o.plus_slot = NONE;
    if (o.found && o.dl < nlev) {
        const uint32_t t = tok_at(o.dl);
        o.push_l = t != TOK_UNKNOWN && ((o.s.lit_bloom >> bloom_bit(t)) & 1u);
        o.push_h = (o.s.lit_bloom & BLOOM_PLUS) != 0;
        const uint4& other = odd ? ln.a0 : ln.b0; // (the node's parent if the node lies beside it; the root's '+' child may have its own '+' child here)
        if (o.push_h && other.x == o.s.node && other.y == TOK_PLUS) o.plus_slot = slot_rel ^ 1u;
    }
}

template <class TokAt>
__device__ __forceinline__ void step_item(const DistIndexView& ix, const TenantSlot& rg, uint32_t node, uint32_t level,
                                          bool kind_p, uint32_t nlev, bool sys, TokAt&& tok_at, StepOut& o) {
    const uint32_t tok = kind_p ? TOK_PLUS : tok_at(level);
    const uint32_t bk = edge_bucket(node, tok, rg.buckets);
    Line64 ln;
    load_line64(ix.trie + rg.base + 2 * (size_t)bk, ln);
    resolve_item(ix, ln, true, node, tok, bk, rg.base, rg.buckets, level, nlev, sys, tok_at, o);
}

// profiling experiments only (BMQ_DEBUG=2): a time stamp behind everything this wave has requested so far
#ifndef BMQ_WAVE_EMU
__device__ __forceinline__ unsigned long long dbg_clock(bool on) {
    if (!on) return 0ull;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    return __builtin_amdgcn_s_memtime();
}
#else
inline unsigned long long dbg_clock(bool) { return 0ull; }
#endif
} // namespace bmq
#ifndef BMQ_WAVE_EMU
#include "bmq_dedup_kernels.h" // k_dedup / k_fill: identical (tenant, topic) rows of a batch are walked once
#endif
#include "bmq_walk_kernel.h"   // k_walk<TC, QC, PC, MIXED>: one wave (= one 64-thread workgroup) per 64 topics
namespace bmq {

// ------------------------------------------------------------------------------------------------------------
// k_walk_slow -- per-lane depth-first walk, tokens + stack in global scratch, two passes (count, write)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_walk_slow(BatchArgs a) {
    const uint32_t n_slow = a.ctr->slow_count < a.slow_cap ? a.ctr->slow_count : a.slow_cap;
    const uint8_t* gbytes = a.topics;
    auto byte_at = [&](uint32_t i) -> uint32_t { return gbytes[i]; };
    auto word_at = [&](uint32_t i) -> uint32_t { return global_word_at(gbytes, i); };
    for (uint32_t i = blockIdx.x * 64 + threadIdx.x; i < n_slow; i += gridDim.x * 64) {
        const uint32_t t = a.slow_list[i];
        const uint32_t beg = a.topic_off[t], end = a.topic_off[t + 1];
        const uint32_t ti = a.topic_tenant[t];
        if (a.visit_cnt) a.visit_cnt[t] = 0;
        if (ti >= a.n_tenants) continue;
        const TenantSlot rg = resolve_tenant(a, ti);
        if (!tenant_known(rg)) continue;
        // level count first (cheap scan), then scratch: nlev tokens + stack of 2-word entries.  A DFS pop pushes at most
        // two items one level deeper: <= 1 pending sibling per level + 2.
        uint32_t nlev = 1;
        for (uint32_t j = beg; j < end; j++) nlev += byte_at(j) == '/';
        const unsigned long long need = (unsigned long long)nlev + 2ull * (2ull * nlev + 8);
        const unsigned long long so = atomicAdd(&a.ctr->scratch_alloc, need);
        if (so + need > a.scratch_cap) {
            atomicOr(&a.ctr->status, ST_NEED_SCRATCH);
            continue;
        }
        uint32_t* toks = a.scratch + so;
        uint32_t* stack = toks + nlev;
        {
            uint32_t pos = beg;
            for (uint32_t l = 0; l < nlev; l++) {
                LevelHash h;
                uint32_t inl[4], len;
                bool last;
                const uint32_t start = pos;
                scan_level(pos, end, true, word_at, h, inl, len, last);
                toks[l] = dict_lookup(a.ix, h, len, inl, start, byte_at);
            }
        }
        const bool sys = end > beg && byte_at(beg) == '$';
        unsigned long long base = 0;
        uint32_t np = 0, nr = 0, visits = 0;
        bool ok = true;
        for (int pass = 0; pass < 2 && ok; pass++) {
            uint32_t sp = 0, wp = 0;
            { // the tenant root: its payload is in the directory entry
                if (rg.root_hash_count != 0 && !sys) { // the filter "#"; never for '$' topics
                    if (pass == 0) {
                        np++;
                        nr += rg.root_hash_count & ~RANGE_INDIRECT;
                    } else a.pairs[base + wp++] = MatchRange{rg.root_hash_begin, rg.root_hash_count};
                }
                const uint32_t t0 = toks[0];
                if (t0 != TOK_UNKNOWN && ((rg.root_lit_bloom >> bloom_bit(t0)) & 1u)) {
                    stack[2 * sp] = 0;
                    stack[2 * sp + 1] = make_meta(0, 0, 0);
                    sp++;
                }
                if ((rg.root_lit_bloom & BLOOM_PLUS) && !sys) { // the root's '+' child: the directory entry knows its slot
                    stack[2 * sp] = rg.root_plus != NONE ? rg.root_plus : 0u;
                    stack[2 * sp + 1] = make_meta(0, 0, rg.root_plus != NONE ? (KIND_P | KIND_DIRECT) : KIND_P);
                    sp++;
                }
            }
            while (sp) {
                sp--;
                const uint32_t node = stack[2 * sp], meta = stack[2 * sp + 1];
                StepOut o;
                if (meta & KIND_DIRECT) step_direct(a.ix, rg, node, meta_level(meta), nlev, [&](uint32_t l) { return toks[l]; }, o);
                else step_item(a.ix, rg, node, meta_level(meta), (meta & KIND_P) != 0, nlev, sys, [&](uint32_t l) { return toks[l]; }, o);
                if (!o.found) continue;
                if (pass == 0) visits++;
                if (o.emit_own) {
                    if (pass == 0) {
                        np++;
                        nr += o.s.own_count & ~RANGE_INDIRECT;
                    } else a.pairs[base + wp++] = MatchRange{o.s.own_begin, o.s.own_count};
                }
                if (o.emit_hash) {
                    if (pass == 0) {
                        np++;
                        nr += o.s.hash_count & ~RANGE_INDIRECT;
                    } else a.pairs[base + wp++] = MatchRange{o.s.hash_begin, o.s.hash_count};
                }
                if (o.push_l) {
                    stack[2 * sp] = o.idx;
                    stack[2 * sp + 1] = make_meta(0, o.dl, 0);
                    sp++;
                }
                if (o.push_h) {
                    stack[2 * sp] = o.plus_slot != NONE ? o.plus_slot : o.idx;
                    stack[2 * sp + 1] = make_meta(0, o.dl, o.plus_slot != NONE ? (KIND_P | KIND_DIRECT) : KIND_P);
                    sp++;
                }
            }
            if (pass == 0) {
                if (np && !pair_alloc(a.subs, a.pair_cap, t >> a.tpw_shift, np, base)) {
                    atomicOr(&a.ctr->status, ST_NEED_PAIRS);
                    ok = false;
                }
            }
        }
        if (!ok) continue;
        a.pair_off[t] = (uint32_t)base;
        a.pair_cnt[t] = np;
        a.route_cnt[t] = nr;
        if (a.rep) { // de-duplication on: k_fill sums the blocks and the statistics, duplicates of this row included
            a.visit_cnt[t] = visits;
            continue;
        }
        if (nr) {
            atomicAdd(&a.wave_sums[t >> a.tpw_shift], (unsigned long long)nr);
            atomicAdd(&a.super_sums[(size_t)(t >> (a.tpw_shift + SUPER_SHIFT)) * SUPER_STRIDE], (unsigned long long)nr);
        }
        if (visits) atomicAdd(&a.ctr->n_visit, (unsigned long long)visits);
        if (np) atomicAdd(&a.ctr->n_ranges, (unsigned long long)np);
    }
}

} // namespace bmq
#include "bmq_expand_kernel.h" // k_expand -- CSR row pointers + ids
#include "bmq_dedup_adj_kernels.h" // k_dd_adj_heads / k_dd_adj_scatter / k_fill_adj: an ORDERED batch is reduced to its distinct rows by comparing neighbours
namespace bmq {

// ------------------------------------------------------------------------------------------------------------
// k_sort_rows -- one workgroup per flagged row, normalised bitonic network: in LDS for rows of up to SORT_LDS ids (one read and one write
// of the row in global memory; round 3 ran the whole network in global memory: 66 round trips for a 1000-id row, 157 us per C2 batch
// for a few hundred rows), in global memory beyond
// ------------------------------------------------------------------------------------------------------------
#ifndef BMQ_WAVE_EMU // (a 256-thread workgroup with __syncthreads: not a single-wave kernel)
constexpr uint32_t SORT_LDS = 4096;
__global__ __launch_bounds__(256) void k_sort_rows(BatchArgs a) {
    __shared__ uint32_t s_v[SORT_LDS];
    const uint32_t n_rows = a.ctr->sort_count < a.sort_cap ? a.ctr->sort_count : a.sort_cap;
    if (a.ctr->status & (ST_NOSPACE | ST_RANGE | ST_RERUN)) return;
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const uint32_t t = a.sort_list[r];
        const uint32_t lo = a.out_row_ptr[t], n = a.out_row_ptr[t + 1] - lo;
        uint32_t* g = a.out_ids + lo;
        uint32_t np2 = 1;
        while (np2 < n) np2 <<= 1;
        const bool in_lds = np2 <= SORT_LDS;
        uint32_t* v = in_lds ? s_v : g;
        if (in_lds) {
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_v[i] = g[i];
            __syncthreads();
        }
        for (uint32_t k = 2; k <= np2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = threadIdx.x; i < np2; i += blockDim.x) {
                    const uint32_t p = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j); // first stage mirrors
                    if (p > i && p < n) {
                        const uint32_t x = v[i], y = v[p];
                        if (x > y) {
                            v[i] = y;
                            v[p] = x;
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (in_lds) {
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) g[i] = s_v[i];
            __syncthreads();
        }
    }
}

#endif

} // namespace bmq
