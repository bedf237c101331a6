// bmq_dedup_kernels.h -- in-batch de-duplication of identical (tenant, topic) rows (included by bmq_dist_kernels.h).
//
// TenantRouteMatcher.matchAll takes a SET of topics (DW/cache/TenantRouteMatcher.java:67-78): a batch of publishes holds the same topic
// many times (Zipf publishes -- 405 k distinct of 726 k in the sample the CPU baseline runs), and the walk is bound by the
// rate of random 64-byte lines the memory system delivers (profiles/r04/k_walk_experiments.md), so walking a topic once per batch is the
// one lever that cuts its line requests.  Every input row keeps its own row in the CSR.
//   k_dedup : one lane per topic; 64-bit hash of (tenant index, topic bytes) -> open-addressing table of {generation, hash tag, row}
//             entries in HBM; the first row of a (tenant, topic) claims the slot by compare-and-swap and is the REPRESENTATIVE, every
//             later row finds it and compares its bytes with the representative's (exact: a hash decides nothing) -> rep[row].
//             The table is never cleared between batches: entries of other generations count as free (8-bit generation; the host
//             zeroes the table when it wraps).
//   k_walk  : walks the representatives only (bmq_walk_kernel.h); counts the nodes it discovers per topic.
//   k_fill  : one lane per row; a duplicate row takes its representative's (range list, id count); per 64-row block the id count
//             (k_expand derives the row pointers from them) and the statistics, every row counted with its representative's figures
//             -- N_visit stays the property of the DATA the roofline accounting takes it for.
#pragma once

namespace bmq {

constexpr uint32_t DEDUP_STAGE = 4096; // bytes of LDS a k_dedup wave stages its topics in

__device__ __forceinline__ uint64_t dd_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t dd_cas(unsigned long long* p, unsigned long long expect, unsigned long long desired) {
    __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect; // the value that was there
}

__global__ __launch_bounds__(64) void k_dedup(BatchArgs a) {
    __shared__ __align__(16) uint32_t stage[DEDUP_STAGE / 4];
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const uint32_t tpw = 1u << a.tpw_shift;
    const uint32_t t = (blk << a.tpw_shift) + lane;
    const bool valid = lane < tpw && t < a.n_topics;
    const uint32_t t_first = blk << a.tpw_shift, t_end = min(t_first + tpw, a.n_topics);
    const uint32_t s_beg = scalar_words(a.topic_off)[t_first], s_end = scalar_words(a.topic_off)[t_end];
    const uint32_t a0 = s_beg & ~15u;
    const bool staged = (s_end - a0) + 32u <= DEDUP_STAGE;
    if (staged) {
        uint4* dst = reinterpret_cast<uint4*>(stage);
        const uint4* src = reinterpret_cast<const uint4*>(a.topics + a0);
        const uint32_t n16 = (s_end - a0 + 15) >> 4;
        for (uint32_t o = lane; o < n16; o += 64) dst[o] = src[o];
    }
    wave_sync();
    if (!valid) return;
    const uint32_t pos = a.topic_off[t], end = a.topic_off[t + 1], ti = a.topic_tenant[t];
    if (ti >= a.n_tenants) { // no such tenant in the batch's table: an empty row of its own
        a.rep[t] = t;
        return;
    }
    // hash of (tenant index, length, bytes): the ALIGNED words the topic lies in, bytes outside it masked away and the words shifted to the
    // topic's own alignment (two topics with the same bytes hash alike wherever they lie)
    const uint8_t* gbytes = a.topics;
    auto word_at = [&](uint32_t i) -> uint32_t { // 4 bytes at byte offset i (any alignment); bytes beyond `end` may be garbage: masked by the caller
        if (!staged) return global_word_at(gbytes, i);
        const uint32_t rel = i - a0;
        return __builtin_amdgcn_alignbyte(stage[(rel >> 2) + 1], stage[rel >> 2], rel & 3u);
    };
    const uint32_t len = end - pos;
    LevelHash h{0x811C9DC5u ^ (ti * 0x9E3779B1u), 0x9747B28Cu + len};
    for (uint32_t k = 0; k < len; k += 4) {
        uint32_t w = word_at(pos + k);
        if (len - k < 4) w &= (1u << (8u * (len - k))) - 1u;
        level_hash_word(h, w);
    }
    const uint32_t hi = mix32(h.h1 ^ rotl32(h.h2, 13)), lo = mix32(h.h2 + 0x7F4A7C15u * h.h1);
    const uint32_t tag = hi & 0xFFFFFFu;
    const unsigned long long mine = ((unsigned long long)a.dd_gen << 56) | ((unsigned long long)tag << 32) | t;
    uint32_t slot = lo & a.dd_mask;
    uint32_t rep = t;
    for (uint32_t steps = 0; steps <= 2u * a.dd_mask + 2u; steps++) { // bounded: a damaged table must not hang the GPU (then: no de-duplication for this row)
        const unsigned long long e = dd_load(a.dd_table + slot);
        if ((uint32_t)(e >> 56) != a.dd_gen) { // free: empty, or left by another batch
            if (dd_cas(a.dd_table + slot, e, mine) == e) break; // claimed: this row is the representative of its (tenant, topic)
            continue;                                           // somebody else took the slot meanwhile: look at it again
        }
        if (((uint32_t)(e >> 32) & 0xFFFFFFu) == tag) {
            const uint32_t r = (uint32_t)e;
            const uint32_t rpos = a.topic_off[r], rlen = a.topic_off[r + 1] - rpos;
            bool same = rlen == len && a.topic_tenant[r] == ti;
            for (uint32_t k = 0; k < len && same; k += 4) {
                uint32_t x = word_at(pos + k) ^ global_word_at(gbytes, rpos + k);
                if (len - k < 4) x &= (1u << (8u * (len - k))) - 1u;
                same = x == 0;
            }
            if (same) {
                rep = r;
                break;
            }
        }
        slot = (slot + 1) & a.dd_mask;
    }
    a.rep[t] = rep;
}

// k_fill: behind k_walk (and k_walk_slow): rows of duplicates take their representatives' results; per-block id counts and statistics.
__global__ __launch_bounds__(64) void k_fill(BatchArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const uint32_t tpw = 1u << a.tpw_shift;
    const uint32_t t = (blk << a.tpw_shift) + lane;
    const bool valid = lane < tpw && t < a.n_topics;
    uint32_t nr = 0, np = 0, vis = 0, bytes = 0;
    if (valid) {
        const uint32_t r = a.rep[t];
        np = a.pair_cnt[r];
        nr = a.route_cnt[r];
        vis = a.visit_cnt[r];
        bytes = a.topic_off[t + 1] - a.topic_off[t];
        if (r != t) {
            a.pair_off[t] = a.pair_off[r];
            a.pair_cnt[t] = np;
            a.route_cnt[t] = nr;
        }
    }
    const unsigned long long wsum = wave_sum_u64(nr), wvis = wave_sum_u64(vis), wnp = wave_sum_u64(np), wbytes = wave_sum_u64(bytes);
    if (lane == 0) {
        a.wave_sums[blk] = wsum;
        if (wsum) atomicAdd(&a.super_sums[(size_t)(blk >> SUPER_SHIFT) * SUPER_STRIDE], wsum);
        a.blk_stats[blk] = make_uint4((uint32_t)wvis, (uint32_t)wnp, (uint32_t)wbytes, heavy_mark(a, blk, wnp, wsum));
    }
}

} // namespace bmq
