// bmq_dedup_adj_kernels.h -- de-duplication of a batch that arrives ordered by (tenant, topic) (included by bmq_dist_kernels.h behind
// bmq_expand_kernel.h, whose cross-lane vocabulary it uses; tools/emu/dedup_adj_emu.cpp runs the same source on the host).
//
// BatchDistRequest carries its packs "sorted by tenantId and topic" and every topic once (bifromq-dist-coproc-proto/src/main/proto/
// distservice/DistWorkerCoProc.proto:75-83, built by BatchDistServerCall.java:138-152 from a TreeMap per tenant); a caller that hands the
// publishes over as they are -- ordered, repeats included -- gets them reduced to that shape here.  In an ordered batch equal rows are
// NEIGHBOURS: no hash table, no compare-and-swap (k_dedup spends 0.095 ms of a 1 M-publish batch on 10 k rows of one hot topic meeting
// in one table slot), and -- what the hash variant cannot do -- the representatives come out in batch order, so
// they are COPIED into a dense batch and the walk kernels run on that: 64 representatives per wave instead of 64 rows of which some are.
//   k_dd_adj_heads   : one wave per block of rows; the block's bytes are staged in LDS (coalesced 16-byte loads); a row is a HEAD unless it
//                      equals the row before it (tenant index, length, bytes); per block: the 64-bit mask of its heads, heads | bytes of
//                      the heads' topics; the latter summed per super-block (one atomic).
//   k_dd_adj_scatter : one wave per block; heads and bytes in front of the block from <= 4 + n_super / 64 loads per lane (the scheme of
//                      k_expand's row pointers); every row learns the DENSE ROW that answers for it (its own if it is a head, else the
//                      last head at or before it: in front of the block that is dense row heads_before - 1); every head gets its dense
//                      row (tenant, offset, bytes: the block's heads are ONE contiguous piece of the dense batch, assembled in LDS and
//                      stored 16 bytes at a time); every other row marks one row behind the last head as "no such tenant".
//   k_fill_adj       : (k_fill's part) behind the walk: a row takes the ranges and counts of its dense row; per-block id counts
//                      and statistics, every row counted with its representative's figures.
// First version (profiles/r05b/ordered_kernels_v1.txt): rows compared through dependent reads of global memory, row indices of the heads
// resolved by a second look-up: 26 + 28 + 14 us per 1 M rows.
// Nothing here depends on the batch BEING ordered: a row that equals no neighbour is its own head, whatever else the batch holds.
#pragma once

namespace bmq {

#ifndef BMQ_ADJ_IMG
#define BMQ_ADJ_IMG 4096
#endif
constexpr uint32_t ADJ_IMG = BMQ_ADJ_IMG; // bytes of LDS a wave stages a block's topics in (k_dd_adj_heads) / assembles its heads' topics in (k_dd_adj_scatter);
                                          // a block of longer topics: reads from global memory / byte copies
static_assert(ADJ_IMG % 16 == 0 && ADJ_IMG >= 64, "16-byte chunks");

#ifndef BMQ_WAVE_EMU
// 4 bytes at byte offset rel of an LDS image (any alignment; the image is readable 4 bytes past what is asked for)
__device__ __forceinline__ uint32_t lds_word_at(const uint32_t* words, uint32_t rel) {
    return __builtin_amdgcn_alignbyte(words[(rel >> 2) + 1], words[rel >> 2], rel & 3u);
}
#endif
// rows at pa and pb hold the same `len` bytes; word_at(i) = 4 bytes at byte offset i of the batch
template <class W> __device__ __forceinline__ bool adj_same_bytes(W word_at, uint32_t pa, uint32_t pb, uint32_t len) {
    for (uint32_t k = 0; k < len; k += 16) { // four words per step
        uint32_t x = 0;
#pragma unroll
        for (uint32_t j = 0; j < 16; j += 4) {
            if (k + j >= len) break;
            uint32_t w = word_at(pa + k + j) ^ word_at(pb + k + j);
            if (len - (k + j) < 4) w &= (1u << (8u * (len - (k + j)))) - 1u;
            x |= w;
        }
        if (x) return false;
    }
    return true;
}

__global__ __launch_bounds__(64) void k_dd_adj_heads(AdjArgs a) {
    __shared__ uint4 img16[ADJ_IMG / 16 + 1]; // (+ 16 bytes: a word read at the image's last byte reaches past it)
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const uint32_t tpw = 1u << a.tpw_shift;
    const uint32_t t_first = blk << a.tpw_shift, t_end = min(t_first + tpw, a.n_topics);
    const uint32_t t = t_first + lane;
    const bool valid = lane < tpw && t < a.n_topics;
    // the block's bytes and those of the row in front of it, staged with coalesced 16-byte loads: every row is compared with its
    // predecessor out of LDS (one trip to memory for the offsets, one for the bytes)
    const uint32_t s_beg = uniform_word(a.topic_off + (t_first ? t_first - 1u : 0u)), s_end = uniform_word(a.topic_off + t_end);
    const uint32_t a0 = s_beg & ~15u;
    const bool staged = (s_end - a0) <= ADJ_IMG;
    if (staged) {
        const uint4* src = reinterpret_cast<const uint4*>(a.topics + a0);
        const uint32_t n16 = (s_end - a0 + 15u) >> 4;
        for (uint32_t o = lane; o < n16; o += 64) img16[o] = src[o];
    }
    uint32_t pos = 0, len = 0, ppos = 0;
    bool cand = false; // same tenant and length as the row before: the bytes decide
    if (valid) {
        pos = a.topic_off[t];
        len = a.topic_off[t + 1] - pos;
        if (t > 0) {
            ppos = a.topic_off[t - 1];
            cand = pos - ppos == len && a.topic_tenant[t - 1] == a.topic_tenant[t];
        }
    }
    wave_sync();
    bool head = valid;
    if (cand) {
        const uint32_t* words = reinterpret_cast<const uint32_t*>(img16);
        const uint8_t* gbytes = a.topics;
        if (staged) head = !adj_same_bytes([&](uint32_t i) { return lds_word_at(words, i - a0); }, pos, ppos, len);
        else head = !adj_same_bytes([&](uint32_t i) { return global_word_at(gbytes, i); }, pos, ppos, len);
    }
    const unsigned long long heads = ballot64(head);
    const unsigned long long packed = wave_total_u64(head ? (1ull << 32) | len : 0ull); // (a batch's topic bytes are < 2^32: its offsets are 32 bits)
    if (lane == 0) {
        a.blk_mask[blk] = heads;
        a.blk_cnt[blk] = packed;
        if (heads) atomicAdd(a.super_cnt + (size_t)(blk >> SUPER_SHIFT) * SUPER_STRIDE, packed);
    }
}

__global__ __launch_bounds__(64) void k_dd_adj_scatter(AdjArgs a) {
    __shared__ uint4 img16[ADJ_IMG / 16]; // (16-byte aligned by its type)
    uint8_t* const img = reinterpret_cast<uint8_t*>(img16);
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const uint32_t t_first = blk << a.tpw_shift;
    const uint32_t t = t_first + lane;
    const bool valid = lane < (1u << a.tpw_shift) && t < a.n_topics;
    // what lies in front of this block: whole super-blocks + the blocks of its own super-block before it -- and the batch's totals
    const uint32_t sb = blk >> SUPER_SHIFT, w0 = sb << SUPER_SHIFT, n_super = ((a.n_blocks - 1u) >> SUPER_SHIFT) + 1u;
    unsigned long long before = 0, all = 0;
    for (uint32_t i = lane; i < n_super; i += 64) {
        const unsigned long long c = a.super_cnt[(size_t)i * SUPER_STRIDE];
        all += c;
        if (i < sb) before += c;
    }
    static_assert(SUPER_SHIFT == 8, "four loads per lane cover a super-block");
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t i = w0 + lane + 64u * j;
        const unsigned long long c = a.blk_cnt[min(i, blk)];
        if (i < blk) before += c;
    }
    const unsigned long long heads = ((unsigned long long)uniform_word(reinterpret_cast<const uint32_t*>(a.blk_mask + blk) + 1) << 32) |
                                     uniform_word(reinterpret_cast<const uint32_t*>(a.blk_mask + blk));
    uint32_t pos = 0, len = 0, ti = 0;
    if (valid) {
        pos = a.topic_off[t];
        len = a.topic_off[t + 1] - pos;
        ti = a.topic_tenant[t];
    }
    const bool head = lane_bit(heads) != 0;
    // a head's first words, requested before anything is waited for (a topic of the survey's workloads is 37 bytes: one trip)
    constexpr uint32_t PRE = 12; // words held in registers
    uint32_t pre[PRE];
#pragma unroll
    for (uint32_t k = 0; k < PRE; k++) pre[k] = (head && 4u * k < len) ? global_word_at(a.topics, pos + 4u * k) : 0u;
    before = wave_total_u64(before);
    all = wave_total_u64(all);
    const uint32_t heads_before = (uint32_t)(before >> 32), bytes_before = (uint32_t)before;
    const uint32_t n_heads = (uint32_t)(all >> 32), bytes_all = (uint32_t)all;
    // The dense batch's bytes do not fit the buffer: every wave sees the same totals and takes this turn together -- the dense rows are all
    // marked "no such tenant" with empty topics (the walk kernels touch nothing), the host grows the buffer and runs the batch again.
    const bool over = (unsigned long long)bytes_all + 64u > a.c_cap;
    const uint32_t rk = rank_below(heads);
    const uint32_t hlen = head ? len : 0u;
    const uint32_t incl = wave_incl_scan(hlen);
    const uint32_t excl = incl - hlen, wbytes = read_lane(incl, 63);
    // the dense row that answers for this row: its own if it is a head, else the last head at or before it -- in front of the block
    // that is dense row heads_before - 1 (row 0 is a head: there is one)
    if (valid) a.drow[t] = heads_before + rk + (head ? 1u : 0u) - 1u;
    if (head) {
        const uint32_t d = heads_before + rk;
        a.c_tenant[d] = over ? 0xFFFFFFFFu : ti;
        a.c_off[d] = over ? 0u : bytes_before + excl;
        a.c_rep[d] = d;
    } else if (valid) { // one row behind the last head per row that is not one: never walked, an empty topic
        const uint32_t r = n_heads + (t - heads_before - rk);
        a.c_tenant[r] = 0xFFFFFFFFu;
        a.c_off[r + 1] = over ? 0u : bytes_all;
        a.c_rep[r] = r;
    }
    if (blk == a.n_blocks - 1 && lane == 0) {
        a.c_off[n_heads] = over ? 0u : bytes_all;
        a.ctr->n_walked = n_heads;
        if (over) {
            a.ctr->adj_bytes = bytes_all;
            atomicOr(&a.ctr->status, (uint32_t)ST_NEED_ADJ);
        }
    }
    if (wbytes == 0 || over) return;
    // the heads' bytes: dense bytes [bytes_before, bytes_before + wbytes) are this block's
    const uint32_t lead = bytes_before & 15u, total = lead + wbytes;
    uint8_t* const dst0 = a.c_topics + (bytes_before - lead); // 16-byte aligned; image byte i <-> dst0[i]
    if (total <= ADJ_IMG) {
        if (head) {
            uint8_t* o = img + lead + excl;
#pragma unroll
            for (uint32_t k = 0; k < PRE; k++) {
                if (4u * k < hlen) {
                    const uint32_t w = pre[k];
                    o[4u * k] = (uint8_t)w;
                    if (4u * k + 1 < hlen) o[4u * k + 1] = (uint8_t)(w >> 8);
                    if (4u * k + 2 < hlen) o[4u * k + 2] = (uint8_t)(w >> 16);
                    if (4u * k + 3 < hlen) o[4u * k + 3] = (uint8_t)(w >> 24);
                }
            }
            for (uint32_t k = 4u * PRE; k < hlen; k += 4) { // (longer topics)
                const uint32_t w = global_word_at(a.topics, pos + k);
                o[k] = (uint8_t)w;
                if (k + 1 < hlen) o[k + 1] = (uint8_t)(w >> 8);
                if (k + 2 < hlen) o[k + 2] = (uint8_t)(w >> 16);
                if (k + 3 < hlen) o[k + 3] = (uint8_t)(w >> 24);
            }
        }
        wave_sync();
        for (uint32_t c = lane * 16u; c < total; c += 1024u) {
            if (c >= lead && c + 16u <= total) { // a chunk that is all this block's: one 16-byte store
                *reinterpret_cast<uint4*>(dst0 + c) = *reinterpret_cast<const uint4*>(img + c);
            } else { // the first / last chunk is shared with the neighbouring blocks: bytes
                for (uint32_t b = max(c, lead); b < min(c + 16u, total); b++) dst0[b] = img[b];
            }
        }
    } else if (head) { // (a block of very long topics)
        for (uint32_t k = 0; k < hlen; k++) dst0[lead + excl + k] = a.topics[pos + k];
    }
}

struct AdjFill {
    const uint32_t* drow;
    const uint32_t* c_pair_off;
    const uint32_t* c_pair_cnt;
    const uint32_t* c_route_cnt;
    const uint32_t* c_visit;
};
// `a` = the batch as the caller handed it over (k_expand runs on it next)
__global__ __launch_bounds__(64) void k_fill_adj(BatchArgs a, AdjFill f) {
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const uint32_t t = (blk << a.tpw_shift) + lane;
    const bool valid = lane < (1u << a.tpw_shift) && t < a.n_topics;
    uint32_t nr = 0, np = 0, vis = 0, bytes = 0;
    if (valid) {
        const uint32_t r = f.drow[t];
        bytes = a.topic_off[t + 1] - a.topic_off[t];
        np = f.c_pair_cnt[r];
        nr = f.c_route_cnt[r];
        vis = f.c_visit[r];
        a.pair_off[t] = f.c_pair_off[r];
        a.pair_cnt[t] = np;
        a.route_cnt[t] = nr;
    }
    const unsigned long long wsum = wave_total_u64(nr), wvis = wave_total_u64(vis), wnp = wave_total_u64(np), wbytes = wave_total_u64(bytes);
    if (lane == 0) {
        a.wave_sums[blk] = wsum;
        if (wsum) atomicAdd(&a.super_sums[(size_t)(blk >> SUPER_SHIFT) * SUPER_STRIDE], wsum);
        a.blk_stats[blk] = make_uint4((uint32_t)wvis, (uint32_t)wnp, (uint32_t)wbytes, heavy_mark(a, blk, wnp, wsum));
    }
}

} // namespace bmq
