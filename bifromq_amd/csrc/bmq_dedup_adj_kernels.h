// bmq_dedup_adj_kernels.h -- de-duplication of a batch that arrives ordered by (tenant, topic) (included by bmq_dist_kernels.h behind
// bmq_expand_kernel.h, whose cross-lane vocabulary it uses; tools/emu/dedup_adj_emu.cpp runs the same source on the host).
//
// BatchDistRequest carries its packs "sorted by tenantId and topic" and every topic once (bifromq-dist-coproc-proto/src/main/proto/
// distservice/DistWorkerCoProc.proto:75-83, built by BatchDistServerCall.java:138-152 from a TreeMap per tenant); a caller that hands the
// publishes over as they are -- ordered, repeats included -- gets them reduced to that shape here.  In an ordered batch equal rows are
// NEIGHBOURS: no hash table, no compare-and-swap (k_dedup spends 0.095 ms of a 1 M-publish batch on 10 k rows of one hot topic meeting
// in one table slot), and -- what the hash variant cannot do -- the representatives come out in batch order, so
// they are COPIED into a dense batch and the walk kernels run on that: 64 representatives per wave instead of 64 rows of which some are.
//   k_dd_adj_heads   : one wave per block of rows; a row is a HEAD unless it equals the row before it (tenant index, length, bytes);
//                      rep[row] = the nearest head at or before it inside the block, ADJ_PENDING where the run began in an earlier
//                      block; per block: heads | bytes of the heads' topics, the last head; the same per super-block (one atomic each).
//   k_dd_adj_scatter : one wave per block; heads and bytes in front of the block from <= 4 + n_super / 64 loads per lane (the scheme of
//                      k_expand's row pointers), the pending rows take the nearest head in front of the block; every head gets its dense
//                      row (tenant, offset, bytes: the block's heads are ONE contiguous piece of the dense batch, assembled in LDS and
//                      stored 16 bytes at a time); every other row marks one row behind the last head as "no such tenant".
//   k_fill_adj       : (k_fill's part) behind the walk: a row takes the ranges and counts of its head's dense row; per-block id counts
//                      and statistics, every row counted with its representative's figures.
// Nothing here depends on the batch BEING ordered: a row that equals no neighbour is its own head, whatever else the batch holds.
#pragma once

namespace bmq {

#ifndef BMQ_ADJ_IMG
#define BMQ_ADJ_IMG 4096
#endif
constexpr uint32_t ADJ_IMG = BMQ_ADJ_IMG; // bytes of LDS a k_dd_adj_scatter wave assembles its heads' topics in (a block of longer topics: byte copies)

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((unsigned long long)v, d));
    return v;
}
// rows a and b hold the same bytes (of the same length `len`)
__device__ __forceinline__ bool adj_same_bytes(const uint8_t* base, uint32_t pa, uint32_t pb, uint32_t len) {
    for (uint32_t k = 0; k < len; k += 16) { // four words per step: eight loads in flight
        uint32_t x = 0;
#pragma unroll
        for (uint32_t j = 0; j < 16; j += 4) {
            if (k + j >= len) break;
            uint32_t w = global_word_at(base, pa + k + j) ^ global_word_at(base, pb + k + j);
            if (len - (k + j) < 4) w &= (1u << (8u * (len - (k + j)))) - 1u;
            x |= w;
        }
        if (x) return false;
    }
    return true;
}

__global__ __launch_bounds__(64) void k_dd_adj_heads(AdjArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const uint32_t t_first = blk << a.tpw_shift;
    const uint32_t t = t_first + lane;
    const bool valid = lane < (1u << a.tpw_shift) && t < a.n_topics;
    uint32_t len = 0;
    bool head = false;
    if (valid) {
        const uint32_t pos = a.topic_off[t], end = a.topic_off[t + 1];
        len = end - pos;
        head = true;
        if (t > 0) {
            const uint32_t ppos = a.topic_off[t - 1];
            if (pos - ppos == len && a.topic_tenant[t - 1] == a.topic_tenant[t]) head = !adj_same_bytes(a.topics, pos, ppos, len);
        }
    }
    const unsigned long long heads = ballot64(head);
    const unsigned long long upto = heads & ((2ull << lane) - 1ull); // (lane 63: 2 << 63 wraps to 0, minus one = every lane)
    if (valid) a.rep[t] = upto ? t_first + 63u - (uint32_t)__builtin_clzll(upto) : ADJ_PENDING;
    const unsigned long long packed = wave_total_u64(head ? (1ull << 32) | len : 0ull); // (a batch's topic bytes are < 2^32: its offsets are 32 bits)
    if (lane == 0) {
        const unsigned long long last1 = heads ? (unsigned long long)t_first + 64u - (uint32_t)__builtin_clzll(heads) : 0ull;
        a.blk_cnt[blk] = packed;
        a.blk_last[blk] = (uint32_t)last1;
        if (heads) {
            unsigned long long* line = a.super_cnt + (size_t)(blk >> SUPER_SHIFT) * SUPER_STRIDE;
            atomicAdd(line, packed);
            atomicMax(line + 1, last1);
        }
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
    uint32_t near1 = 0; // 1 + the last head in front of the block
    for (uint32_t i = lane; i < n_super; i += 64) {
        const unsigned long long c = a.super_cnt[(size_t)i * SUPER_STRIDE], l = a.super_cnt[(size_t)i * SUPER_STRIDE + 1];
        all += c;
        if (i < sb) before += c, near1 = max(near1, (uint32_t)l);
    }
    static_assert(SUPER_SHIFT == 8, "four loads per lane cover a super-block");
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t i = w0 + lane + 64u * j, ii = min(i, blk);
        const unsigned long long c = a.blk_cnt[ii];
        const uint32_t l = a.blk_last[ii];
        if (i < blk) before += c, near1 = max(near1, l);
    }
    uint32_t rep = 0, pos = 0, len = 0, ti = 0;
    if (valid) {
        rep = a.rep[t];
        pos = a.topic_off[t];
        len = a.topic_off[t + 1] - pos;
        ti = a.topic_tenant[t];
    }
    before = wave_total_u64(before);
    all = wave_total_u64(all);
    near1 = wave_max_u32(near1);
    const uint32_t heads_before = (uint32_t)(before >> 32), bytes_before = (uint32_t)before;
    const uint32_t n_heads = (uint32_t)(all >> 32), bytes_all = (uint32_t)all;
    const bool head = valid && rep == t;
    if (valid && rep == ADJ_PENDING) a.rep[t] = near1 - 1u; // (row 0 is a head: a pending row has one in front of its block)
    // The dense batch's bytes do not fit the buffer: every wave sees the same totals and takes this turn together -- the dense rows are all
    // marked "no such tenant" with empty topics (the walk kernels touch nothing), the host grows the buffer and runs the batch again.
    const bool over = (unsigned long long)bytes_all + 64u > a.c_cap;
    const unsigned long long heads = ballot64(head);
    const uint32_t rk = rank_below(heads);
    const uint32_t hlen = head ? len : 0u;
    const uint32_t incl = wave_incl_scan(hlen);
    const uint32_t excl = incl - hlen, wbytes = read_lane(incl, 63);
    if (head) {
        const uint32_t d = heads_before + rk;
        a.dense[t] = d;
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
            for (uint32_t k = 0; k < hlen; k += 4) {
                const uint32_t w = global_word_at(a.topics, pos + k);
                uint8_t* o = img + lead + excl + k;
                o[0] = (uint8_t)w;
                if (k + 1 < hlen) o[1] = (uint8_t)(w >> 8);
                if (k + 2 < hlen) o[2] = (uint8_t)(w >> 16);
                if (k + 3 < hlen) o[3] = (uint8_t)(w >> 24);
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
    const uint32_t* rep;
    const uint32_t* dense;
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
        const uint32_t r = f.dense[f.rep[t]];
        np = f.c_pair_cnt[r];
        nr = f.c_route_cnt[r];
        vis = f.c_visit[r];
        bytes = a.topic_off[t + 1] - a.topic_off[t];
        a.pair_off[t] = f.c_pair_off[r];
        a.pair_cnt[t] = np;
        a.route_cnt[t] = nr;
    }
    const unsigned long long wsum = wave_total_u64(nr), wvis = wave_total_u64(vis), wnp = wave_total_u64(np), wbytes = wave_total_u64(bytes);
    if (lane == 0) {
        a.wave_sums[blk] = wsum;
        if (wsum) atomicAdd(&a.super_sums[(size_t)(blk >> SUPER_SHIFT) * SUPER_STRIDE], wsum);
        a.blk_stats[blk] = make_uint4((uint32_t)wvis, (uint32_t)wnp, (uint32_t)wbytes, 0u);
    }
}

} // namespace bmq
