// bmq_fanout_kernels.h -- gfx950 kernels of the fan-out grouping's fast path (bmq_fanout.h): a COUNTING SORT that carries its payload.
//
// The (topic, route) pairs of a match batch are born in (topic, route) order -- they ARE the CSR -- and the sort key (the group of
// the route's DelivererKey, bmq_fanout_core.h) has a few dozen to a few hundred values.  So instead of a radix sort of (key, position)
// pairs followed by a gather of the route ids and a binary search of row_ptr per pair (round 2: nine launches, 4x the algorithmic
// HBM traffic), the pairs are split in three launches:
//   k_fo_hist     every wave owns a TILE of consecutive pairs: one gather per pair (dgroup[id] -> dense group number) gives the key,
//                 kept as 16 bits for the last pass; the wave's histogram goes out transposed (hist[key][tile]), so that
//   (scan)        ONE exclusive prefix sum over hist[] yields, in final order, where every (key, tile) run starts;
//   k_fo_scatter  the same waves walk their tiles again, 64 pairs at a time in order: the topic of a pair comes from a window of 64
//                 row ends (one coalesced load, a 6-step search by shuffle); its rank among the segment's pairs of the same key from
//                 one ballot per key BIT (the lanes whose key equals mine = AND over the bits of ballot or its complement) -- no loop
//                 over the distinct keys, and stable, so a group's pairs stay in (topic, route) order; the key's running offset
//                 lives in LDS, bumped once per segment by the key's first lane;
//   k_fo_groups2  one lane per key: the non-empty keys, in key order, are the groups.
// Group numbers are DENSE (k_fo_dense: rank of a group-table slot among the used slots; the two special groups follow), so the
// histogram has as many columns as there are deliverer keys, not as the table has slots.
// Ids that have no group slot yet (first batch after a rebuild, routes added since) are counted by k_fo_hist; the control then
// runs the mapping passes of bmq_fanout_core.h (fo_fill / fo_verify) once and starts over.
#pragma once
#include <hip/hip_runtime.h>

#include "bmq_dist_kernels.h" // rank_below, wave_sync
#include "bmq_fanout_core.h"

namespace bmq {

// dense[s] = number of used slots in front of slot s (one workgroup; the table has at most 2^29 slots but in practice a few thousand)
__global__ __launch_bounds__(1024) void k_fo_dense(const unsigned long long* gt_hash, uint32_t gt_cap, uint16_t* dense, uint32_t* n_used) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (gt_cap + 1023) / 1024, lo = min(gt_cap, tid * per), hi = min(gt_cap, lo + per);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += gt_hash[i] != 0ull ? 1u : 0u;
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (uint32_t i = lo; i < hi; i++) {
        dense[i] = (uint16_t)min(run, 0xFFFFu);
        run += gt_hash[i] != 0ull ? 1u : 0u;
    }
    if (tid == 1023) *n_used = part[1023];
}

__global__ __launch_bounds__(FO_WAVES * 64) void k_fo_hist(DistIndexMut ix, FanoutState st, FanoutFast f) {
    __shared__ uint32_t cnt_all[FO_WAVES][FO_MAX_BINS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * FO_WAVES + wave;
    if (tile >= f.n_tiles) return;
    uint32_t* cnt = cnt_all[wave];
    for (uint32_t b = lane; b < f.n_bins; b += 64) cnt[b] = 0;
    wave_sync();
    const uint32_t p0 = tile * f.tile, p1 = min(f.total, p0 + f.tile);
    uint32_t unset = 0;
    for (uint32_t p = p0 + lane; p < p1; p += 64) {
        const uint32_t id = f.ids[p];
        uint32_t key = f.n_bins - 1; // dead: never handed out, or deleted since the match
        if (id < f.id_end && id < st.id_cap) {
            // one gather per pair: a deleted route carries FO_DEAD_ID here (written by the lane that deleted it, bmq_build_core.h group_one)
            const uint32_t g = st.dgroup[id];
            if (g == FO_DEAD_ID) {
            } else if (g == FO_UNSET || (g & FO_NEW)) unset++;
            else key = g == st.gt_cap ? f.n_bins - 2 : (uint32_t)f.dense[g];
        }
        f.key16[p] = (uint16_t)key;
        atomicAdd(&cnt[key], 1u);
    }
    wave_sync();
    for (uint32_t b = lane; b < f.n_bins; b += 64) f.hist[(size_t)b * f.n_tiles + tile] = cnt[b];
    if (tile == 0 && lane == 0) f.hist[(size_t)f.n_bins * f.n_tiles] = 0; // scanned along: becomes the total, the end of the last run
    if (__any(unset != 0)) {
        uint32_t s = unset;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) atomicAdd(f.need_fill, s);
    }
}

// LDS of one wave of k_fo_scatter: delta[nb] | loff[nb] | topic[tile] | route[tile] | key[tile] (16 bits)
__host__ __device__ inline uint32_t fo_scatter_bins(uint32_t n_bins) { return (n_bins + 63u) & ~63u; }
__host__ __device__ inline uint32_t fo_scatter_lds(uint32_t n_bins, uint32_t tile) { return fo_scatter_bins(n_bins) * 8u + tile * 10u; }

__global__ __launch_bounds__(FO_SC_WAVES * 64) void k_fo_scatter(FanoutFast f) {
    extern __shared__ __align__(16) unsigned char fo_lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * FO_SC_WAVES + wave;
    if (tile >= f.n_tiles) return;
    const uint32_t nb = fo_scatter_bins(f.n_bins);
    uint32_t* delta = reinterpret_cast<uint32_t*>(fo_lds + (size_t)wave * fo_scatter_lds(f.n_bins, f.tile));
    uint32_t* loff = delta + nb;
    uint32_t* s_topic = loff + nb;
    uint32_t* s_route = s_topic + f.tile;
    uint16_t* s_key = reinterpret_cast<uint16_t*>(s_route + f.tile);
    const uint32_t p0 = tile * f.tile, p1 = min(f.total, p0 + f.tile);
    // ---- where the tile's run of every key starts: in the output (scanned histogram; the next entry of the flat array is the end of
    // the run, hist[n_bins * n_tiles] = total) and inside the tile (prefix sum of the run lengths over the keys)
    {
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
            const uint32_t b = b0 + lane;
            uint32_t g = 0, c = 0;
            if (b < f.n_bins) {
                const size_t at = (size_t)b * f.n_tiles + tile;
                g = f.hist[at];
                c = f.hist[at + 1] - g;
            }
            uint32_t inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(inc, d);
                if ((int)lane >= d) inc += v;
            }
            const uint32_t l = carry + inc - c;
            loff[b] = l;
            delta[b] = g - l;
            carry += __shfl(inc, 63);
        }
    }
    uint32_t r; // the row of the tile's first pair: the last r with row_ptr[r] <= p0 (rows may be empty)
    {
        uint32_t lo = 0, hi = f.n_topics;
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (f.row_ptr[mid] <= p0) lo = mid;
            else hi = mid;
        }
        r = lo;
    }
    wave_sync();
    const unsigned long long below = (1ull << lane) - 1ull;
    // The window of row ends: e = row_ptr[wbase + 1 + lane] = end of row wbase + lane, kept in registers ACROSS segments -- a segment of
    // 64 pairs spans only a few rows, so the window moves on (one dependent load) every few segments, not in every one; row_ptr[wbase] <= p
    // holds for every pair still to come.
    uint32_t wbase = r;
    auto load_window = [&](uint32_t base) {
        const uint32_t idx = base + 1 + lane;
        return idx <= f.n_topics ? f.row_ptr[idx] : 0xFFFFFFFFu; // (behind the last row: never reached)
    };
    uint32_t e = load_window(wbase);
    uint32_t id_next = p0 + lane < p1 ? f.ids[p0 + lane] : 0u;
    uint32_t key_next = p0 + lane < p1 ? (uint32_t)f.key16[p0 + lane] : 0u;
    for (uint32_t s = p0; s < p1; s += 64) {
        const uint32_t p = s + lane;
        const bool in = p < p1;
        const uint32_t id = id_next, key = key_next;
        {   // the next segment's pair is on its way while this one is ranked
            const uint32_t pn = p + 64;
            id_next = pn < p1 ? f.ids[pn] : 0u;
            key_next = pn < p1 ? (uint32_t)f.key16[pn] : 0u;
        }
        // ---- topic of every pair: the first row of the window that ends behind the pair (6-step search by shuffle); a lane whose pair
        // lies behind the whole window makes the wave move the window on
        uint32_t topic = wbase;
        bool placed = !in;
        for (;;) {
            uint32_t lo = 0, hi = 63; // first j with e_j > p, if e_63 > p
#pragma unroll
            for (int step = 0; step < 6; step++) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t em = __shfl(e, (int)mid);
                if (em <= p) lo = mid + 1;
                else hi = mid;
            }
            const uint32_t e_last = __shfl(e, 63);
            if (!placed && e_last > p) {
                topic = wbase + min(lo, 63u);
                placed = true;
            }
            if (__all(placed)) break;
            wbase += 64; // the pairs not placed yet start at or behind row_ptr[wbase + 64]
            e = load_window(wbase);
        }
        // ---- stable rank among the segment's pairs of the same key: `same` = lanes whose key equals mine
        const unsigned long long m_in = __ballot(in);
        unsigned long long same = m_in;
        for (uint32_t b = 0; b < f.key_bits; b++) {
            const unsigned long long mb = __ballot((key >> b) & 1u);
            same &= ((key >> b) & 1u) ? mb : ~mb;
        }
        const uint32_t rank = (uint32_t)__popcll(same & below), cnt = (uint32_t)__popcll(same);
        const int leader = __ffsll((long long)same) - 1;
        uint32_t base_off = 0;
        if (in && (int)lane == leader) { // one lane per distinct key: no two leaders share a counter
            base_off = loff[key];
            loff[key] = base_off + cnt;
        }
        base_off = __shfl(base_off, in ? leader : 0);
        if (in) { // into the tile's LDS copy, sorted by key: the runs leave for HBM in whole pieces below
            const uint32_t at = base_off + rank;
            s_topic[at] = topic;
            s_route[at] = id;
            s_key[at] = (uint16_t)key;
        }
        wave_sync();
    }
    // ---- the tile, now ordered by (key, topic, route): position j of the tile goes to delta[key] + j -- neighbouring lanes write
    // neighbouring words of a run (the scattered 4-byte stores of a direct scatter cost 3x the time: partial lines thrash L2)
    const uint32_t n = p1 - p0;
    for (uint32_t j = lane; j < n; j += 64) {
        const uint32_t dst = delta[s_key[j]] + j;
        f.out_topic[dst] = s_topic[j];
        f.out_route[dst] = s_route[j];
    }
}

// one lane per key: group g = the g-th non-empty key; its pairs start where the key's first tile run starts
__global__ __launch_bounds__(1024) void k_fo_groups2(FanoutState st, FanoutFast f) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (f.n_bins + 1023) / 1024, lo = min(f.n_bins, tid * per), hi = min(f.n_bins, lo + per);
    auto start_of = [&](uint32_t b) { return f.hist[(size_t)b * f.n_tiles]; };
    auto end_of = [&](uint32_t b) { return b + 1 < f.n_bins ? f.hist[(size_t)(b + 1) * f.n_tiles] : f.total; };
    uint32_t s = 0;
    for (uint32_t b = lo; b < hi; b++) s += end_of(b) > start_of(b) ? 1u : 0u;
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t g = part[tid] - s;
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t b0 = start_of(b), b1 = end_of(b);
        if (b1 == b0) continue;
        if (g < f.group_cap) {
            f.group_off[g] = b0;
            // a normal group is named by one of ITS routes of this batch (alive as of the match; the slot's first route may be gone by now)
            f.group_rep[g] = b + 2 < f.n_bins ? f.out_route[b0] : (b + 2 == f.n_bins ? 0xFFFFFFFEu : 0xFFFFFFFFu);
        }
        g++;
    }
    if (tid == 1023) {
        const uint32_t n = part[1023];
        st.flags[2] = n;
        if (n <= f.group_cap) f.group_off[n] = f.total;
        // which special groups are present (they sort last): bit 0 shared subscriptions, bit 1 dead ids
        const uint32_t sh0 = start_of(f.n_bins - 2), d0 = start_of(f.n_bins - 1);
        st.flags[3] = (d0 > sh0 ? 1u : 0u) | (f.total > d0 ? 2u : 0u);
    }
}

} // namespace bmq
