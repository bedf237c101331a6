// the round-2/3 k_expand (kept for the emulator cross-check of tools/emu/expand_emu.cpp only)
#pragma once
namespace bmq {
// ------------------------------------------------------------------------------------------------------------
// k_expand -- CSR row pointers + ids.  One wave per 64 topics (same blocking as k_walk).
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t SORT_PAIRS = 32;  // range lists up to this length are ordered in place (insertion sort)
#ifndef BMQ_EXP_K
#define BMQ_EXP_K 320 // measured (round 2, 4 workgroups per CU by VGPRs): 256 -> C3 0.099 / C2 1.128 / C4 0.689 ms, 320 -> 0.098 / 1.045 / 0.687, 384 -> 0.100 / 1.083 / 0.700 (LDS then allows 3 workgroups)
#endif
#ifndef BMQ_EXP_WAVES
#define BMQ_EXP_WAVES 4
#endif
constexpr uint32_t EXP_K = BMQ_EXP_K;         // ranges laid out per LDS pass
constexpr uint32_t EXP_WAVES = BMQ_EXP_WAVES; // independent waves per k_expand workgroup
constexpr uint32_t EXP_LONG = 64;    // ranges at least this long are streamed, shorter ones are flattened

__device__ __forceinline__ uint32_t range_first_id(const DistIndexView& ix, const MatchRange& r) {
    return (r.count & RANGE_INDIRECT) ? ix.route_pos[r.begin] : r.begin;
}

// The 64 rows of a wave are one contiguous piece of the output.  Their ranges are laid out in LDS in output order (whole rows per
// pass) with the exclusive prefix of their lengths.  Long ranges (>= EXP_LONG ids) are streamed by the whole wave.  The short ones
// are flattened: a bitmap marks the element at which every short range starts, so the range that covers element u is
// popcount(bitmap[0..u]) - 1 -- one broadcast LDS read per 64 elements instead of a binary search per element (measured with
// BMQ_DEBUG=4 on C3: generation 28 k of 54 k clocks per wave with the search).  Stores are coalesced and all lanes stay busy
// whatever the mix of range lengths (a 5000-subscriber filter next to 60 singletons).
constexpr uint32_t EXP_FLAG_WORDS = (EXP_K * (EXP_LONG - 1) + 63) / 64 + 2;
constexpr uint32_t EXP_EPL = EXP_K / 64; // entries per lane in the prefix step
static_assert(EXP_K % 64 == 0 && EXP_K * (EXP_LONG - 1) < 65536, "short-range space: offsets are packed into 16 bits below");

// compare-exchange of (key, begin, count) triples held in registers
__device__ __forceinline__ void cex(uint32_t& ka, uint32_t& ba, uint32_t& ca, uint32_t& kb, uint32_t& bb, uint32_t& cb) {
    const bool sw = kb < ka;
    const uint32_t k0 = sw ? kb : ka, k1 = sw ? ka : kb, b0 = sw ? bb : ba, b1 = sw ? ba : bb, c0 = sw ? cb : ca, c1 = sw ? ca : cb;
    ka = k0, kb = k1, ba = b0, bb = b1, ca = c0, cb = c1;
}

#ifndef BMQ_EXP_MIN_WAVES
#define BMQ_EXP_MIN_WAVES 4
#endif
#ifndef BMQ_EXP_PREFETCH
#define BMQ_EXP_PREFETCH 1
#endif
__global__ __launch_bounds__(EXP_WAVES * 64, BMQ_EXP_MIN_WAVES) void k_expand(BatchArgs a) {
    __shared__ uint32_t s_begin[EXP_WAVES][EXP_K], s_cnt[EXP_WAVES][EXP_K], s_off[EXP_WAVES][EXP_K + 8], s_delta[EXP_WAVES][EXP_K];
    __shared__ unsigned long long s_flag[EXP_WAVES][EXP_FLAG_WORDS];
    __shared__ uint32_t s_ind[EXP_WAVES][EXP_K / 32];
    __shared__ uint32_t s_bad[EXP_WAVES][64];
    __shared__ unsigned long long s_rs[EXP_WAVES][2][EXP_EPL];
    __shared__ uint32_t s_lpo[EXP_WAVES][64], s_lpx[EXP_WAVES][64];
    __shared__ uint8_t s_nz[EXP_WAVES][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t blk = blockIdx.x * EXP_WAVES + wave; // every wave owns one 64-row block and its own LDS slice
    if (blk >= a.n_blocks) return;
    uint32_t* r_begin = s_begin[wave];
    uint32_t* r_cnt = s_cnt[wave];
    uint32_t* r_off = s_off[wave];
    uint32_t* c_delta = s_delta[wave];       // per SHORT range, in order: first id (or route_pos index) - its start in the short space
    unsigned long long* flag = s_flag[wave]; // bit u: a short range starts at element u of the pass's short-range space
    uint32_t* c_ind = s_ind[wave];           // bit o: short range o is RANGE_INDIRECT
    uint32_t* l_po = s_lpo[wave];            // per row: where its range list starts in `pairs`, and in the wave's concatenated list
    uint32_t* l_px = s_lpx[wave];
    uint8_t* nz = s_nz[wave];                // the rows that have ranges, in order
    uint32_t* row_bad = s_bad[wave];
    const uint32_t t = (blk << a.tpw_shift) + lane;
    const bool valid = lane < (1u << a.tpw_shift) && t < a.n_topics;
    const bool dbg_x = a.dbg_wave && (a.debug_flags & 4u); // BMQ_DEBUG=4: per-wave phase clocks of k_expand
    const unsigned long long xc0 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long xc_load = 0, xc_scan = 0, xc_gen = 0;
    if (a.blk_stats && ((blk & ((1u << SUPER_SHIFT) - 1u)) == (1u << SUPER_SHIFT) - 1u || blk == a.n_blocks - 1)) {
        // the batch statistics: the last wave of every super-block sums the records k_walk left for its (up to) 256 blocks
        unsigned long long v = 0, r = 0, b = 0;
        for (uint32_t i = ((blk >> SUPER_SHIFT) << SUPER_SHIFT) + lane; i <= blk; i += 64) {
            const uint4 q = a.blk_stats[i];
            v += q.x, r += q.y, b += q.z;
        }
        v = wave_sum_u64(v), r = wave_sum_u64(r), b = wave_sum_u64(b);
        if (lane == 0) {
            if (v) atomicAdd(&a.ctr->n_visit, v);
            if (r) atomicAdd(&a.ctr->n_ranges, r);
            if (b) atomicAdd(&a.ctr->topic_bytes, b);
        }
    }
    const uint32_t status = a.ctr->status;
    const uint32_t nr = valid ? a.route_cnt[t] : 0u;
    const uint32_t po = valid ? a.pair_off[t] : 0u; // requested together with the counts: one round trip less in front of the ranges
    const uint32_t np = valid ? a.pair_cnt[t] : 0u;
    uint32_t wtotal;
    const uint32_t excl = wave_excl_scan(nr, lane, wtotal);
    // ids in front of this wave's rows: whole super-blocks + the waves of this wave's own super-block before it
    unsigned long long wbase;
    {
        const uint32_t sb = blk >> SUPER_SHIFT;
        unsigned long long acc = 0;
        for (uint32_t i = lane; i < sb; i += 64) acc += a.super_sums[(size_t)i * SUPER_STRIDE];
        for (uint32_t i = (sb << SUPER_SHIFT) + lane; i < blk; i += 64) acc += a.wave_sums[i];
        wbase = wave_sum_u64(acc);
    }
    const unsigned long long row = wbase + excl;
    const unsigned long long wend = wbase + wtotal;
    const bool range_err = wend >= 0xFFFFFFFFull, no_space = wend > a.out_capacity;
    if (blk == a.n_blocks - 1 && lane == 0) { // the last wave knows the grand total
        a.ctr->total_ids = wend;
        *a.out_total = wend;
    }
    if ((range_err || no_space) && lane == 0) atomicOr(&a.ctr->status, range_err ? (uint32_t)ST_RANGE : (uint32_t)ST_NOSPACE);
    const bool writable = !(status & ST_RERUN) && !range_err && !no_space; // rows in front of the overflow are still written
    if (valid && !range_err) {
        a.out_row_ptr[t] = (uint32_t)row;
        if (t == a.n_topics - 1) a.out_row_ptr[a.n_topics] = (uint32_t)(row + nr);
    }
    if (!writable || wtotal == 0) return;
    uint32_t ptotal;
    const uint32_t pexcl = wave_excl_scan(np, lane, ptotal);
    // k_walk lays the ranges of a wave's 64 rows out as ONE contiguous piece of `pairs`, row after row (rows finished by
    // k_walk_slow live elsewhere: then every lane copies its own list)
    const unsigned long long m_np = __ballot(np != 0);
    const uint32_t first_l = m_np ? (uint32_t)__ffsll((long long)m_np) - 1u : 0u;
    const uint32_t po0 = __shfl(po, first_l) - __shfl(pexcl, first_l);
    const bool contiguous = __all(np == 0 || po == po0 + pexcl) && !(a.debug_flags & 64u); // (BMQ_DEBUG=64: experiment, always gather)
#if BMQ_EXP_PREFETCH
    MatchRange pf[EXP_EPL]; // the ranges of the coming pass (contiguous layout only)
#pragma unroll
    for (uint32_t i = 0; i < EXP_EPL; i++) {
        pf[i] = MatchRange{0u, 0u};
        const uint32_t k = lane + 64 * i;
        if (contiguous && k < ptotal) pf[i] = a.pairs[po0 + k];
    }
#endif
    row_bad[lane] = 0;
    l_po[lane] = po;
    l_px[lane] = pexcl;
    if (np) nz[rank_below(m_np)] = (uint8_t)lane;
    for (uint32_t i = lane; i < EXP_FLAG_WORDS; i += 64) flag[i] = 0ull;
    if (lane < EXP_K / 32) c_ind[lane] = 0u;
    if (lane < 2 * EXP_EPL) (&s_rs[wave][0][0])[lane] = 0ull;
    wave_sync();
    const unsigned long long xc1 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t pass = 0;
    // the row an entry of the current pass belongs to: rows with ranges are counted through the pass's row-start bitmap
    auto row_of = [&](const unsigned long long* rs, uint32_t ord0, uint32_t e) -> uint32_t {
        uint32_t c = (uint32_t)__popcll(rs[e >> 6] & ((2ull << (e & 63u)) - 1ull));
        for (uint32_t w = 0; w < (e >> 6); w++) c += (uint32_t)__popcll(rs[w]);
        return nz[ord0 + c - 1u];
    };
    unsigned long long out_done = 0; // output elements produced by earlier LDS passes
    uint32_t carry_last = 0;
    for (uint32_t k0 = 0; k0 < ptotal;) {
        const unsigned long long xp0 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
        // A pass takes EXP_K ranges, but never a part of a row that is ordered here (<= SORT_PAIRS ranges: usually 1-5; longer
        // lists are left to the order check + k_sort_rows): such a row waits for the next pass.
        uint32_t kn = min(EXP_K, ptotal - k0);
        {
            const uint32_t kend = k0 + kn;
            const unsigned long long m = __ballot(pexcl < kend && kend < pexcl + np && np <= SORT_PAIRS);
            if (m) kn = __shfl(pexcl, (int)__ffsll((long long)m) - 1) - k0;
        }
        const uint32_t lo = pexcl > k0 ? pexcl : k0, hi = min(pexcl + np, k0 + kn);
        // row-start bitmap of this pass: bit e = a row's list starts (or, for e = 0, continues) at entry e
        unsigned long long* rs = s_rs[wave][pass & 1u];
        if (lane < EXP_EPL) s_rs[wave][(pass + 1u) & 1u][lane] = 0ull; // the next pass's bitmap
        if (lo < hi) atomicOr(&rs[(lo - k0) >> 6], 1ull << ((lo - k0) & 63u));
        const unsigned long long m_k0 = __ballot(np != 0 && pexcl <= k0 && k0 < pexcl + np); // the row entry 0 belongs to
        const uint32_t lk = (uint32_t)__ffsll((long long)m_k0) - 1u;
        const uint32_t ord0 = (uint32_t)__popcll(m_np & ((1ull << lk) - 1ull));
        const bool continues = __shfl(pexcl, lk) < k0; // entry 0 continues the last row of the previous pass
        pass++;
        if (contiguous) { // one request per 64 ranges
#if BMQ_EXP_PREFETCH
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) {
                const uint32_t k = lane + 64 * i;
                if (k < kn) {
                    r_begin[k] = pf[i].begin;
                    r_cnt[k] = pf[i].count;
                }
            }
            // the next pass's ranges are requested now and land while this pass is produced
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) {
                const uint32_t k = k0 + kn + lane + 64 * i;
                if (k < ptotal) pf[i] = a.pairs[po0 + k];
            }
#else
            for (uint32_t k = lane; k < kn; k += 64) {
                const MatchRange r = a.pairs[po0 + k0 + k];
                r_begin[k] = r.begin;
                r_cnt[k] = r.count;
            }
#endif
            wave_sync();
        } else { // every row has its own list (retain direction, rows finished by k_walk_slow): gathered, still 64 ranges per request
            wave_sync();
            for (uint32_t e = lane; e < kn; e += 64) {
                const uint32_t l = row_of(rs, ord0, e);
                const MatchRange r = a.pairs[l_po[l] + (k0 + e - l_px[l])];
                r_begin[e] = r.begin;
                r_cnt[e] = r.count;
            }
            wave_sync();
        }
        // order this lane's own (whole) segment by first id
        if (np > 1 && np <= SORT_PAIRS && lo < hi) {
            const uint32_t sb = pexcl - k0;
            if (np <= 8) { // in registers: one round of LDS reads, a 19-comparator network, one round of writes
                uint32_t kk[8], bb[8], cc[8];
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    const bool in = i < np;
                    bb[i] = in ? r_begin[sb + i] : 0u;
                    cc[i] = in ? r_cnt[sb + i] : 0u;
                }
#pragma unroll
                for (uint32_t i = 0; i < 8; i++)
                    kk[i] = i < np ? ((cc[i] & RANGE_INDIRECT) ? a.ix.route_pos[bb[i]] : bb[i]) : 0xFFFFFFFFu;
#define BMQ_CEX(x, y) cex(kk[x], bb[x], cc[x], kk[y], bb[y], cc[y])
                BMQ_CEX(0, 1); BMQ_CEX(2, 3); BMQ_CEX(4, 5); BMQ_CEX(6, 7);
                BMQ_CEX(0, 2); BMQ_CEX(1, 3); BMQ_CEX(4, 6); BMQ_CEX(5, 7);
                BMQ_CEX(1, 2); BMQ_CEX(5, 6); BMQ_CEX(0, 4); BMQ_CEX(3, 7);
                BMQ_CEX(1, 5); BMQ_CEX(2, 6);
                BMQ_CEX(1, 4); BMQ_CEX(3, 6);
                BMQ_CEX(2, 4); BMQ_CEX(3, 5);
                BMQ_CEX(3, 4);
#undef BMQ_CEX
#pragma unroll
                for (uint32_t i = 0; i < 8; i++)
                    if (i < np) {
                        r_begin[sb + i] = bb[i];
                        r_cnt[sb + i] = cc[i];
                    }
            } else { // insertion sort in LDS
                for (uint32_t i = sb + 1; i < sb + np; i++) {
                    const uint32_t xb = r_begin[i], xc = r_cnt[i];
                    const uint32_t kx = (xc & RANGE_INDIRECT) ? a.ix.route_pos[xb] : xb;
                    uint32_t j = i;
                    while (j > sb) {
                        const uint32_t yb = r_begin[j - 1], yc = r_cnt[j - 1];
                        if (((yc & RANGE_INDIRECT) ? a.ix.route_pos[yb] : yb) <= kx) break;
                        r_begin[j] = yb;
                        r_cnt[j] = yc;
                        j--;
                    }
                    if (j != i) {
                        r_begin[j] = xb;
                        r_cnt[j] = xc;
                    }
                }
            }
        }
        wave_sync();
        const unsigned long long xp1 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
        uint32_t stot; // short ranges of this pass: elements | ranges << 16
        // exclusive prefixes: EXP_EPL consecutive entries per lane + wave scans.  Every range gets its output offset; a SHORT range
        // also its ordinal among the short ranges, its start in the short-range space (marked in the bitmap) and c_delta.
        {
            uint32_t eb[EXP_EPL], ec[EXP_EPL];
            uint32_t s = 0, ss = 0; // ss: short length sum | short count << 16
            const uint32_t e0 = lane * EXP_EPL;
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) {
                const uint32_t e = e0 + i;
                eb[i] = e < kn ? r_begin[e] : 0u;
                ec[i] = e < kn ? r_cnt[e] : 0u;
                const uint32_t len = ec[i] & ~RANGE_INDIRECT;
                s += len;
                if (len != 0 && len < EXP_LONG) ss += len + (1u << 16);
            }
            uint32_t tot;
            uint32_t run = wave_excl_scan(s, lane, tot);
            const uint32_t srun = wave_excl_scan(ss, lane, stot);
            uint32_t us = srun & 0xFFFFu, ord = srun >> 16;
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) {
                const uint32_t e = e0 + i;
                if (e < kn) {
                    const uint32_t len = ec[i] & ~RANGE_INDIRECT;
                    r_off[e] = run;
                    run += len;
                    if (len != 0 && len < EXP_LONG) {
                        c_delta[ord] = eb[i] - us;
                        if (ec[i] & RANGE_INDIRECT) atomicOr(&c_ind[ord >> 5], 1u << (ord & 31u));
                        atomicOr(&flag[us >> 6], 1ull << (us & 63u));
                        us += len;
                        ord++;
                    }
                }
            }
            if (lane == 63) r_off[kn] = tot;
        }
        wave_sync();
        const uint32_t T = r_off[kn];
        uint32_t* out = a.out_ids + wbase + out_done;
        // ids ascend inside a range by construction, so order is checked at range boundaries only: the first id of a range
        // against the last id of the previous range of the same row (also across LDS passes: carry_last)
        for (uint32_t e = lane; e < kn; e += 64) {
            const bool same_row = e ? !((rs[e >> 6] >> (e & 63u)) & 1ull) : continues;
            if (same_row) {
                const uint32_t fid = (r_cnt[e] & RANGE_INDIRECT) ? a.ix.route_pos[r_begin[e]] : r_begin[e];
                uint32_t plast = carry_last;
                if (e > 0) {
                    const uint32_t c = r_cnt[e - 1] & ~RANGE_INDIRECT;
                    plast = (r_cnt[e - 1] & RANGE_INDIRECT) ? a.ix.route_pos[r_begin[e - 1] + c - 1] : r_begin[e - 1] + c - 1;
                }
                if (fid <= plast) row_bad[row_of(rs, ord0, e)] = 1;
            }
        }
        {
            const uint32_t c = r_cnt[kn - 1] & ~RANGE_INDIRECT;
            carry_last = (r_cnt[kn - 1] & RANGE_INDIRECT) ? a.ix.route_pos[r_begin[kn - 1] + c - 1] : r_begin[kn - 1] + c - 1;
        }
        const unsigned long long xp2 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
        // element generation.  (Measured and dropped in round 2: one lane per short range + wave-streaming of everything above
        // 4 / 8 / 16 ids -- C3 k_expand 0.172 / 0.130 / 0.119 ms, C2 1.47 against 1.22 ms: streaming the many medium ranges one after
        // the other costs more than the LDS lookups save.)
        uint32_t ub = 0;          // short-range space: start of the current run of short ranges
        uint32_t fw = 0, fo = 0;  // fo = short ranges that start before bitmap word fw
        unsigned long long fm = flag[0], fnext = flag[1]; // words fw and fw + 1 (EXP_FLAG_WORDS has one spare word)
        for (uint32_t k = 0; k < kn;) {
            uint32_t kl = kn; // first long range at or after k
            for (uint32_t c0 = k; c0 < kn && kl == kn; c0 += 64) {
                const unsigned long long m = __ballot(c0 + lane < kn && (r_cnt[c0 + lane] & ~RANGE_INDIRECT) >= EXP_LONG);
                if (m) kl = c0 + (uint32_t)__ffsll((long long)m) - 1;
            }
            if (kl > k) {
                const uint32_t jb = r_off[k];
                const uint32_t ue = ub + (r_off[kl] - jb);
                for (uint32_t c = ub & ~63u; c < ue; c += 64) {
                    if (fw < (c >> 6)) { // the chunks of a pass are visited in order: at most one word further
                        fo += (uint32_t)__popcll(fm);
                        fw++;
                        fm = fnext;
                        fnext = flag[min(fw + 1u, EXP_FLAG_WORDS - 1u)]; // for the chunk after this one
                    }
                    const unsigned long long m = fm;
                    const uint32_t u = c + lane;
                    if (u >= ub && u < ue) {
                        const uint32_t o = fo + (uint32_t)__popcll(m & ((2ull << lane) - 1ull)) - 1u;
                        const uint32_t v = c_delta[o] + u;
                        const bool ind = (c_ind[o >> 5] >> (o & 31u)) & 1u;
                        out[jb + (u - ub)] = ind ? a.ix.route_pos[v] : v;
                    }
                }
                ub = ue;
            }
            if (kl < kn) {
                const uint32_t b = r_begin[kl], cf = r_cnt[kl], c = cf & ~RANGE_INDIRECT;
                uint32_t* dst = out + r_off[kl];
                if (cf & RANGE_INDIRECT)
                    for (uint32_t o = lane; o < c; o += 64) dst[o] = a.ix.route_pos[b + o];
                else { // consecutive ids: 16-byte stores (four ids per lane) between an aligning head and a tail
                    const uint32_t head = min((uint32_t)(((16u - ((uintptr_t)dst & 15u)) & 15u) >> 2), c);
                    if (lane < head) dst[lane] = b + lane;
                    uint4* d4 = reinterpret_cast<uint4*>(dst + head);
                    const uint32_t n4 = (c - head) >> 2;
                    for (uint32_t q = lane; q < n4; q += 64) {
                        const uint32_t v = b + head + 4 * q;
                        d4[q] = make_uint4(v, v + 1, v + 2, v + 3);
                    }
                    for (uint32_t o = head + 4 * n4 + lane; o < c; o += 64) dst[o] = b + o;
                }
            }
            k = kl + 1;
        }
        out_done += T;
        k0 += kn;
        wave_sync();
        // only what this pass marked is cleared for the next one
        for (uint32_t i = lane; i <= ((stot & 0xFFFFu) >> 6); i += 64) flag[i] = 0ull;
        if (lane < EXP_K / 32 && lane <= (stot >> 21)) c_ind[lane] = 0u;
        wave_sync();
        if (dbg_x) {
            const unsigned long long xp3 = __builtin_amdgcn_s_memtime();
            xc_load += xp1 - xp0, xc_scan += xp2 - xp1, xc_gen += xp3 - xp2;
        }
    }
    if (dbg_x && lane == 0) a.dbg_wave[blk] = make_uint4((uint32_t)(xc1 - xc0), (uint32_t)xc_load, (uint32_t)xc_scan, (uint32_t)xc_gen);
    if (valid && row_bad[lane] && nr > 1) {
        const uint32_t sp = atomicAdd(&a.ctr->sort_count, 1u);
        if (sp < a.sort_cap) a.sort_list[sp] = t;
        else atomicOr(&a.ctr->status, ST_NEED_SORTLIST);
    }
}


} // namespace bmq
