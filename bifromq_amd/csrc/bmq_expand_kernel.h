// bmq_expand_kernel.h -- k_expand: CSR row pointers + ids of one match batch (included by bmq_dist_kernels.h; both directions use it).
//
// What it replaces: the per-topic result lists TenantRouteMatcher.matchAll builds route by route (DW/cache/TenantRouteMatcher.java:118-158)
// and RetainMatcher's per-filter lists.  Input: what the walk kernels left per row -- a list of matched (begin, count) id ranges
// (`pairs`, pair_off / pair_cnt), the row's id count (route_cnt), per-wave and per-256-wave id sums.  Output: out_row_ptr[n + 1] and the
// ids of every row in ASCENDING order (the order the reference's sorted route keys give; rows whose ranges cannot be ordered here are
// listed for k_sort_rows).
//
// One wave (= one 64-thread workgroup) per 64 rows; the rows of a wave are one contiguous piece of the output:
//   head      row pointers: exclusive scan of the rows' id counts + the ids in front of the wave (<= 61 + 255 sums, no scan kernel);
//   per pass  (EXP_K ranges, whole rows where a row is ordered here)
//     load    the ranges, one 8-byte request per lane and 64 ranges (the walk kernel lays a wave's ranges out as one piece);
//     order   first id of a range against the last id of the range before it in the same row; where that fails the rows of up to
//             SORT_PAIRS ranges are RANK-sorted, all entries in parallel (an entry counts the keys of its row below its own -- the walk
//             discovers a topic's filters level by level, the ids ascend in depth-first order, so rows of three and more ranges
//             usually arrive out of order); what is still out of order goes to k_sort_rows;
//     prefix  exclusive prefix of the range lengths; SHORT ranges (< EXP_LONG ids) are flattened into one element space: a bitmap marks
//             the LAST element of every short range, so the range that covers element u is the number of marks below u -- two
//             v_mbcnt per 64 elements; ranges of EXP_LONG ids and more are streamed by the whole wave with 16-byte stores;
//     write   coalesced stores, all lanes busy whatever the mix of range lengths (a 5000-subscriber filter next to 60 singletons).
//
// Round 4 rewrite.  The round-2 kernel ordered a row's ranges with one LANE per row (an 8-input network in registers, insertion sort
// in LDS above that: the wave waited for its longest row), scanned with ds_bpermute shuffles and needed 125 VGPRs + 8.6 KB of LDS per
// wave: 4 waves per SIMD, 1 605 VALU instructions per wave on the survey's workload, 54 % of the wave cycles waiting
// (profiles/r04/c3_pmc_sq.csv).  This one: entry-parallel rank sort, DPP scans, end-marked bitmap, one wave per workgroup (75-80 VGPRs,
// 5.8 KB of LDS: 6 waves per SIMD), a body without a single load when the pass has no indirect ranges, the coming pass's ranges requested
// a pass ahead.  Measured (1 x MI355X, HIP events): C3 0.096 -> 0.060 ms, C2 1.09 -> 0.97 ms, C4 0.71 -> 0.62 ms.  Variants that were
// measured and lost: 192 ranges per pass at 7-8 waves per SIMD (C3 0.117 ms: more passes per wave cost more than the occupancy buys),
// streaming from 32 / 48 ids on (C2 1.16-1.17 ms), 5 waves per SIMD with the row look-ups of the four entries interleaved (0.063 / 1.00 / 0.62).
// The kernel's logic runs under the wave64 emulator of tools/emu/ on the host (tests/test_expand_emu.py).
#pragma once

namespace bmq {

constexpr uint32_t SORT_PAIRS = 32; // range lists up to this length are ordered here (rank sort in LDS)
#ifndef BMQ_EXP_K
#define BMQ_EXP_K 256
#endif
#ifndef BMQ_EXP_LONG
#define BMQ_EXP_LONG 64
#endif
#ifndef BMQ_EXP_CLOCKS
#define BMQ_EXP_CLOCKS 0
#endif
#ifndef BMQ_EXP_MIN_WAVES
#define BMQ_EXP_MIN_WAVES 6
#endif
constexpr uint32_t EXP_K = BMQ_EXP_K;       // ranges laid out per LDS pass
constexpr uint32_t EXP_LONG = BMQ_EXP_LONG; // ranges at least this long are streamed, shorter ones are flattened
constexpr uint32_t EXP_EPL = EXP_K / 64;    // entries per lane and pass
constexpr uint32_t EXP_FLAG_WORDS = (EXP_K * (EXP_LONG - 1) + 63) / 64 + 2;
static_assert(EXP_K % 64 == 0 && EXP_K >= 2 * SORT_PAIRS && EXP_K * (EXP_LONG - 1) < 65536, "short-range space: sums are packed into 16 bits below");

#ifndef BMQ_WAVE_EMU
__device__ __forceinline__ uint32_t read_lane(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
// v of the lane a DPP control names, 0 where that lane does not exist or the row is masked out
template <int CTRL, int ROWS> __device__ __forceinline__ uint32_t dpp_take(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false);
}
// bit `lane` of a wave-uniform mask
__device__ __forceinline__ uint32_t lane_bit(unsigned long long m) { return (uint32_t)(m >> threadIdx.x) & 1u; }
__device__ __forceinline__ uint32_t first_bit(unsigned long long m) { return (uint32_t)__ffsll((long long)m) - 1u; } // m != 0
__device__ __forceinline__ uint32_t count_bits(unsigned long long m) { return (uint32_t)__popcll(m); }
// a register copy that stays where it is written (the compiler otherwise sinks the hand-over of the prefetched ranges to the end of the pass loop
// -- behind the pass's stores, where waiting for the prefetch means waiting for the stores as well)
__device__ __forceinline__ uint32_t copy_here(uint32_t v) {
    uint32_t r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// a word every lane reads from the same address, through the scalar cache (nothing this wave could see change writes it during the launch)
__device__ __forceinline__ uint32_t uniform_word(const uint32_t* p) { return *scalar_words(p); }
#endif
__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {
    return ((unsigned long long)sgpr((uint32_t)(v >> 32)) << 32) | sgpr((uint32_t)v);
}
// inclusive wave scans on the DPP path (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast 15 into rows 1 and 3 and row_bcast 31
// into rows 2 and 3): six v_add_u32_dpp instead of six ds_bpermute round trips
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += dpp_take<0x111, 0xF>(v);
    v += dpp_take<0x112, 0xF>(v);
    v += dpp_take<0x114, 0xF>(v);
    v += dpp_take<0x118, 0xF>(v);
    v += dpp_take<0x142, 0xA>(v);
    v += dpp_take<0x143, 0xC>(v);
    return v;
}
template <int CTRL, int ROWS> __device__ __forceinline__ unsigned long long dpp_take64(unsigned long long v) {
    return ((unsigned long long)dpp_take<CTRL, ROWS>((uint32_t)(v >> 32)) << 32) | dpp_take<CTRL, ROWS>((uint32_t)v);
}
__device__ __forceinline__ unsigned long long wave_total_u64(unsigned long long v) { // the sum over the wave, wave-uniform
    v += dpp_take64<0x111, 0xF>(v);
    v += dpp_take64<0x112, 0xF>(v);
    v += dpp_take64<0x114, 0xF>(v);
    v += dpp_take64<0x118, 0xF>(v);
    v += dpp_take64<0x142, 0xA>(v);
    v += dpp_take64<0x143, 0xC>(v);
    return ((unsigned long long)read_lane((uint32_t)(v >> 32), 63) << 32) | read_lane((uint32_t)v, 63);
}

// The body of k_expand as a function of (argument block, block index): the kernel below is this and nothing else; the persistent matcher of
// the batching front (k_poll, bmq_poll_kernel.h) runs the same code behind its walk.
// (blk, r0, r1): the wave expands rows r0 .. r1 - 1 of block blk -- all 64 but for the blocks k_walk listed as heavy (bmq_batch_args.h).
__device__ __forceinline__ void expand_wave(const BatchArgs& a, const uint32_t block_x, const uint32_t r0 = 0u, const uint32_t r1 = 64u) {
    __shared__ uint32_t s_begin[EXP_K], s_cnt[EXP_K];
    __shared__ uint32_t s_delta[EXP_K + 4]; // order step: first id of every range | then, per SHORT range in order: first id (or route_pos index) - its
                                        // start in the short space; from the top down, per other range in order: its offset in the pass's output
    __shared__ unsigned long long s_flag[EXP_FLAG_WORDS]; // bit u: a short range ENDS at element u of the pass's short-range space
    __shared__ unsigned long long s_rs[2][EXP_EPL];       // bit e: a row's list starts (or, for e = 0, continues) at entry e of the pass
    __shared__ unsigned long long s_rows[1];              // rows whose ids do not come out ascending
    __shared__ uint32_t s_ind[EXP_K / 32];                // bit o: short range o is RANGE_INDIRECT
    __shared__ uint32_t l_po[64], l_px[65];               // per row: where its range list starts in `pairs`, and in the wave's concatenated list
    __shared__ uint8_t nz[64];                            // the rows that have ranges, in order
    uint32_t* const s_key = s_delta;
    const uint32_t lane = threadIdx.x;
    const uint32_t blk = block_x; // every wave owns one block of 2^tpw_shift rows
    if (blk >= a.n_blocks) return;
    const uint32_t t = (blk << a.tpw_shift) + lane;
    const bool valid = lane < (1u << a.tpw_shift) && t < a.n_topics;
    // per-wave phase clocks (BMQ_DEBUG=4) only in builds with -DBMQ_EXP_CLOCKS=1: five time stamps held across the pass loop cost the
    // scalar registers that keep the loop free of spills
    const bool dbg_x = BMQ_EXP_CLOCKS && BMQ_DBG(a, 4u) && a.dbg_wave;
    const unsigned long long xc0 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long xc_load = 0, xc_scan = 0, xc_gen = 0;
    // everything the head needs is requested before anything is waited for: no load below depends on another one, none sits in a branch
    // (rows beyond the batch read row 0 and drop what they get) -- one memory round trip
    const uint32_t status = uniform_word(&a.ctr->status); // (written by the kernels in front of this one only, as far as this wave cares)
    const uint32_t tt = valid ? t : 0u;
    uint32_t nr = a.route_cnt[tt], po = a.pair_off[tt], np = a.pair_cnt[tt];
    // ids in front of this wave's rows: whole super-blocks + the waves of this wave's own super-block before it
    unsigned long long acc = 0;
    {
        const uint32_t sb = blk >> SUPER_SHIFT, w0 = sb << SUPER_SHIFT;
        const unsigned long long s0 = a.super_sums[(size_t)min(lane, sb) * SUPER_STRIDE]; // (entry sb exists: this wave's own super-block)
        unsigned long long w[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) w[j] = a.wave_sums[min(w0 + lane + 64u * j, blk)]; // 2^SUPER_SHIFT = 4 x 64 waves at most
        static_assert(SUPER_SHIFT == 8, "four loads per lane cover a super-block");
        for (uint32_t i = lane + 64u; i < sb; i += 64) acc += a.super_sums[(size_t)i * SUPER_STRIDE]; // (batches of more than 1 M rows)
        if (lane < sb) acc += s0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
            if (w0 + lane + 64u * j < blk) acc += w[j];
    }
    if (!valid) nr = 0u, po = 0u, np = 0u;
    const bool mine = lane >= r0 && lane < r1; // this wave's part of the block: offsets come from all 64 rows, ranges and ids from these only
    if (!mine) np = 0u;
    if (r0 == 0u && a.blk_stats && ((blk & ((1u << SUPER_SHIFT) - 1u)) == (1u << SUPER_SHIFT) - 1u || blk == a.n_blocks - 1)) {
        // the batch statistics: the last wave of every super-block sums the records the walk left for its (up to) 256 blocks
        unsigned long long v = 0, r = 0, b = 0;
        for (uint32_t i = ((blk >> SUPER_SHIFT) << SUPER_SHIFT) + lane; i <= blk; i += 64) {
            const uint4 q = a.blk_stats[i];
            v += q.x, r += q.y, b += q.z;
        }
        v = wave_total_u64(v), r = wave_total_u64(r), b = wave_total_u64(b);
        if (lane == 0) {
            if (v) atomicAdd(&a.ctr->n_visit, v);
            if (r) atomicAdd(&a.ctr->n_ranges, r);
            if (b) atomicAdd(&a.ctr->topic_bytes, b);
        }
    }
    const uint32_t nr_incl = wave_incl_scan(nr);
    const uint32_t wtotal = read_lane(nr_incl, 63);
    const unsigned long long wbase = wave_total_u64(acc);
    const unsigned long long row = wbase + (nr_incl - nr);
    const unsigned long long wend = wbase + wtotal;
    const bool range_err = wend >= 0xFFFFFFFFull, no_space = wend > a.out_capacity;
    if (blk == a.n_blocks - 1 && lane == 0 && r0 == 0u) { // the last wave knows the grand total
        a.ctr->total_ids = wend;
        *a.out_total = wend;
    }
    if ((range_err || no_space) && lane == 0) atomicOr(&a.ctr->status, range_err ? (uint32_t)ST_RANGE : (uint32_t)ST_NOSPACE);
    const bool writable = !(status & ST_RERUN) && !range_err && !no_space; // rows in front of the overflow are still written
    if (valid && mine && !range_err) {
        a.out_row_ptr[t] = (uint32_t)row;
        if (t == a.n_topics - 1) a.out_row_ptr[a.n_topics] = (uint32_t)(row + nr);
    }
    if (!writable || wtotal == 0) return;
    const uint32_t np_incl = wave_incl_scan(np);
    const uint32_t ptotal = read_lane(np_incl, 63);
    if (ptotal == 0) return; // (a part of a split block without ranges)
    const uint32_t pexcl = np_incl - np;
    // the walk lays the ranges of a wave's 64 rows out as ONE contiguous piece of `pairs`, row after row (rows finished by k_walk_slow or
    // filled in by k_fill live elsewhere: then every entry is fetched from its own row's list)
    const unsigned long long m_np = ballot64(np != 0);
    const uint32_t first_l = m_np ? first_bit(m_np) : 0u;
    const uint32_t po0 = read_lane(po, first_l) - read_lane(pexcl, first_l);
    const bool contiguous = ballot64(np != 0 && po != po0 + pexcl) == 0ull && !BMQ_DBG(a, 64u); // (BMQ_DEBUG=64: experiment, always gather)
    l_po[lane] = po;
    l_px[lane] = pexcl;
    if (lane == 63) l_px[64] = ptotal;
    if (np) nz[rank_below(m_np)] = (uint8_t)lane;
    for (uint32_t i = lane; i < EXP_FLAG_WORDS; i += 64) s_flag[i] = 0ull;
    if (lane < EXP_K / 32) s_ind[lane] = 0u;
    if (lane == 0) s_rows[0] = 0ull;
    wave_sync();
    // A pass takes EXP_K ranges, but never a part of a row that is ordered here (<= SORT_PAIRS ranges): such a row waits for the next pass.
    struct Pass {
        uint32_t kn;        // ranges of the pass
        uint32_t ord0;      // ordinal (among the rows that have ranges) of the row entry 0 belongs to
        uint32_t continues; // entry 0 continues the last row of the pass before
    };
    // lays a pass out: its size, and in `rs` the row-start bitmap -- bit e = a row's list starts (or, for e = 0, continues) at entry e
    auto plan = [&](uint32_t k0, unsigned long long* rs) -> Pass {
        Pass p;
        p.kn = min(EXP_K, ptotal - k0);
        {
            const uint32_t kend = k0 + p.kn;
            const unsigned long long m = ballot64(pexcl < kend && kend < pexcl + np && np <= SORT_PAIRS);
            if (m) p.kn = read_lane(pexcl, first_bit(m)) - k0;
        }
        const uint32_t lo = pexcl > k0 ? pexcl : k0, hi = min(pexcl + np, k0 + p.kn);
        if (lane < EXP_EPL) rs[lane] = 0ull;
        wave_sync();
        if (lo < hi) atomicOr(&rs[(lo - k0) >> 6], 1ull << ((lo - k0) & 63u));
        const unsigned long long m_k0 = ballot64(np != 0 && pexcl <= k0 && k0 < pexcl + np); // the row entry 0 belongs to
        const uint32_t lk = first_bit(m_k0);
        p.ord0 = count_bits(m_np & ((1ull << lk) - 1ull));
        p.continues = read_lane(pexcl, lk) < k0 ? 1u : 0u;
        wave_sync();
        return p;
    };
    // requests a pass's ranges: one 8-byte request per lane and 64 ranges.  The ranges of pass p + 1 are requested at the START of pass p
    // and waited for before pass p's first store (below): a wait behind the stores would wait for the stores too.
    MatchRange pf[EXP_EPL];
    auto fetch = [&](uint32_t k0, const Pass& p, const unsigned long long* rs) {
        const uint32_t lane = lane_here(); // (opaque: the addresses below are computed here, not once in front of the loop and kept in registers)
        // no branch around a request, no request that keeps an old value: entries beyond the pass ask for its last entry again
        if (contiguous) {
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) pf[i] = a.pairs[po0 + k0 + min(lane + 64 * i, p.kn - 1u)];
        } else { // every row has its own list: entry e is fetched from the list of the row it belongs to
            uint32_t c = p.ord0 - 1u;
            uint32_t at[EXP_EPL];
            const uint32_t l0 = nz[p.ord0];
            const uint32_t at0 = l_po[l0] + (k0 - l_px[l0]); // entry 0 of the pass
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) {
                const unsigned long long w = sgpr64(rs[i]);
                const uint32_t e = lane + 64 * i;
                at[i] = at0;
                if (e < p.kn) {
                    const uint32_t l = nz[c + rank_below(w) + lane_bit(w)];
                    at[i] = l_po[l] + (k0 + e - l_px[l]);
                }
                c += count_bits(w);
            }
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) pf[i] = a.pairs[at[i]];
        }
    };
#pragma unroll
    for (uint32_t i = 0; i < EXP_EPL; i++) pf[i] = MatchRange{0u, 0u};
    Pass cur = plan(0u, s_rs[0]);
    fetch(0u, cur, s_rs[0]);
    uint32_t eb[EXP_EPL], ec[EXP_EPL]; // the ranges of the pass at hand, entry lane + 64 i
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), here and not at the first use inside the loop (where every later pass would wait again)
#pragma unroll
    for (uint32_t i = 0; i < EXP_EPL; i++) eb[i] = copy_here(pf[i].begin), ec[i] = copy_here(pf[i].count);
    const unsigned long long xc1 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t pass = 0;
    unsigned long long out_done = read_lane(nr_incl - nr, r0); // output elements in front of the pass: the block's rows in front of this wave's part, then what earlier LDS passes produced
    uint32_t carry_last = 0;         // last id of the previous pass's last range
    for (uint32_t k0 = 0; k0 < ptotal; pass++) {
        const unsigned long long xp0 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
        const uint32_t kn = cur.kn;
        const bool continues = cur.continues != 0u;
        const unsigned long long* rs = s_rs[pass & 1u];
        // the pass's row starts, wave-uniform: entry e = lane + 64 i belongs to the row with ordinal ord0 + (row starts at or below e) - 1
        unsigned long long rsw[EXP_EPL];
        uint32_t rsc[EXP_EPL];
        {
            uint32_t c = cur.ord0 - 1u;
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) {
                rsw[i] = sgpr64(rs[i]);
                rsc[i] = c;
                c += count_bits(rsw[i]);
            }
        }
        unsigned long long m_ind = 0;
#pragma unroll
        for (uint32_t i = 0; i < EXP_EPL; i++) m_ind |= ballot64(lane + 64 * i < kn && (ec[i] & RANGE_INDIRECT) != 0u);
        const bool more = k0 + kn < ptotal;
        Pass nxt = cur;
        if (more) {
            nxt = plan(k0 + kn, s_rs[(pass + 1u) & 1u]);
            fetch(k0 + kn, nxt, s_rs[(pass + 1u) & 1u]);
        }
        // IND: the pass has RANGE_INDIRECT ranges (ids through route_pos: routes added since the last rebuild).  The common case has none and
        // gets a body without a single load -- and so without a wait for one between its stores.
        auto body = [&](auto ind_tag) {
            constexpr bool IND = decltype(ind_tag)::value;
            auto first_id = [&](uint32_t b, uint32_t cf) -> uint32_t { return (IND && (cf & RANGE_INDIRECT)) ? a.ix.route_pos[b] : b; };
            auto last_id = [&](uint32_t b, uint32_t cf) -> uint32_t {
                const uint32_t c = cf & ~RANGE_INDIRECT;
                return (IND && (cf & RANGE_INDIRECT)) ? a.ix.route_pos[b + c - 1u] : b + c - 1u;
            };
            uint32_t ek[EXP_EPL];
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) {
                const uint32_t e = lane + 64 * i;
                ek[i] = eb[i];
                if (e < kn) {
                    ek[i] = first_id(eb[i], ec[i]);
                    s_begin[e] = eb[i];
                    s_cnt[e] = ec[i];
                    s_key[e] = ek[i];
                }
            }
            wave_sync();
            // order: the first id of a range against the last id of the range before it in the same row (ids ascend inside a range by
            // construction; across LDS passes: carry_last)
            unsigned long long any_bad = 0;
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) {
                const uint32_t e = lane + 64 * i;
                bool bad = false;
                if (e < kn && (e == 0u ? continues : lane_bit(rsw[i]) == 0u)) {
                    const uint32_t plast = e == 0u ? carry_last : last_id(s_begin[e - 1u], s_cnt[e - 1u]);
                    bad = ek[i] <= plast;
                }
                any_bad |= ballot64(bad);
            }
            if (any_bad) {
                // rank sort of the rows of 2 .. SORT_PAIRS ranges (they lie inside the pass as a whole): an entry's place among its row's
                // entries = the keys of the row below its own (equal keys: the earlier entry first)
                uint32_t to[EXP_EPL];
#pragma unroll
                for (uint32_t i = 0; i < EXP_EPL; i++) {
                    const uint32_t e = lane + 64 * i;
                    to[i] = e;
                    if (e < kn) {
                        const uint32_t l = nz[rsc[i] + rank_below(rsw[i]) + lane_bit(rsw[i])];
                        const uint32_t s0 = l_px[l], n = l_px[l + 1u] - s0;
                        if (n > 1u && n <= SORT_PAIRS) {
                            const uint32_t first = s0 - k0;
                            const unsigned long long mine = ((unsigned long long)ek[i] << 32) | e; // (key, entry): one 64-bit compare per element
                            uint32_t rank = 0;
                            const uint32_t end = first + n;
#pragma nounroll
                            for (uint32_t j = first; j < end; j += 4) { // four keys per round (the array has four entries of slack behind it)
                                const uint32_t q0 = s_key[j], q1 = s_key[j + 1u], q2 = s_key[j + 2u], q3 = s_key[j + 3u];
                                rank += ((((unsigned long long)q0) << 32) | j) < mine ? 1u : 0u;
                                rank += (j + 1u < end && ((((unsigned long long)q1) << 32) | (j + 1u)) < mine) ? 1u : 0u;
                                rank += (j + 2u < end && ((((unsigned long long)q2) << 32) | (j + 2u)) < mine) ? 1u : 0u;
                                rank += (j + 3u < end && ((((unsigned long long)q3) << 32) | (j + 3u)) < mine) ? 1u : 0u;
                            }
                            to[i] = first + rank;
                        }
                    }
                }
                wave_sync();
#pragma unroll
                for (uint32_t i = 0; i < EXP_EPL; i++)
                    if (lane + 64 * i < kn) {
                        s_begin[to[i]] = eb[i];
                        s_cnt[to[i]] = ec[i];
                        s_key[to[i]] = ek[i];
                    }
                wave_sync();
                // what is still out of order (rows of more than SORT_PAIRS ranges, ranges whose ids interleave) is left to k_sort_rows
#pragma unroll
                for (uint32_t i = 0; i < EXP_EPL; i++) {
                    const uint32_t e = lane + 64 * i;
                    if (e < kn && (e == 0u ? continues : lane_bit(rsw[i]) == 0u)) {
                        const uint32_t plast = e == 0u ? carry_last : last_id(s_begin[e - 1u], s_cnt[e - 1u]);
                        if (s_key[e] <= plast) {
                            const uint32_t l = nz[rsc[i] + rank_below(rsw[i]) + lane_bit(rsw[i])];
                            atomicOr(&s_rows[0], 1ull << l);
                        }
                    }
                }
            }
            carry_last = sgpr(last_id(s_begin[kn - 1u], s_cnt[kn - 1u]));
            wave_sync(); // the keys are dead from here on: their array takes the deltas
            const unsigned long long xp1 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
            // exclusive prefixes: EXP_EPL consecutive entries per lane + wave scans.  A SHORT range gets its ordinal among the short ranges,
            // its place in the short-range space (its last element marked in the bitmap) and its delta; any other range its output offset.
            uint32_t T, stot; // ids of this pass | its short ranges: elements | ranges << 16
            {
                uint32_t pb[EXP_EPL], pc[EXP_EPL];
                uint32_t s = 0, ss = 0;
                const uint32_t e0 = lane * EXP_EPL;
#pragma unroll
                for (uint32_t i = 0; i < EXP_EPL; i++) {
                    const uint32_t e = e0 + i;
                    pb[i] = e < kn ? s_begin[e] : 0u;
                    pc[i] = e < kn ? s_cnt[e] : 0u;
                    const uint32_t len = pc[i] & ~RANGE_INDIRECT;
                    s += len;
                    if (len - 1u < EXP_LONG - 1u) ss += len + (1u << 16);
                }
                const uint32_t s_incl = wave_incl_scan(s), ss_incl = wave_incl_scan(ss);
                T = read_lane(s_incl, 63);
                stot = read_lane(ss_incl, 63);
                uint32_t run = s_incl - s;
                uint32_t us = (ss_incl - ss) & 0xFFFFu, ord = (ss_incl - ss) >> 16;
#pragma unroll
                for (uint32_t i = 0; i < EXP_EPL; i++) {
                    const uint32_t e = e0 + i;
                    if (e < kn) {
                        const uint32_t len = pc[i] & ~RANGE_INDIRECT;
                        if (len - 1u < EXP_LONG - 1u) {
                            s_delta[ord] = pb[i] - us;
                            if (IND && (pc[i] & RANGE_INDIRECT)) atomicOr(&s_ind[ord >> 5], 1u << (ord & 31u));
                            us += len;
                            atomicOr(&s_flag[(us - 1u) >> 6], 1ull << ((us - 1u) & 63u));
                            ord++;
                        } else s_delta[EXP_K - 1u - (e - ord)] = run; // (empty ranges too: nothing is streamed for them)
                        run += len;
                    }
                }
            }
            wave_sync();
            const unsigned long long xp2 = dbg_x ? __builtin_amdgcn_s_memtime() : 0ull;
            // the coming pass's ranges have had the whole order / prefix step to arrive: waited for HERE, in front of this pass's stores
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
#pragma unroll
            for (uint32_t i = 0; i < EXP_EPL; i++) eb[i] = copy_here(pf[i].begin), ec[i] = copy_here(pf[i].count); // (this pass's copies are dead: all is in LDS)
            // element generation: runs of short ranges through the bitmap, the other ranges streamed in between
            {
                uint32_t* const out = a.out_ids + wbase + out_done;
                const bool other = (stot >> 16) != kn; // the pass has ranges that are not short
                uint32_t outp = 0;                     // elements of this pass in front of the current position
                uint32_t ub = 0;                       // short-range space: start of the current run of short ranges
                uint32_t lq = 0;                       // streamed ranges so far
                uint32_t fw = 0, fo = 0;               // fo = short ranges that end before bitmap word fw
                unsigned long long fm = sgpr64(s_flag[0]), fnext = sgpr64(s_flag[1]); // words fw and fw + 1 (EXP_FLAG_WORDS has one spare word)
                for (uint32_t k = 0; k < kn;) {
                    uint32_t kl = kn; // first range at or after k that is not short
                    if (other)
                        for (uint32_t c0 = k; c0 < kn && kl == kn; c0 += 64) {
                            const unsigned long long m = ballot64(c0 + lane < kn && (s_cnt[c0 + lane] & ~RANGE_INDIRECT) - 1u >= EXP_LONG - 1u);
                            if (m) kl = c0 + first_bit(m);
                        }
                    const uint32_t run_end = kl < kn ? sgpr(s_delta[EXP_K - 1u - lq]) : T;
                    if (run_end != outp) {
                        const uint32_t ue = ub + (run_end - outp);
                        uint32_t* const dst = out + outp;
                        for (uint32_t c = ub & ~63u; c < ue; c += 64) {
                            if (fw < (c >> 6)) { // the chunks of a pass are visited in order: at most one word further
                                fo += count_bits(fm);
                                fw++;
                                fm = fnext;
                                fnext = sgpr64(s_flag[min(fw + 1u, EXP_FLAG_WORDS - 1u)]); // for the chunk after this one
                            }
                            const uint32_t u = c + lane;
                            if (u >= ub && u < ue) {
                                const uint32_t o = fo + rank_below(fm);
                                uint32_t v = s_delta[o] + u;
                                if (IND && ((s_ind[o >> 5] >> (o & 31u)) & 1u)) v = a.ix.route_pos[v];
                                dst[u - ub] = v;
                            }
                        }
                        ub = ue;
                    }
                    if (kl < kn) {
                        const uint32_t b = sgpr(s_begin[kl]), cf = sgpr(s_cnt[kl]), c = cf & ~RANGE_INDIRECT;
                        uint32_t* const dst = out + run_end;
                        if (IND && (cf & RANGE_INDIRECT)) {
                            for (uint32_t o = lane; o < c; o += 64) dst[o] = a.ix.route_pos[b + o];
                        } else { // consecutive ids: 16-byte stores (four ids per lane) between an aligning head and a tail
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
                        outp = run_end + c;
                        lq++;
                    }
                    k = kl + 1;
                }
            }
            out_done += T;
            if (more) { // only what this pass marked is cleared for the next one
                wave_sync();
                for (uint32_t i = lane; i <= ((stot & 0xFFFFu) >> 6); i += 64) s_flag[i] = 0ull;
                if (IND && lane < EXP_K / 32 && lane <= (stot >> 21)) s_ind[lane] = 0u;
                wave_sync();
            }
            if (dbg_x) {
                const unsigned long long xp3 = __builtin_amdgcn_s_memtime();
                xc_load += xp1 - xp0, xc_scan += xp2 - xp1, xc_gen += xp3 - xp2;
            }
        };
        if (m_ind) body(std::true_type{});
        else body(std::false_type{});
        k0 += kn;
        cur = nxt;
    }
    if (dbg_x && lane == 0) a.dbg_wave[blk] = make_uint4((uint32_t)(xc1 - xc0), (uint32_t)xc_load, (uint32_t)xc_scan, (uint32_t)xc_gen);
    wave_sync();
    if (valid && lane_bit(s_rows[0]) && nr > 1) {
        const uint32_t sp = atomicAdd(&a.ctr->sort_count, 1u);
        if (sp < a.sort_cap) a.sort_list[sp] = t;
        else atomicOr(&a.ctr->status, ST_NEED_SORTLIST);
    }
}

// Grid: [EXPAND_PARTS - 1 helper waves per entry of heavy_list's capacity] [one wave per block].  The helpers come first in the dispatch
// order: the launch begins with the heavy blocks' rows 16-63, a helper without an entry leaves after one scalar load.
__global__ __launch_bounds__(64, BMQ_EXP_MIN_WAVES) void k_expand(BatchArgs a) {
    uint32_t bx = blockIdx.x;
    if (a.heavy_list == nullptr) return expand_wave(a, bx);
    constexpr uint32_t HP = EXPAND_PARTS - 1u, RP = 64u / EXPAND_PARTS;
    const uint32_t n_help = HP * a.heavy_cap;
    if (bx < n_help) {
        const uint32_t listed = min(uniform_word(&a.ctr->heavy_count), a.heavy_cap);
        const uint32_t h = bx / HP, part = 1u + bx % HP;
        if (h >= listed) return;
        return expand_wave(a, uniform_word(&a.heavy_list[h]), part * RP, (part + 1u) * RP);
    }
    bx -= n_help;
    if (bx >= a.n_blocks) return;
    const bool split = uniform_word(reinterpret_cast<const uint32_t*>(a.blk_stats + bx) + 3) != 0u;
    expand_wave(a, bx, 0u, split ? RP : 64u);
}

} // namespace bmq
