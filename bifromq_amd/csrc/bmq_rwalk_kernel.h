// bmq_rwalk_kernel.h -- k_retain_walk<G, DYN>: the walk of the retain direction (wildcard FILTERS against the index of retained TOPICS,
// bmq_retain.h).  Semantics: RetainMatcher (RS/index/RetainTopicIndex.java:36-124) driven by TopicLevelTrie.lookup
// (UTIL/index/TopicLevelTrie.java:190-249); a wildcard in the first topic level skips the '$' children (currentLevel == 1 there: level 0
// is the tenant).
//
// Round 5 rewrite.  The kernel it replaces walked ONE filter per wave: its 64 lanes shared one chain of dependent fetches, a level of
// 512 frontier nodes was eight fetch-and-wait trips, and the counters said what that costs (profiles/r05/c4_pmc_sq_before.csv: 70 % of the
// wave cycles in s_waitcnt, 2 000 instructions per filter more than half of them scalar bookkeeping, 102 spilled SGPRs).  This one
// works on G filters per wave at a time and hands the wave's 64 lanes out to UNITS of work, whichever filters they belong to:
//   tokenise   the group's filter bytes are staged in LDS; 64 / G lanes per filter find their level, hash it and look it up in the
//              dictionary together: one fetch for all levels of all filters of the group;
//   rounds     every filter ("slot") is walked breadth first, one level at a time: its frontier is a node RANGE (the root; the children
//              of a node range after '+': no storage at all) or a LIST of nodes (the children found by a literal level: the first
//              RW_INL entries in LDS, the rest in the wave's arena in global memory).  A round gives every slot its fair share of the
//              64 lanes and the rest of the lanes to whoever has more units; a unit is one node of the frontier against the level:
//                literal   one aligned 64-byte bucket of the edge hash (the home bucket of (node + hash(token)): the buckets of the
//                          nodes of a RANGE are CONSECUTIVE lines); the entry carries the child and whether a topic ends there, so
//                          a filter's last literal level emits without a further fetch;
//                '+'       a range maps to the range of its children (first and last node: two 16-byte reads); a list of single nodes is
//                          expanded into the list of their children;
//                '#', end  one 16-byte node: its subtree's id range / its own topic;
//              all of a round's fetches are requested before the one wait; found children and matched ranges are appended in the
//              order of the frontier (ballot + mbcnt inside the slot's lane segment), which is ascending id order: rows leave ordered.
//   results    matched (begin, count) ranges go straight to `pairs`: a slot reserves room for the at most U ranges of its final level's U
//              units from a chunk the wave holds (a wave-uniform bump; one atomic per chunk, not per filter).
// Slot state lives in the registers of the slot's HOME lane (lane s for slot s); a round reads it through ds_bpermute.
// Filters of more than RW_LV levels (MQTT ingress rejects more than 16: Setting.MaxTopicLevels) are listed for k_retain_walk_deep.
// DYN (topics removed / added since the bulk load): live counts through the DEAD bitmap; the overlay trie is walked by
// k_retain_overlay afterwards (bmq_retain_kernels.h).
// The kernel's logic runs on the host under the wave64 emulator (tools/emu/rwalk_emu.cpp, tests/test_rwalk_emu.py).
#pragma once

namespace bmq {

constexpr uint32_t RW_LV = 16; // levels of a filter walked here
#ifndef BMQ_RW_INL
#define BMQ_RW_INL 16
#endif
#ifndef BMQ_RW_G
#define BMQ_RW_G 8
#endif
#ifndef BMQ_RW_MIN_WAVES
#define BMQ_RW_MIN_WAVES 4
#endif
#ifndef BMQ_RW_CHUNK
#define BMQ_RW_CHUNK 512
#endif
constexpr uint32_t RW_INL = BMQ_RW_INL;     // list entries kept in LDS (per list; a slot has two)
constexpr uint32_t RW_G = BMQ_RW_G;         // filters a wave works on at a time
constexpr uint32_t RW_CHUNK = BMQ_RW_CHUNK; // matched-range entries a wave reserves at a time
constexpr uint32_t RW_STAGE = 512;          // bytes of filter text staged in LDS per refill (more: read from global memory)
constexpr uint32_t RT_END = 0xFFFFFFFBu;    // level kind behind a filter's last level: topics that END at a frontier node
constexpr uint32_t ST_RETAIN_LIST = 512u;   // a frontier list outgrew the wave's arena (the batch is re-run with a larger one)
// slot flags
constexpr uint32_t RF_RANGE = 1u;   // the frontier is a node range (else: a list)
constexpr uint32_t RF_PAR = 2u;     // which of the slot's two lists is the CURRENT one
constexpr uint32_t RF_LASTLIT = 4u; // this level is the filter's last and a literal: found children emit
constexpr uint32_t RF_L0 = 8u;      // this is the filter's first level (the frontier is the tenant's root)
constexpr uint32_t RF_EMIT = 16u;   // this level emits (room in `pairs` is reserved)

template <int G> struct RwLds {
    static_assert(G >= 2 && G <= 8 && (G & (G - 1)) == 0, "slots per wave: G / 2 new filters x 16 level lanes fill the wave's 64 lanes");
    uint32_t tok[G][RW_LV];     // level kinds / dictionary tokens of the group's filters
    uint32_t lst[G][2][RW_INL]; // the first RW_INL entries of every list
    uint32_t stg[RW_STAGE / 4]; // tokeniser: the staged bytes of the filters taken on
    uint32_t ten[G][8];         // the slot's tenant: node_base, edge_base, edge_bucket_mask, id_base, sys_node_lo, sys_node_hi, sys_id_lo, sys_id_hi
    uint32_t mbox[G][4];        // what a '+' unit over a RANGE leaves for the slot's home lane: child range (begin, count), hole (begin, count)
    uint32_t nr[G];             // ids matched (added up by the emitting lanes)
    uint32_t nlev[G];           // levels (0: nothing to walk)
    uint32_t fid[G];            // the filter (row of the batch) a refill put into the slot
    uint32_t ften[G], flen[G];  // ... its tenant (index in the batch's tenant table), its bytes
    uint32_t tc[10];            // the tenant resolved last: index in the batch's tenant table, its 8 words, known
};

#ifndef BMQ_WAVE_EMU
__device__ __forceinline__ uint32_t rw_read_lane(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
// four 16-byte reads from four addresses per lane, all requested before the one wait
__device__ __forceinline__ void rw_load4(bool act, const void* q0, const void* q1, const void* q2, const void* q3, uint4& v0, uint4& v1, uint4& v2, uint4& v3) {
    v0 = v1 = v2 = v3 = make_uint4(0u, 0u, 0u, 0u);
    if (act) {
        uint4 t0, t1, t2, t3;
        asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                     "global_load_dwordx4 %1, %5, off\n\t"
                     "global_load_dwordx4 %2, %6, off\n\t"
                     "global_load_dwordx4 %3, %7, off\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "v"(q0), "v"(q1), "v"(q2), "v"(q3)
                     : "memory");
        v0 = t0, v1 = t1, v2 = t2, v3 = t3;
    }
}
__device__ __forceinline__ uint32_t rw_ctz(uint32_t v) { return (uint32_t)__builtin_ctz(v); }
// a use of v that stays where it is written: the compiler's wait for the load that produces v lands in front of it
__device__ __forceinline__ void rw_settle(uint32_t& v) { asm volatile("" : "+v"(v)); }
#else
inline void rw_settle(uint32_t&) {}
inline uint32_t rw_read_lane(uint32_t v, uint32_t l) { return read_lane(v, l); }
inline void rw_load4(bool act, const void* q0, const void* q1, const void* q2, const void* q3, uint4& v0, uint4& v1, uint4& v2, uint4& v3) {
    v0 = v1 = v2 = v3 = make_uint4(0u, 0u, 0u, 0u);
    if (act) {
        v0 = *reinterpret_cast<const uint4*>(q0);
        v1 = *reinterpret_cast<const uint4*>(q1);
        v2 = *reinterpret_cast<const uint4*>(q2);
        v3 = *reinterpret_cast<const uint4*>(q3);
    }
}
inline uint32_t rw_ctz(uint32_t v) { return (uint32_t)__builtin_ctz(v); }
#endif

// bits of `m` inside the lane segment [off, off + n)
__device__ __forceinline__ unsigned long long rw_seg(unsigned long long m, uint32_t off, uint32_t n) {
    const unsigned long long s = m >> off;
    return n >= 64u ? s : (s & ((1ull << n) - 1ull));
}

// One level of a '/'-separated filter starting at pos (bmq_dist_kernels.h scan_level, restated over a word source so that the emulator
// compiles it): hash, first 16 bytes, length.
template <class WordAt>
__device__ __forceinline__ void rw_scan_level(uint32_t pos, uint32_t end, bool split, WordAt&& word_at, LevelHash& h, uint32_t inl[4], uint32_t& len) {
    h = level_hash_init();
    inl[0] = inl[1] = inl[2] = inl[3] = 0;
    len = 0;
    for (;;) {
        const uint32_t remaining = end - pos;
        if (remaining == 0) break;
        const uint32_t w = word_at(pos);
        uint32_t nb = 4;
        if (split) {
            const uint32_t x = w ^ 0x2F2F2F2Fu;
            const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u; // exact for the lowest hit, which is all that is used
            if (z) nb = rw_ctz(z) >> 3;
        }
        nb = nb < remaining ? nb : remaining;
        if (nb) {
            const uint32_t wm = nb == 4 ? w : (w & ((1u << (8u * nb)) - 1u));
            level_hash_word(h, wm);
            if (len < 16) inl[len >> 2] = wm;
            len += nb;
            pos += nb;
        }
        if (nb < 4) break;
    }
}

// -DBMQ_RW_CLOCKS=1 (profiling builds, tools/build_variant.sh): where a wave's time goes, summed over the launch into g_rw_clk:
// waves, groups, rounds, units, rounds that ran the cold probe loop | ticks: whole wave, tokeniser + tenants, the rounds' fetch waits,
// cold probe loops, slowest wave
#ifndef BMQ_RW_CLOCKS
#define BMQ_RW_CLOCKS 0
#endif
#if BMQ_RW_CLOCKS && !defined(BMQ_WAVE_EMU)
__device__ unsigned long long g_rw_clk[16];
#define RW_CLK() __builtin_amdgcn_s_memtime()
#else
#define RW_CLK() 0ull
#endif

template <int G, bool DYN>
__device__ __forceinline__ void retain_walk_rounds(const RetainArgs& r, const BatchArgs& a, RwLds<G>& L) {
    constexpr uint32_t RQ = G / 2;   // filters taken on at a time: a refill happens when that many slots are free
    constexpr uint32_t SL = 64 / RQ; // tokeniser: lanes (= levels) of one new filter
    constexpr uint32_t FAIR = 64 / G; // rounds: a slot's fair share of the lanes
    static_assert(SL >= RW_LV, "a filter's levels are tokenised in one pass");
    const uint32_t lane = threadIdx.x;
    const bool home = lane < (uint32_t)G;
    const RetainDynView dyn = r.ix.dyn;
    uint32_t* const arena = r.rw_arena + (size_t)blockIdx.x * G * 2 * r.rw_cap; // per slot two lists of rw_cap entries behind the RW_INL in LDS
    const uint32_t list_cap = RW_INL + r.rw_cap;
    // the batch is handed out in QUADS of RQ consecutive filters, dynamically: quad q belongs to partition q % RW_PARTS, a wave takes the
    // next quad of its partition with one atomic (requested a refill ahead of its use: the wave never waits for it)
    const uint32_t n_quads = (r.n_filters + RQ - 1) / RQ, n_parts = gridDim.x < RW_PARTS ? gridDim.x : RW_PARTS, part = blockIdx.x % n_parts;
    uint32_t visits = 0;                        // per lane
    unsigned long long wranges = 0, wbytes = 0; // home lanes
    bool ovf = false;
    uint32_t ck_next = 0, ck_end = 0; // the wave's chunk of `pairs` (wave-uniform)
    if (lane == 0) L.tc[0] = NONE;
    wave_sync();
    unsigned long long c_tok = 0, c_wait = 0, c_cold = 0, n_rounds = 0, n_units = 0, n_cold = 0, n_grp = 0;
    unsigned long long c_s0 = 0, c_s1 = 0, c_s2 = 0, c_s3 = 0, c_s4 = 0; // round segments: hand-out + unit | found + '+' | emission | hand-over | reserve
    (void)c_s0, (void)c_s1, (void)c_s2, (void)c_s3, (void)c_s4;
    const unsigned long long c_start = RW_CLK();
    (void)c_start, (void)c_tok, (void)c_wait, (void)c_cold, (void)n_rounds, (void)n_units, (void)n_cold, (void)n_grp;

    auto list_put = [&](uint32_t s, uint32_t par, uint32_t i, uint32_t v) {
        if (i < RW_INL) L.lst[s][par][i] = v;
        else if (i < list_cap) arena[((size_t)s * 2 + par) * r.rw_cap + (i - RW_INL)] = v;
        else ovf = true;
    };
    uint32_t fetched = 0; // lane 0: the partition's counter as the last request found it
    auto request_quad = [&]() {
        if (lane == 0) fetched = atomicAdd(&a.ctr->rw_next[part], 1u);
    };
    request_quad();

    // ---- slot state: the registers of the slot's home lane -------------------------------------------------------------------------
    uint32_t h_f = NONE, h_nlev = 0;
    uint32_t h_kind = 0, h_fl = 0, h_sb = 0, h_sc = 0, h_hb = 0, h_hc = 0, h_U = 0, h_cu = 0, h_lvl = 0;
    uint32_t h_nxn = 0;                      // entries of the NEXT list so far
    uint32_t h_pb = 0, h_np = 0, h_pcap = 0; // room reserved in `pairs`, ranges written
    bool h_live = false;                     // the slot has units left
    uint32_t h_need = 0;                     // room in `pairs` this slot asks for
    // the level the slot is at: its kind, units, flags; h_need = room for its ranges if it emits
    auto enter_level = [&]() {
        h_kind = h_lvl < h_nlev ? L.tok[lane][h_lvl] : RT_END;
        h_fl &= ~(RF_LASTLIT | RF_L0 | RF_EMIT);
        if (h_lvl == 0) h_fl |= RF_L0;
        const bool lit = h_kind < RT_END;
        if (lit && h_lvl + 1 == h_nlev) h_fl |= RF_LASTLIT;
        h_U = (h_kind == RT_PLUS && (h_fl & RF_RANGE)) ? (h_sc ? 1u : 0u) : h_sc;
        h_cu = 0;
        h_nxn = 0;
        h_live = h_U != 0 && h_kind != TOK_UNKNOWN;
        if (h_live && (h_kind == RT_HASH || h_kind == RT_END || (h_fl & RF_LASTLIT))) {
            h_fl |= RF_EMIT;
            h_need = h_U + 1u; // (the filter "#" emits two ranges from its one unit)
        }
    };
    // a slot whose filter is answered: its row's range list, counts; the slot is free again
    auto finish = [&]() {
        const uint32_t np = h_np < h_pcap ? h_np : h_pcap, nr = (h_np <= h_pcap) ? L.nr[lane] : 0u;
        a.pair_off[h_f] = np ? h_pb : 0u;
        a.pair_cnt[h_f] = np;
        a.route_cnt[h_f] = np ? nr : 0u; // (the per-block sums k_expand wants are added up by k_retain_sums)
        wranges += np;
        h_f = NONE;
        h_need = 0;
    };
    // room in `pairs` for the slots that ask: a wave-uniform bump inside the wave's chunk, a new chunk when it is used up
    auto reserve = [&]() {
        if (ballot64(home && h_need != 0) == 0) return;
        uint32_t total = 0, mine = 0;
#pragma unroll
        for (uint32_t s = 0; s < (uint32_t)G; s++) {
            const uint32_t n = rw_read_lane(h_need, s);
            if (lane == s) mine = total;
            total += n;
        }
        if (ck_end - ck_next < total) { // (rare) one atomic per chunk
            const uint32_t want = total > RW_CHUNK ? total : RW_CHUNK;
            unsigned long long base = 0;
            uint32_t ok = 1;
            if (lane == 0) ok = pair_alloc(a.subs, a.pair_cap, blockIdx.x, want, base) ? 1u : 0u;
            ok = sgpr(ok);
            ck_next = sgpr((uint32_t)base);
            ck_end = ck_next + want;
            if (!ok) {
                if (lane == 0) atomicOr(&a.ctr->status, ST_NEED_PAIRS);
                ck_next = ck_end = 0; // nothing is written: see the bound check of the emitting lanes
            }
        }
        const bool fits = ck_end - ck_next >= total;
        if (home && h_need) {
            h_pb = ck_next + mine;
            h_pcap = fits ? h_need : 0u;
            h_need = 0;
        }
        if (fits) ck_next += total;
    };

    bool exhausted = false;
    for (;;) {
        const unsigned long long m_busy = ballot64(home && h_f != NONE);
        // ==== refill: RQ slots are free and the batch has filters left =================================================================
        if (!exhausted && (uint32_t)G - count_bits(m_busy) >= RQ) {
            const unsigned long long c_g0 = RW_CLK();
            const uint32_t quad = sgpr(fetched) * n_parts + part;
            if (quad >= n_quads) {
                exhausted = true;
                continue;
            }
            request_quad(); // the one after this
            n_grp++;
            const uint32_t f0 = quad * RQ;
            const uint32_t n_in = r.n_filters - f0 < RQ ? r.n_filters - f0 : RQ;
            const uint32_t m_free = ~(uint32_t)m_busy & ((1u << G) - 1u);
            // ---- tokenise: lane (tj, tk) = level tk of the quad's filter tj, which goes to the tj-th free slot ----------------------------
            const uint32_t tj = lane / SL, tk = lane % SL;
            const bool tvalid = tj < n_in;
            uint32_t ts;
            {
                uint32_t mm = m_free;
                for (uint32_t i = 0; i < tj; i++) mm &= mm - 1u;
                ts = rw_ctz(mm | (1u << (G - 1)));
            }
            uint32_t beg = 0, end = 0, ften = NONE;
            if (tvalid) { // (requested together: one trip)
                beg = r.filter_off[f0 + tj];
                end = r.filter_off[f0 + tj + 1];
                if (tk == 0) ften = r.filter_tenant[f0 + tj];
            }
            const uint32_t gbeg = sgpr(beg), gend = rw_read_lane(end, (n_in - 1) * SL);
            const uint32_t a0 = gbeg & ~15u;
            const bool staged = (gend - a0) + 32u <= RW_STAGE;
            uint32_t* const sw = L.stg;
            if (staged) { // coalesced 16-byte copies of the quad's contiguous filter bytes
                const uint32_t n16 = (gend - a0 + 15u) >> 4;
                for (uint32_t o = lane; o < n16; o += 64) reinterpret_cast<uint4*>(sw)[o] = reinterpret_cast<const uint4*>(r.filters + a0)[o];
            }
            wave_sync();
            uint32_t nlev = 0;
            // the quad's levels: count the '/' bytes, find level tk, hash it, look it up -- over LDS words or, for a quad whose bytes do not fit the
            // staging area, over global words (two instantiations: ONE body over "staged ? LDS : global" reads through generic addresses, and a
            // flat load waits for LDS and every global access in flight -- per word)
            auto tokenise = [&](auto aligned_word) {
                auto word_at = [&](uint32_t p) -> uint32_t { // 4 bytes at any p (the packed input is padded by 16 bytes)
                    const uint32_t lo = aligned_word(p & ~3u), hi = aligned_word((p & ~3u) + 4u), sh = 8u * (p & 3u);
                    return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
                };
                uint32_t lstart = beg;
                if (tvalid) { // '/' bytes in front of level tk, and in the whole filter (UTIL/TopicUtil.java:206-225: empty levels count)
                    uint32_t cnt = 0;
                    for (uint32_t p = beg & ~3u; p < end; p += 4) {
                        const uint32_t x = aligned_word(p) ^ 0x2F2F2F2Fu;
                        uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); // 0x80 in every byte that is '/'
                        if (p < beg) z &= 0xFFFFFFFFu << (8u * (beg - p));
                        if (p + 4 > end) z &= 0xFFFFFFFFu >> (8u * (p + 4 - end));
                        const uint32_t n = (uint32_t)__builtin_popcount(z);
                        if (cnt < tk && cnt + n >= tk) {
                            uint32_t zz = z;
                            for (uint32_t i = cnt + 1; i < tk; i++) zz &= zz - 1u;
                            lstart = p + (rw_ctz(zz) >> 3) + 1u;
                        }
                        cnt += n;
                    }
                    nlev = cnt + 1;
                }
                if (tvalid && tk < nlev && nlev <= RW_LV) {
                    LevelHash h;
                    uint32_t inl[4], len;
                    rw_scan_level(lstart, end, true, word_at, h, inl, len);
                    uint32_t tok;
                    if (len == 1 && inl[0] == '+') tok = RT_PLUS;
                    else if (len == 1 && inl[0] == '#' && tk == nlev - 1) tok = RT_HASH;
                    else {
#ifndef BMQ_WAVE_EMU
                        DistIndexView dv{};
                        dv.dict = r.ix.dict;
                        dv.dict_group_mask = r.ix.dict_group_mask;
                        dv.pool = r.ix.pool;
                        const uint8_t* fb = r.filters;
                        tok = dict_lookup(dv, h, len, inl, lstart, [&](uint32_t k) -> uint32_t { return fb[k]; });
#else
                        tok = rdict_find(r.ix, h, len, inl, r.filters, lstart);
#endif
                    }
                    L.tok[ts][tk] = tok;
                }
            };
            if (staged) tokenise([&](uint32_t p) -> uint32_t { return sw[(p - a0) >> 2]; }); // p % 4 == 0
            else tokenise([&](uint32_t p) -> uint32_t { return *reinterpret_cast<const uint32_t*>(r.filters + p); });
            if (tvalid && tk == 0) {
                L.nlev[ts] = nlev <= RW_LV ? nlev : 0u;
                L.fid[ts] = f0 + tj;
                L.ften[ts] = ften;
                L.flen[ts] = end - beg;
                if (nlev > RW_LV) r.deep_list[atomicAdd(&a.ctr->slow_count, 1u)] = f0 + tj; // left empty here; the deep pass answers it
            }
            wave_sync();
            // ---- the new slots' tenants (home lanes; a batch mostly asks for one tenant: the one resolved last is remembered) -------------
            bool h_known = false;
            const bool fresh = home && ((m_free >> lane) & 1u) && rank_below((unsigned long long)m_free) < n_in;
            if (fresh) {
                h_f = L.fid[lane];
                h_nlev = L.nlev[lane];
                const uint32_t ti = L.ften[lane];
                uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ti < r.n_tenants) {
                    if (ti == L.tc[0]) {
                        for (uint32_t k = 0; k < 8; k++) w[k] = L.tc[1 + k];
                        h_known = L.tc[9] != 0;
                    } else {
                        const uint8_t* tb = r.tenants;
                        const uint32_t tbeg = r.tenant_off[ti], tend = r.tenant_off[ti + 1];
                        LevelHash h;
                        uint32_t inl[4], len;
                        rw_scan_level(tbeg, tend, false, [&](uint32_t p) -> uint32_t { return bytes_word_at(tb, p); }, h, inl, len);
                        const uint32_t ttok = rdict_find(r.ix, h, len, inl, tb, tbeg);
                        const RTenantSlot* t = ttok != TOK_UNKNOWN ? rtenant_find(r.ix, ttok) : nullptr;
                        if (t) {
                            w[0] = t->node_base, w[1] = t->edge_base, w[2] = t->edge_bucket_mask, w[3] = t->id_base;
                            w[4] = t->sys_node_lo, w[5] = t->sys_node_hi, w[6] = t->sys_id_lo, w[7] = t->sys_id_hi;
                            h_known = true;
                        }
                    }
                }
                for (uint32_t k = 0; k < 8; k++) L.ten[lane][k] = w[k];
                L.nr[lane] = 0;
                if (h_nlev) wbytes += L.flen[lane]; // (a deeper filter is counted by the pass that answers it)
            }
            wave_sync();
            const unsigned long long m_fresh = ballot64(fresh);
            if (fresh && rank_below(m_fresh) + 1u == count_bits(m_fresh)) { // remember the last new slot's tenant
                L.tc[0] = L.ften[lane];
                for (uint32_t k = 0; k < 8; k++) L.tc[1 + k] = L.ten[lane][k];
                L.tc[9] = h_known ? 1u : 0u;
            }
            if (fresh) {
                h_fl = RF_RANGE; // the tenant's root: the node range [0, 1)
                h_sb = 0, h_sc = (h_known && h_nlev) ? 1u : 0u;
                h_hb = h_hc = 0;
                h_lvl = 0;
                h_np = h_pcap = h_pb = 0;
                enter_level();
                if (!h_live) finish(); // no such tenant, a level the dictionary does not know, deeper than RW_LV: an empty row
            }
            wave_sync();
            reserve();
            c_tok += RW_CLK() - c_g0;
            continue;
        }
        if (m_busy == 0) break; // nothing left to take on, nothing in flight
        // ==== one round ===================================================================================================================
        const unsigned long long c_r0 = RW_CLK();
        // (1) lanes <- units: every slot its fair share, what is left to the slots that have more, in slot order
        const uint32_t rem = (home && h_live) ? h_U - h_cu : 0u;
        uint32_t s_rem[G], s_take[G];
        uint32_t left = 64;
#pragma unroll
        for (uint32_t s = 0; s < (uint32_t)G; s++) {
            s_rem[s] = rw_read_lane(rem, s);
            s_take[s] = s_rem[s] < FAIR ? s_rem[s] : FAIR;
            left -= s_take[s];
        }
        uint32_t mys = 0, myoff = 0, mytake = 0, off = 0, h_take = 0, h_off = 0;
#pragma unroll
        for (uint32_t s = 0; s < (uint32_t)G; s++) {
            const uint32_t more = s_rem[s] - s_take[s], ex = more < left ? more : left;
            s_take[s] += ex;
            left -= ex;
            if (lane >= off && s_take[s] != 0) mys = s, myoff = off, mytake = s_take[s];
            if (lane == s) h_take = s_take[s], h_off = off;
            off += s_take[s];
        }
        const bool act = lane < off;
        // (2) the unit: the slot's state comes from its home lane
        const uint32_t k_kind = __shfl(h_kind, mys), k_fl = __shfl(h_fl, mys), k_sb = __shfl(h_sb, mys), k_sc = __shfl(h_sc, mys);
        const uint32_t k_cu = __shfl(h_cu, mys);
        uint32_t k_hb = 0, k_hc = 0;
        if (ballot64(home && h_live && h_hc != 0) != 0) { // (rare: a first-level '+' of a tenant that holds '$' topics)
            k_hb = __shfl(h_hb, mys);
            k_hc = __shfl(h_hc, mys);
        }
        const uint32_t u = k_cu + (lane - myoff);
        const uint32_t par = (k_fl & RF_PAR) ? 1u : 0u;
        const bool lit = k_kind < RT_END, is_plus = k_kind == RT_PLUS, ranged = (k_fl & RF_RANGE) != 0;
        uint32_t node = 0;
        {
            // a list entry: the LDS part with an LDS read, the arena part -- only in rounds that have such lanes -- with a global one.  (Written as
            // one expression the compiler selects the ADDRESS and issues a flat load, whose wait also covers every global store in flight:
            // measured at 5.6 k ticks per round, more than the round's fetches.)
            const bool listed = act && !ranged, far = listed && u >= RW_INL;
            const uint32_t near_v = L.lst[mys][par][u < RW_INL ? u : 0u];
            uint32_t far_v = 0;
            if (ballot64(far) != 0) {
                if (far) {
                    far_v = arena[((size_t)mys * 2 + par) * r.rw_cap + (u - RW_INL)];
                    rw_settle(far_v); // (the wait for it belongs in here: placed behind the branch it would also run in rounds without such lanes)
                }
            }
            if (act) {
                if (ranged) {
                    node = k_sb + u;
                    if (k_hc && node >= k_hb) node += k_hc; // the hole: the '$' children a first-level wildcard skipped, or what descends from them
                } else node = far ? far_v : near_v;
            }
        }
        const uint32_t t_nb = L.ten[mys][0], t_eb = L.ten[mys][1], t_em = L.ten[mys][2], t_idb = L.ten[mys][3];
        // (3) the round's fetches: a bucket of the edge hash (four entries), or up to four nodes
        const uint8_t *q0, *q1, *q2, *q3;
        uint32_t bk = 0;
        if (lit) {
            bk = redge_bucket(node, k_kind, t_em);
            q0 = reinterpret_cast<const uint8_t*>(r.ix.edges + t_eb + 4 * (size_t)bk);
            q1 = q0 + 16, q2 = q0 + 32, q3 = q0 + 48;
        } else {
            const RNode* nb = r.ix.nodes + t_nb;
            q0 = reinterpret_cast<const uint8_t*>(nb + node);
            q1 = q2 = q3 = q0;
            if (is_plus && ranged) {
                q1 = reinterpret_cast<const uint8_t*>(nb + (k_sb + k_sc + k_hc - 1u)); // the range's last node
                if (k_hc) {
                    q2 = reinterpret_cast<const uint8_t*>(nb + k_hb);
                    q3 = reinterpret_cast<const uint8_t*>(nb + (k_hb + k_hc - 1u));
                }
            }
        }
        uint4 v0, v1, v2, v3;
        const unsigned long long c_l0 = RW_CLK();
        c_s0 += c_l0 - c_r0;
        rw_load4(act, q0, q1, q2, q3, v0, v1, v2, v3);
        const unsigned long long c_l1 = RW_CLK();
        c_wait += c_l1 - c_l0;
        n_rounds++;
        n_units += off;
        // (4) what the units found
        uint32_t child = NONE, cpad = 0;
        bool again = false;
        if (lit) {
            auto pick = [&](const uint4& e) {
                if (e.x == node && e.y == k_kind) child = e.z, cpad = e.w;
            };
            pick(v0), pick(v1), pick(v2), pick(v3);
            // absent from a home bucket nothing was ever pushed out of: absent.  (Full buckets are common -- a bucket holds four edges and takes
            // one on average --, buckets that overflowed are not: without the flag 36 % of the rounds ran this loop for a lane or two.)
            again = act && child == NONE && (v0.z & RE_OVERFLOW) != 0;
            child &= child == NONE ? NONE : ~RE_OVERFLOW;
        }
        if (ballot64(again) != 0) { // (cold) first-free probing continues behind an overflowed bucket (bounded by the region size)
            const unsigned long long c_c0 = RW_CLK();
            n_cold++;
            for (uint32_t probes = 1; ballot64(again) != 0; probes++) { // (bounded by the region size: not even a damaged image hangs the GPU)
                bk = (bk + 1) & t_em;
                const uint8_t* e = reinterpret_cast<const uint8_t*>(r.ix.edges + t_eb + 4 * (size_t)bk);
                uint4 x0, x1, x2, x3;
                rw_load4(again, e, e + 16, e + 32, e + 48, x0, x1, x2, x3);
                if (again) {
                    auto pick2 = [&](const uint4& x) {
                        if (x.x == node && x.y == k_kind) child = x.z & ~RE_OVERFLOW, cpad = x.w;
                    };
                    pick2(x0), pick2(x1), pick2(x2), pick2(x3);
                    again = child == NONE && x0.x != NONE && x1.x != NONE && x2.x != NONE && x3.x != NONE && probes < t_em;
                }
            }
            c_cold += RW_CLK() - c_c0;
        }
        if (act) visits += lit ? ((child != NONE && (k_fl & RF_LASTLIT)) ? 2u : 1u) // (the node a topic may end at: counted as the fetch it used to be)
                               : (is_plus ? 2u : 1u);
        const uint32_t k_nxn = __shfl(h_nxn, mys);
        const bool emitting = (k_fl & RF_EMIT) != 0;
        // ... literal level inside the filter: the children found, in frontier order, are the next list
        const bool grow = act && lit && !emitting && child != NONE;
        const unsigned long long m_grow = ballot64(grow);
        if (grow) list_put(mys, par ^ 1u, k_nxn + rank_below(rw_seg(m_grow, myoff, mytake) << myoff), child);
        if (home) h_nxn += count_bits(rw_seg(m_grow, h_off, h_take));
        // ... '+'
        const bool plus_range = act && is_plus && ranged, plus_list = act && is_plus && !ranged;
        if (ballot64(plus_range) != 0) {
            if (plus_range) { // one unit: the children of a node range are one node range; so are the children of its hole
                uint32_t cb = v0.x, ce = v1.x + (v1.y & ~RN_TERM), hb = 0, hc = 0;
                if (k_hc) hb = v2.x, hc = v3.x + (v3.y & ~RN_TERM) - v2.x;
                else if ((k_fl & RF_L0) && L.ten[mys][5] > L.ten[mys][4]) hb = L.ten[mys][4], hc = L.ten[mys][5] - L.ten[mys][4]; // the '$' children of the root
                if (hc) { // a hole at either end is no hole
                    if (hb <= cb) cb = hb + hc > cb ? hb + hc : cb, hc = 0;
                    else if (hb + hc >= ce) ce = hb, hc = 0;
                }
                if (ce < cb + hc) ce = cb + hc;
                if (hc == 0) hb = 0;
                L.mbox[mys][0] = cb, L.mbox[mys][1] = ce - cb - hc, L.mbox[mys][2] = hb, L.mbox[mys][3] = hc;
            }
            wave_sync();
        }
        if (ballot64(plus_list) != 0) { // single nodes: their children, node after node, are the next list (written by the whole wave)
            const uint32_t cb = v0.x, cc = plus_list ? (v0.y & ~RN_TERM) : 0u;
            for (unsigned long long m = ballot64(plus_list && cc != 0); m != 0; m &= m - 1ull) {
                const uint32_t l = first_bit(m), b = rw_read_lane(cb, l), c = rw_read_lane(cc, l), s = rw_read_lane(mys, l), p = rw_read_lane(par, l);
                const uint32_t at = rw_read_lane(h_nxn, s);
                for (uint32_t j = lane; j < c; j += 64) list_put(s, p ^ 1u, at + j, b + j);
                if (lane == s) h_nxn = at + c < list_cap ? at + c : list_cap + 1u;
            }
        }
        const unsigned long long c_e0 = RW_CLK();
        c_s1 += c_e0 - c_l1;
        // ... matched ranges, in frontier order
        if (ballot64(act && emitting) != 0) {
            bool pred = false, pred2 = false;
            uint32_t gb = 0, gc = 0, gb2 = 0, gc2 = 0;
            if (act && emitting) {
                if (lit) { // the filter's last level: a topic ends at the child (its rank came with the edge)
                    pred = child != NONE && (cpad & RN_TERM) != 0;
                    gb = cpad & ~RN_TERM, gc = 1;
                } else if (k_kind == RT_END) {
                    pred = (v0.y & RN_TERM) != 0;
                    gb = v0.z, gc = 1;
                } else if (k_fl & RF_L0) { // the filter "#": everything of the tenant except what lies below its '$' children
                    const uint32_t slo = L.ten[mys][6], shi = L.ten[mys][7];
                    const bool has_sys = shi > slo;
                    pred = has_sys ? slo > v0.z : v0.w > v0.z;
                    gb = v0.z, gc = (has_sys ? slo : v0.w) - v0.z;
                    pred2 = has_sys && v0.w > shi;
                    gb2 = shi, gc2 = v0.w - shi;
                } else { // "<path>/#": the node's whole subtree (its own topic included)
                    pred = v0.w > v0.z;
                    gb = v0.z, gc = v0.w - v0.z;
                }
            }
            const uint32_t k_pb = __shfl(h_pb, mys), k_np = __shfl(h_np, mys), k_pcap = __shfl(h_pcap, mys);
            const unsigned long long m1 = ballot64(pred), m2 = ballot64(pred2);
            const uint32_t seg1 = count_bits(rw_seg(m1, myoff, mytake));
            uint32_t live = 0;
            if (pred) {
                const uint32_t p = k_np + rank_below(rw_seg(m1, myoff, mytake) << myoff), g = t_idb + gb;
                if (p < k_pcap) a.pairs[k_pb + p] = MatchRange{g, gc};
                live = gc;
                if (DYN && dyn.use_dead && g < dyn.base_n) live = gc - (dead_before(dyn.dead_bits, dyn.dead_rank, g + gc) - dead_before(dyn.dead_bits, dyn.dead_rank, g));
            }
            if (pred2) {
                const uint32_t p = k_np + seg1 + rank_below(rw_seg(m2, myoff, mytake) << myoff), g = t_idb + gb2;
                if (p < k_pcap) a.pairs[k_pb + p] = MatchRange{g, gc2};
                uint32_t l2 = gc2;
                if (DYN && dyn.use_dead && g < dyn.base_n) l2 = gc2 - (dead_before(dyn.dead_bits, dyn.dead_rank, g + gc2) - dead_before(dyn.dead_bits, dyn.dead_rank, g));
                live += l2;
            }
            if (live) atomicAdd(&L.nr[mys], live);
            if (home) h_np += count_bits(rw_seg(m1, h_off, h_take)) + count_bits(rw_seg(m2, h_off, h_take));
        }
        wave_sync();
        const unsigned long long c_h0 = RW_CLK();
        c_s2 += c_h0 - c_e0;
        // (5) home lanes: a level that is used up hands over to the next one; a filter that is answered leaves its slot
        if (home && h_live) {
            h_cu += h_take;
            if (h_cu == h_U) {
                if (h_fl & RF_EMIT) h_live = false; // that was the filter's final level
                else {
                    if (h_kind == RT_PLUS && (h_fl & RF_RANGE)) {
                        h_sb = L.mbox[lane][0], h_sc = L.mbox[lane][1], h_hb = L.mbox[lane][2], h_hc = L.mbox[lane][3];
                    } else {
                        h_fl = (h_fl & ~RF_RANGE) ^ RF_PAR;
                        h_sb = 0, h_hb = h_hc = 0;
                        h_sc = h_nxn < list_cap ? h_nxn : list_cap;
                        if (h_nxn > list_cap) ovf = true;
                    }
                    h_lvl++;
                    enter_level();
                }
                if (!h_live) finish();
            }
        }
        const unsigned long long c_v0 = RW_CLK();
        c_s3 += c_v0 - c_h0;
        reserve();
        c_s4 += RW_CLK() - c_v0;
    }
    if (ballot64(ovf) != 0 && lane == 0) atomicOr(&a.ctr->status, ST_RETAIN_LIST);
#if BMQ_RW_CLOCKS && !defined(BMQ_WAVE_EMU)
    if (lane == 0) {
        const unsigned long long c_all = RW_CLK() - c_start;
        atomicAdd(&g_rw_clk[0], 1ull), atomicAdd(&g_rw_clk[1], n_grp), atomicAdd(&g_rw_clk[2], n_rounds), atomicAdd(&g_rw_clk[3], n_units), atomicAdd(&g_rw_clk[4], n_cold);
        atomicAdd(&g_rw_clk[5], c_all), atomicAdd(&g_rw_clk[6], c_tok), atomicAdd(&g_rw_clk[7], c_wait), atomicAdd(&g_rw_clk[8], c_cold);
        atomicMax(&g_rw_clk[9], c_all);
        atomicAdd(&g_rw_clk[10], c_s0), atomicAdd(&g_rw_clk[11], c_s1), atomicAdd(&g_rw_clk[12], c_s2), atomicAdd(&g_rw_clk[13], c_s3), atomicAdd(&g_rw_clk[14], c_s4);
    }
#endif
    const unsigned long long wv = wave_total_u64((unsigned long long)visits), wr = wave_total_u64(wranges), wb = wave_total_u64(wbytes);
    if (lane == 0) { // once per persistent wave
        if (wv) atomicAdd(&a.ctr->n_visit, wv);
        if (wr) atomicAdd(&a.ctr->n_ranges, wr);
        if (wb) atomicAdd(&a.ctr->topic_bytes, wb);
    }
}

} // namespace bmq
