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
constexpr uint32_t RT_GP = 0xFFFFFFFAu;     // a literal level over ALL children of one node, answered by one look-up in the (grandparent, token) hash
constexpr uint32_t RT_PCOPY = 0xFFFFFFF9u;  // ... the postings slice it found becomes the next list
constexpr uint32_t RT_PEMIT = 0xFFFFFFF8u;  // ... or, behind the filter's last level, the matched topics
constexpr uint32_t RT_LIT_MAX = 0xFFFFFFF0u; // level kinds from here on are not dictionary tokens
// slot flags
constexpr uint32_t RF_RANGE = 1u;   // the frontier is a node range (else: a list)
constexpr uint32_t RF_PAR = 2u;     // which of the slot's two lists is the CURRENT one
constexpr uint32_t RF_LASTLIT = 4u; // this level is the filter's last and a literal: found children emit
constexpr uint32_t RF_L0 = 8u;      // this is the filter's first level (the frontier is the tenant's root)
constexpr uint32_t RF_EMIT = 16u;   // this level emits (room in `pairs` is reserved)
constexpr uint32_t RF_ONEP = 32u;   // the range holds ALL children of one node (h_gpn): postings and merged subtrees apply
constexpr uint32_t RF_MERGE = 64u;  // "<path>/+/#" over such a range: the children's subtrees are one id range (two around a hole)
constexpr uint32_t RF_SYSX = 128u;  // ... and they are the tenant root's: the '$' children are not part of it (a first-level wildcard skips them)
constexpr uint32_t RF_LVL_SHIFT = 8; // bits 8-12: the level

template <int G> struct RwLds {
    static_assert(G >= 2 && G <= 8 && (G & (G - 1)) == 0, "slots per wave: G / 2 new filters x 16 level lanes fill the wave's 64 lanes");
    uint32_t tok[G][RW_LV];     // level kinds / dictionary tokens of the group's filters
    uint32_t lst[G][2][RW_INL]; // the first RW_INL entries of every list
    uint32_t stg[RW_STAGE / 4]; // tokeniser: the staged bytes of the filters taken on
    uint32_t ten[G][12];        // the slot's tenant: node_base, edge_base, edge_bucket_mask, id_base, sys_node_lo, sys_node_hi, sys_id_lo, sys_id_hi, post_base, gp_base, gp_bucket_mask
    uint32_t mbox[G][8];        // what a '+' unit over a RANGE leaves for the slot's home lane: child range (begin, count), hole (begin, count), the one parent or NONE | a (grandparent, token) look-up: slice (begin, count)
    uint32_t nr[G];             // ids matched (added up by the emitting lanes)
    uint32_t nlev[G];           // levels (0: nothing to walk)
    uint32_t fid[G];            // the filter (row of the batch) a refill put into the slot
    uint32_t ften[G], flen[G];  // ... its tenant (index in the batch's tenant table), its bytes
    uint32_t tc[14];            // the tenant resolved last: index in the batch's tenant table, its 11 words, known
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
// ... the same plus one whole 64-byte line (the bulk chunk's bucket), every lane: eight requests, one wait
__device__ __forceinline__ void rw_load8(const void* q0, const void* q1, const void* q2, const void* q3, const void* b, uint4& v0, uint4& v1, uint4& v2, uint4& v3,
                                         uint4& w0, uint4& w1, uint4& w2, uint4& w3) {
    asm volatile("global_load_dwordx4 %0, %8, off\n\t"
                 "global_load_dwordx4 %1, %9, off\n\t"
                 "global_load_dwordx4 %2, %10, off\n\t"
                 "global_load_dwordx4 %3, %11, off\n\t"
                 "global_load_dwordx4 %4, %12, off\n\t"
                 "global_load_dwordx4 %5, %12, off offset:16\n\t"
                 "global_load_dwordx4 %6, %12, off offset:32\n\t"
                 "global_load_dwordx4 %7, %12, off offset:48\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                 : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(b)
                 : "memory");
}
__device__ __forceinline__ uint32_t rw_ctz(uint32_t v) { return (uint32_t)__builtin_ctz(v); }
// a use of v that stays where it is written: the compiler's wait for the load that produces v lands in front of it
__device__ __forceinline__ void rw_settle(uint32_t& v) { asm volatile("" : "+v"(v)); }
#define RW_COVER(i) ((void)0)
#else
// the emulator's harness checks that its cases reach the paths that are easy to miss: [0] postings look-ups, [1] slices copied, [2] slices
// emitted, [3] merged subtrees, [4] bulk chunks, [5] list entries read from the arena, [6] probes behind an overflowed bucket, [7] '+' over a list
inline unsigned long long g_rw_cover[8];
#define RW_COVER(i) (g_rw_cover[i]++)
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
inline void rw_load8(const void* q0, const void* q1, const void* q2, const void* q3, const void* b, uint4& v0, uint4& v1, uint4& v2, uint4& v3, uint4& w0, uint4& w1,
                     uint4& w2, uint4& w3) {
    v0 = *reinterpret_cast<const uint4*>(q0), v1 = *reinterpret_cast<const uint4*>(q1), v2 = *reinterpret_cast<const uint4*>(q2), v3 = *reinterpret_cast<const uint4*>(q3);
    const uint4* l = reinterpret_cast<const uint4*>(b);
    w0 = l[0], w1 = l[1], w2 = l[2], w3 = l[3];
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
__device__ unsigned long long g_rw_clk[24];
__device__ uint4 g_rw_wave[16384]; // per wave: ticks, quads taken, rounds, units | start tick (low 32 bits) in w
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
    // Round 6: a wave whose partition has run dry goes on with a NEIGHBOUR's (the partitions part ^ 1, ^ 2, ^ 4 ... in turn: at most six
    // more counters per wave, a few hundred requests per counter at the end of a launch) instead of leaving -- the partitions hold equal
    // numbers of quads, not equal amounts of work, and the waves of a dry partition used to sit out the end of the launch.
    const uint32_t n_quads = (r.n_filters + RQ - 1) / RQ, n_parts = gridDim.x < RW_PARTS ? gridDim.x : RW_PARTS, part0 = blockIdx.x % n_parts;
    uint32_t part = part0, steal = 0; // (wave-uniform)
    uint32_t visits = 0;                        // per lane
    unsigned long long wranges = 0, wbytes = 0; // home lanes
    bool ovf = false;
    uint32_t ck_next = 0, ck_end = 0; // the wave's chunk of `pairs` (wave-uniform)
    if (lane == 0) L.tc[0] = NONE;
    wave_sync();
    unsigned long long c_tok = 0, c_wait = 0, c_cold = 0, n_rounds = 0, n_units = 0, n_cold = 0, n_grp = 0;
    unsigned long long c_s0 = 0, c_s1 = 0, c_s2 = 0, c_s3 = 0, c_s4 = 0; // round segments: hand-out + unit | found + '+' | emission | hand-over | reserve
    (void)c_s0, (void)c_s1, (void)c_s2, (void)c_s3, (void)c_s4;
    unsigned long long c_h[4] = {0, 0, 0, 0}; // inside the hand-out segment: bulk set-up | lanes <- units | slot state | entry + tenant + addresses
    (void)c_h;
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
    uint32_t h_gpn = 0;                      // RF_ONEP: the node whose children the range holds
    uint32_t h_nxn = 0;                      // entries of the NEXT list so far
    uint32_t h_pb = 0, h_np = 0, h_pcap = 0; // room reserved in `pairs`, ranges written
    bool h_live = false;                     // the slot has units left
    uint32_t h_need = 0;                     // room in `pairs` this slot asks for
    // the level the slot is at: its kind, units, flags; h_need = room for its ranges if it emits
    auto enter_level = [&]() {
        h_kind = h_lvl < h_nlev ? L.tok[lane][h_lvl] : RT_END;
        h_fl = (h_fl & (RF_RANGE | RF_PAR | RF_ONEP | RF_SYSX)) | (h_lvl << RF_LVL_SHIFT);
        if (h_lvl == 0) h_fl |= RF_L0;
        const bool lit = h_kind < RT_LIT_MAX;
        if (lit && h_lvl + 1 == h_nlev) h_fl |= RF_LASTLIT;
        if (h_kind == RT_PLUS && !(h_fl & RF_RANGE) && h_sc == 1) { // a list of one node under '+': as good as a range of one
            h_sb = L.lst[lane][(h_fl & RF_PAR) ? 1 : 0][0];
            h_fl |= RF_RANGE;
            h_fl &= ~(RF_ONEP | RF_SYSX);
            h_hb = h_hc = 0;
        }
        h_U = (h_kind == RT_PLUS && (h_fl & RF_RANGE)) ? (h_sc ? 1u : 0u) : h_sc;
        const bool onep = (h_fl & (RF_RANGE | RF_ONEP)) == (RF_RANGE | RF_ONEP) && h_sc != 0;
        if (onep && h_kind == RT_HASH && h_lvl != 0) h_fl |= RF_MERGE, h_U = 1; // the children's subtrees: one id range
        else if (onep && lit && h_kind != TOK_UNKNOWN && h_sc + h_hc >= RPOST_MIN) h_kind = RT_GP, h_U = 1; // one look-up instead of one per child
        h_cu = 0;
        h_nxn = 0;
        h_live = h_U != 0 && h_kind != TOK_UNKNOWN;
        if (h_live && h_kind != RT_GP && (h_kind == RT_HASH || h_kind == RT_END || (h_fl & RF_LASTLIT))) {
            h_fl |= RF_EMIT;
            h_need = h_U + 2u; // (the filter "#" and a merged range around a hole emit two ranges from their one unit)
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
                // this partition is dry: the next neighbour that exists (power-of-two distances), or done
                do steal = steal ? steal << 1 : 1u;
                while (steal < RW_PARTS && (part0 ^ steal) >= n_parts);
                if (steal >= RW_PARTS) exhausted = true;
                else {
                    part = part0 ^ steal;
                    request_quad();
                }
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
                uint32_t w[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (ti < r.n_tenants) {
                    if (ti == L.tc[0]) {
                        for (uint32_t k = 0; k < 11; k++) w[k] = L.tc[1 + k];
                        h_known = L.tc[12] != 0;
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
                            w[8] = t->post_base, w[9] = t->gp_base, w[10] = t->gp_bucket_mask;
                            h_known = true;
                        }
                    }
                }
                for (uint32_t k = 0; k < 11; k++) L.ten[lane][k] = w[k];
                L.nr[lane] = 0;
                if (h_nlev) wbytes += L.flen[lane]; // (a deeper filter is counted by the pass that answers it)
            }
            wave_sync();
            const unsigned long long m_fresh = ballot64(fresh);
            if (fresh && rank_below(m_fresh) + 1u == count_bits(m_fresh)) { // remember the last new slot's tenant
                L.tc[0] = L.ften[lane];
                for (uint32_t k = 0; k < 11; k++) L.tc[1 + k] = L.ten[lane][k];
                L.tc[12] = h_known ? 1u : 0u;
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
        // (0) a slot with a whole wave of literal look-ups left gets a BULK chunk -- 64 consecutive units of its frontier, one per lane, the slot's
        // state wave-uniform: no hand-out, no per-lane slot look-ups -- NEXT TO the round's mixed units (the same wait serves both)
        const uint32_t rem_all = (home && h_live) ? h_U - h_cu : 0u;
        const unsigned long long m_bulk = ballot64(rem_all >= 64u && h_kind < RT_LIT_MAX && L.ten[home ? lane : 0u][2] < (1u << 26));
        const bool has_bulk = m_bulk != 0;
        const uint32_t bs = has_bulk ? first_bit(m_bulk) : 0u;
        uint32_t b_kind = 0, b_fl = 0, b_node = 0, b_bk = 0, b_eb = 0, b_em = 0;
        const uint8_t* bq = reinterpret_cast<const uint8_t*>(r.ix.edges);
        if (has_bulk) {
            b_kind = rw_read_lane(h_kind, bs), b_fl = rw_read_lane(h_fl, bs);
            const uint32_t bu = rw_read_lane(h_cu, bs) + lane, b_par = (b_fl & RF_PAR) ? 1u : 0u;
            if (b_fl & RF_RANGE) {
                const uint32_t hb = rw_read_lane(h_hb, bs), hc = rw_read_lane(h_hc, bs);
                b_node = rw_read_lane(h_sb, bs) + bu;
                if (hc && b_node >= hb) b_node += hc;
            } else {
                const bool far = bu >= RW_INL;
                const uint32_t near_v = L.lst[bs][b_par][far ? 0u : bu];
                uint32_t far_v = 0;
                if (ballot64(far) != 0) {
                    if (far) {
                        far_v = arena[((size_t)bs * 2 + b_par) * r.rw_cap + (bu - RW_INL)];
                        rw_settle(far_v);
                    }
                }
                b_node = far ? far_v : near_v;
            }
            b_eb = L.ten[bs][1], b_em = L.ten[bs][2];
            b_bk = redge_bucket(b_node, b_kind, b_em);
            bq = reinterpret_cast<const uint8_t*>(r.ix.edges + b_eb + 4 * (size_t)b_bk);
        }
        // (1) lanes <- units: every slot its fair share, what is left to the slots that have more, in slot order.  A round whose work is all in
        // its bulk chunk skips the mixed part altogether: the rounds a heavy filter is walked alone in, at the end of a launch, cost a fraction
        const uint32_t rem = (has_bulk && lane == bs) ? 0u : rem_all;
        const bool has_mixed = ballot64(rem != 0) != 0;
        uint32_t mys = 0, myoff = 0, mytake = 0, off = 0, h_take = 0, h_off = 0;
        uint32_t k_kind = RT_END, k_fl = 0, k_sb = 0, k_sc = 0, k_hb = 0, k_hc = 0, par = 0, node = 0, bk = 0, k_gpn = 0, k_tok = 0;
        uint32_t t_eb = 0, t_em = 0, t_idb = 0;
        bool act = false, lit = false, is_plus = false, ranged = false;
        const uint8_t *q0 = bq, *q1 = bq, *q2 = bq, *q3 = bq;
        if (has_mixed) {
            uint32_t s_rem[G], s_take[G];
            uint32_t left = 64;
#pragma unroll
            for (uint32_t s = 0; s < (uint32_t)G; s++) {
                s_rem[s] = rw_read_lane(rem, s);
                s_take[s] = s_rem[s] < FAIR ? s_rem[s] : FAIR;
                left -= s_take[s];
            }
#pragma unroll
            for (uint32_t s = 0; s < (uint32_t)G; s++) {
                const uint32_t more = s_rem[s] - s_take[s], ex = more < left ? more : left;
                s_take[s] += ex;
                left -= ex;
                if (lane >= off && s_take[s] != 0) mys = s, myoff = off, mytake = s_take[s];
                if (lane == s) h_take = s_take[s], h_off = off;
                off += s_take[s];
            }
            act = lane < off;
            // (2) the unit: the slot's state comes from its home lane
            k_kind = __shfl(h_kind, mys), k_fl = __shfl(h_fl, mys), k_sb = __shfl(h_sb, mys), k_sc = __shfl(h_sc, mys);
            const uint32_t k_cu = __shfl(h_cu, mys);
            if (ballot64(home && h_live && h_hc != 0) != 0) { // (rare: a first-level '+' of a tenant that holds '$' topics)
                k_hb = __shfl(h_hb, mys);
                k_hc = __shfl(h_hc, mys);
            }
            const uint32_t u = k_cu + (lane - myoff);
            par = (k_fl & RF_PAR) ? 1u : 0u;
            lit = k_kind < RT_LIT_MAX, is_plus = k_kind == RT_PLUS, ranged = (k_fl & RF_RANGE) != 0;
            const bool posted = k_kind == RT_PCOPY || k_kind == RT_PEMIT;
            if (ballot64(k_kind == RT_GP) != 0) {
                k_gpn = __shfl(h_gpn, mys);
                k_tok = L.tok[mys][(k_fl >> RF_LVL_SHIFT) & 31u];
            }
            {
                // a list entry: the LDS part with an LDS read, the arena part -- only in rounds that have such lanes -- with a global one.  (Written as
                // one expression the compiler selects the ADDRESS and issues a flat load, whose wait also covers every global store in flight.)
                const bool listed = act && !ranged && !posted && k_kind != RT_GP, far = listed && u >= RW_INL;
                const uint32_t near_v = L.lst[mys][par][u < RW_INL ? u : 0u];
                uint32_t far_v = 0;
                if (ballot64(far) != 0) {
                    if (far) {
                        RW_COVER(5);
                        far_v = arena[((size_t)mys * 2 + par) * r.rw_cap + (u - RW_INL)];
                        rw_settle(far_v); // (the wait for it belongs in here: placed behind the branch it would also run in rounds without such lanes)
                    }
                }
                if (act) {
                    if (posted) node = k_sb + u; // (an entry of the postings slice)
                    else if (ranged) {
                        node = k_sb + u;
                        if (k_hc && node >= k_hb) node += k_hc; // the hole: the '$' children a first-level wildcard skipped, or what descends from them
                    } else node = far ? far_v : near_v;
                }
            }
            const uint32_t t_nb = L.ten[mys][0];
            t_eb = L.ten[mys][1], t_em = L.ten[mys][2], t_idb = L.ten[mys][3];
            // (3) the round's fetches: a bucket of the edge hash (four entries), or up to four nodes
            if (lit) {
                bk = redge_bucket(node, k_kind, t_em);
                q0 = reinterpret_cast<const uint8_t*>(r.ix.edges + t_eb + 4 * (size_t)bk);
                q1 = q0 + 16, q2 = q0 + 32, q3 = q0 + 48;
            } else if (k_kind == RT_GP) { // the home bucket of (the range's one parent, the level's token)
                bk = rgp_bucket(k_gpn, k_tok, L.ten[mys][10]);
                q0 = reinterpret_cast<const uint8_t*>(r.ix.gps + L.ten[mys][9] + 4 * (size_t)bk);
                q1 = q0 + 16, q2 = q0 + 32, q3 = q0 + 48;
            } else if (posted) {
                q0 = reinterpret_cast<const uint8_t*>(r.ix.posts + L.ten[mys][8] + node);
                q1 = q2 = q3 = q0;
            } else {
                const RNode* nb = r.ix.nodes + t_nb;
                q0 = reinterpret_cast<const uint8_t*>(nb + node);
                q1 = q2 = q3 = q0;
                if ((is_plus || (k_fl & RF_MERGE)) && ranged) {
                    q1 = reinterpret_cast<const uint8_t*>(nb + (k_sb + k_sc + k_hc - 1u)); // the range's last node
                    if (k_hc) {
                        q2 = reinterpret_cast<const uint8_t*>(nb + k_hb);
                        q3 = reinterpret_cast<const uint8_t*>(nb + (k_hb + k_hc - 1u));
                    }
                }
            }
        }
        uint4 v0, v1, v2, v3;
        const unsigned long long c_l0 = RW_CLK();
        c_s0 += c_l0 - c_r0;
        uint4 w0, w1, w2, w3;
        if (has_bulk && has_mixed) {
            if (!act) q0 = q1 = q2 = q3 = bq; // (a lane without a mixed unit asks for its bulk line once more)
            rw_load8(q0, q1, q2, q3, bq, v0, v1, v2, v3, w0, w1, w2, w3);
        } else if (has_bulk) {
            rw_load4(true, bq, bq + 16, bq + 32, bq + 48, w0, w1, w2, w3);
            v0 = v1 = v2 = v3 = make_uint4(0u, 0u, 0u, 0u);
        } else {
            rw_load4(act, q0, q1, q2, q3, v0, v1, v2, v3);
            w0 = w1 = w2 = w3 = make_uint4(0u, 0u, 0u, 0u);
        }
        const unsigned long long c_l1 = RW_CLK();
        c_wait += c_l1 - c_l0;
        n_rounds++;
        n_units += off;
        // (cold) first-free probing continues behind an overflowed bucket (bounded by the region size: not even a damaged image hangs the GPU)
        auto probe_on = [&](bool more, uint32_t p_node, uint32_t p_tok, uint32_t p_bk, uint32_t p_eb, uint32_t p_em, uint32_t& p_child, uint32_t& p_pad) {
            if (ballot64(more) == 0) return;
            const unsigned long long c_c0 = RW_CLK();
            n_cold++;
            if (more) RW_COVER(6);
            for (uint32_t probes = 1; ballot64(more) != 0; probes++) {
                p_bk = (p_bk + 1) & p_em;
                const uint8_t* e = reinterpret_cast<const uint8_t*>(r.ix.edges + p_eb + 4 * (size_t)p_bk);
                uint4 x0, x1, x2, x3;
                rw_load4(more, e, e + 16, e + 32, e + 48, x0, x1, x2, x3);
                if (more) {
                    auto pick2 = [&](const uint4& x) {
                        if (x.x == p_node && x.y == p_tok) p_child = x.z & ~RE_OVERFLOW, p_pad = x.w;
                    };
                    pick2(x0), pick2(x1), pick2(x2), pick2(x3);
                    more = p_child == NONE && x0.x != NONE && x1.x != NONE && x2.x != NONE && x3.x != NONE && probes < p_em;
                }
            }
            c_cold += RW_CLK() - c_c0;
        };
        if (has_mixed) {
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
            const bool posted = k_kind == RT_PCOPY || k_kind == RT_PEMIT;
            if (posted && act) { // an entry of the slice: (parent, token, child, child_topic); a parent inside the hole was skipped by the wildcard
                const bool keep = !((k_fl & RF_SYSX) && v0.x >= L.ten[mys][4] && v0.x < L.ten[mys][5]);
                child = keep ? v0.z : NONE;
                cpad = v0.w;
                RW_COVER(k_kind == RT_PCOPY ? 1 : 2);
            }
            if (ballot64(act && k_kind == RT_GP) != 0) { // the slice of the postings that answers the level
                if (act && k_kind == RT_GP) {
                    RW_COVER(0);
                    uint32_t p_begin = 0, p_count = 0;
                    bool hit = false;
                    auto pickg = [&](const uint4& e) {
                        if (e.x == k_gpn && e.y == k_tok) p_begin = e.z, p_count = e.w & ~RE_OVERFLOW, hit = true;
                    };
                    pickg(v0), pickg(v1), pickg(v2), pickg(v3);
                    bool more = !hit && (v0.w & RE_OVERFLOW) != 0;
                    const uint32_t gmask = L.ten[mys][10];
                    for (uint32_t probes = 1; more && probes <= gmask; probes++) { // (cold) behind an overflowed bucket
                        bk = (bk + 1) & gmask;
                        const RGp* e = r.ix.gps + L.ten[mys][9] + 4 * (size_t)bk;
                        bool free_slot = false;
                        for (uint32_t j = 0; j < 4; j++) {
                            const uint4 x = *reinterpret_cast<const uint4*>(e + j);
                            if (x.x == k_gpn && x.y == k_tok) p_begin = x.z, p_count = x.w & ~RE_OVERFLOW, hit = true;
                            free_slot = free_slot || x.x == NONE;
                        }
                        more = !hit && !free_slot;
                    }
                    L.mbox[mys][5] = p_begin, L.mbox[mys][6] = p_count;
                }
                wave_sync();
            }
            probe_on(again, node, k_kind, bk, t_eb, t_em, child, cpad);
            // nodes the walk of the reference touches (TopicLevelTrie.lookup: every child under a '+'), whatever this kernel fetched for them
            if (act) visits += lit ? ((child != NONE && (k_fl & RF_LASTLIT)) ? 2u : 1u) // (the node a topic may end at: counted as the fetch it used to be)
                               : k_kind == RT_GP || (k_fl & RF_MERGE) ? k_sc
                               : posted ? ((k_kind == RT_PEMIT && child != NONE) ? 1u : 0u)
                               : (is_plus ? 2u : 1u);
            const uint32_t k_nxn = __shfl(h_nxn, mys);
            const bool emitting = (k_fl & RF_EMIT) != 0;
            // ... literal level inside the filter: the children found, in frontier order, are the next list
            const bool grow = act && (lit || k_kind == RT_PCOPY) && !emitting && child != NONE;
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
                    L.mbox[mys][4] = (k_sc == 1 && k_hc == 0) ? k_sb : NONE; // all children of ONE node: postings and merged subtrees apply
                }
                wave_sync();
            }
            if (ballot64(plus_list) != 0) { // single nodes: their children, node after node, are the next list (written by the whole wave)
                const uint32_t cb = v0.x, cc = plus_list ? (v0.y & ~RN_TERM) : 0u;
                if (plus_list) RW_COVER(7);
                for (unsigned long long m = ballot64(plus_list && cc != 0); m != 0; m &= m - 1ull) {
                    const uint32_t l = first_bit(m), b = rw_read_lane(cb, l), c = rw_read_lane(cc, l), s = rw_read_lane(mys, l), p = rw_read_lane(par, l);
                    const uint32_t at = rw_read_lane(h_nxn, s);
                    for (uint32_t j = lane; j < c; j += 64) list_put(s, p ^ 1u, at + j, b + j);
                    if (lane == s) h_nxn = at + c < list_cap ? at + c : list_cap + 1u;
                }
            }
            // ... matched ranges, in frontier order
            if (ballot64(act && emitting) != 0) {
                bool pred = false, pred2 = false;
                uint32_t gb = 0, gc = 0, gb2 = 0, gc2 = 0;
                if (act && emitting) {
                    if (lit || k_kind == RT_PEMIT) { // the filter's last level: a topic ends at the child (its rank came with the edge)
                        pred = child != NONE && (cpad & RN_TERM) != 0;
                        gb = cpad & ~RN_TERM, gc = 1;
                    } else if (k_kind == RT_END) {
                        pred = (v0.y & RN_TERM) != 0;
                        gb = v0.z, gc = 1;
                    } else if (k_fl & RF_MERGE) { RW_COVER(3); // "<path>/+/#": the subtrees of ALL children of one node lie side by side (v0, v1: first and last child; v2, v3: the hole's)
                        const uint32_t lo = v0.z, hi = v1.w;
                        if (k_hc) {
                            pred = v2.z > lo, gb = lo, gc = v2.z - lo;
                            pred2 = hi > v3.w, gb2 = v3.w, gc2 = hi - v3.w;
                        } else pred = hi > lo, gb = lo, gc = hi - lo;
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
        }
        const unsigned long long c_e0 = RW_CLK();
        c_s1 += c_e0 - c_l1;
        // ... the bulk chunk: the same look-up, the slot wave-uniform
        if (has_bulk) {
            uint32_t b_child = NONE, b_pad = 0;
            auto pickb = [&](const uint4& e) {
                if (e.x == b_node && e.y == b_kind) b_child = e.z, b_pad = e.w;
            };
            pickb(w0), pickb(w1), pickb(w2), pickb(w3);
            const bool b_more = b_child == NONE && (w0.z & RE_OVERFLOW) != 0;
            b_child &= b_child == NONE ? NONE : ~RE_OVERFLOW;
            probe_on(b_more, b_node, b_kind, b_bk, b_eb, b_em, b_child, b_pad);
            const bool b_found = b_child != NONE;
            visits += (b_found && (b_fl & RF_LASTLIT)) ? 2u : 1u;
            if (!(b_fl & RF_EMIT)) { // children found, in frontier order, go to the slot's next list
                const unsigned long long m = ballot64(b_found);
                const uint32_t at = rw_read_lane(h_nxn, bs);
                if (b_found) list_put(bs, ((b_fl & RF_PAR) ? 1u : 0u) ^ 1u, at + rank_below(m), b_child);
                if (lane == bs) h_nxn = at + count_bits(m);
            } else { // the filter's last level: the topics that end at the children found
                const bool pred = b_found && (b_pad & RN_TERM) != 0;
                const unsigned long long m = ballot64(pred);
                const uint32_t pb = rw_read_lane(h_pb, bs), np = rw_read_lane(h_np, bs), pcap = rw_read_lane(h_pcap, bs);
                bool alive = pred;
                if (pred) {
                    const uint32_t p = np + rank_below(m), g = L.ten[bs][3] + (b_pad & ~RN_TERM);
                    if (p < pcap) a.pairs[pb + p] = MatchRange{g, 1u};
                    if (DYN && dyn.use_dead && g < dyn.base_n) alive = !id_dead(dyn.dead_bits, g);
                }
                const unsigned long long m_alive = ballot64(alive);
                if (lane == bs) {
                    h_np = np + count_bits(m);
                    L.nr[bs] += count_bits(m_alive);
                }
            }
            if (lane == bs) h_cu += 64u, RW_COVER(4);
            n_units += 64;
        }
        wave_sync();
        const unsigned long long c_h0 = RW_CLK();
        c_s2 += c_h0 - c_e0;
        // (5) home lanes: a level that is used up hands over to the next one; a filter that is answered leaves its slot
        if (home && h_live) {
            h_cu += h_take;
            if (h_cu == h_U) {
                if (h_fl & RF_EMIT) h_live = false; // that was the filter's final level
                else if (h_kind == RT_GP) { // the slice is the level's answer: its topics (last level) or the next list
                    h_sb = L.mbox[lane][5], h_sc = h_U = L.mbox[lane][6];
                    h_cu = 0, h_nxn = 0;
                    h_fl &= ~(RF_RANGE | RF_ONEP); // (RF_SYSX stays: the slice holds the children of the '$' nodes too)
                    h_kind = (h_fl & RF_LASTLIT) ? RT_PEMIT : RT_PCOPY;
                    h_live = h_U != 0;
                    if (h_live && h_kind == RT_PEMIT) h_fl |= RF_EMIT, h_need = h_U + 2u;
                } else {
                    if (h_kind == RT_PLUS && (h_fl & RF_RANGE)) {
                        h_sb = L.mbox[lane][0], h_sc = L.mbox[lane][1], h_hb = L.mbox[lane][2], h_hc = L.mbox[lane][3];
                        h_gpn = L.mbox[lane][4];
                        h_fl &= ~(RF_ONEP | RF_SYSX);
                        if (h_gpn != NONE) h_fl |= (h_fl & RF_L0) ? (RF_ONEP | RF_SYSX) : RF_ONEP;
                    } else {
                        h_fl = (h_fl & ~(RF_RANGE | RF_ONEP | RF_SYSX)) ^ RF_PAR;
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
        if (blockIdx.x < 16384) g_rw_wave[blockIdx.x] = make_uint4((uint32_t)c_all, (uint32_t)n_grp | ((uint32_t)n_rounds << 16), (uint32_t)n_units, (uint32_t)(c_start >> 4));
        atomicAdd(&g_rw_clk[15], c_h[0]), atomicAdd(&g_rw_clk[16], c_h[1]), atomicAdd(&g_rw_clk[17], c_h[2]), atomicAdd(&g_rw_clk[18], c_h[3]);
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
