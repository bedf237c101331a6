// bmq_walk_kernel.h -- k_walk<LV, QC, PC, MIXED>: the dominant kernel of the dist direction (included by bmq_dist_kernels.h).
//
// One wave (= one 64-thread workgroup) per 64 topics:
//   phase 1  the wave stages its topics' bytes in LDS (coalesced 16-byte loads); every lane scans its own topic level by level and
//            all lanes look their level up in the dictionary together (one 64-byte line per level);
//   phase 2  the wave drains a depth-first LDS work STACK of (node id, topic, level, kind) items, one item per lane per round; an
//            item costs exactly one aligned 64-byte line (the home bucket of its edge, which holds the child's whole slot);
//            pushes and matched ranges are compacted with ballot + mbcnt;
//   phase 3  the matched (begin, count) ranges are counting-sorted by topic and written out; per-wave id count + its contribution
//            to the sum of its 256-wave super-block (k_expand derives the row pointers from them).
//
// Round 4 rewrite.  The SQ counters of the round-3 kernel (profiles/r04/c3_pmc_sq.csv) say what bounded it: its waves spent 68 % of
// their cycles parked on s_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES), 25 % issuing, the VALU pipes were 45 % busy -- a latency-bound
// kernel that ran 4 waves per SIMD because it held 111 VGPRs, 106 SGPRs (9 spilled) and 9.1 KB of LDS per wave.  This version is
// built for occupancy -- 8 waves per SIMD need <= 64 VGPRs, <= 80 SGPRs, no scratch and <= 5 KB of LDS per one-wave workgroup:
//   * the LDS geometry is a compile-time constant of the instantiation (no run-time address arithmetic, no registers for it).  The
//     token table is RAGGED: a topic's tokens sit one after the other behind those of the topic before it (a count of the '/' bytes
//     comes first), so 2 KB hold the 64 topics of a wave whatever their depths add up to (5.5 levels on average in the survey's
//     workload, 9 at most -- a [16 levels][64 topics] matrix took 4 KB, one of 8 levels left the 9-level topics out); a wave whose
//     topics need more than the table holds walks them in several chunks; a stack item is one 64-bit LDS word that carries the
//     position of its level's token and the number of levels behind it, a matched range a 64-bit word + a byte;
//   * everything the wave's lanes agree on (list fill levels, spill chains, the tenant's region) is kept wave-uniform EXPLICITLY
//     (readfirstlane / ballot builtins): the compiler then holds it in scalar registers and branches on the scalar unit -- the round-3
//     kernel's list bookkeeping lived in vector registers behind exec-mask juggling;
//   * a wave walks ONE tenant at a time with the tenant's region in scalar registers; the few waves of a tenant-grouped batch that
//     straddle tenants walk them one after the other.  A batch whose waves hold many tenants each (not grouped by tenant) is run again
//     through the MIXED instantiation, which keeps (base, buckets) per topic in LDS (ST_WANT_MIXED; bmq_engine.hip remembers).
#pragma once

namespace bmq {

constexpr uint32_t WALK_MAX_TENANTS = 4; // distinct tenants a wave of the grouped instantiations walks one after the other
// stack item = (node id, meta); meta: bits 0-5 topic-local index, 6-15 position of the level's token in the table, 16-20 levels of the
// topic BEHIND this one, 31 kind (KIND_P: '+' probe).  The child of an item: the same topic, the next token, one level fewer behind.
__device__ __forceinline__ uint32_t walk_meta(uint32_t tl, uint32_t tokpos, uint32_t rem) { return tl | (tokpos << 6) | (rem << 16); }
constexpr uint32_t WALK_META_CHILD = 64u - 65536u; // (wraps: + 1 token position, - 1 level behind)

// The instantiations the engine launches: the default geometry and the smallest lists (tests: every overflow path runs all the time)
#define BMQ_WALK_GEOM_DEFAULT 512, 176, 152
#define BMQ_WALK_GEOM_SMALLEST 192, 128, 128
constexpr uint32_t WALK_QC_DEFAULT = 176, WALK_PC_DEFAULT = 152, WALK_QC_SMALLEST = 128, WALK_PC_SMALLEST = 128;

// TC: token table, entries (<= 1023); QC: work stack, items; PC: matched-range buffer, entries (both >= 128: one sink pushes / emits up to 128)
template <int TC, int QC, int PC, bool MIXED> struct WalkLds { // byte offsets into the wave's LDS
    static constexpr uint32_t TOK = 0;                    // u32 [TC]       tokens of the wave's topics, topic after topic | phase 3: cnt_pairs, cnt_routes, cursor
    static constexpr uint32_t STK = TOK + TC * 4;         // uint2 [QC]     work stack (node id, meta)     | phase 1: staged topic bytes from here on
    static constexpr uint32_t PRG = STK + QC * 8;         // uint2 [PC]     matched ranges (begin, count)
    static constexpr uint32_t PTP = PRG + PC * 8;         // u8 [PC]        ... topic-local index
    static constexpr uint32_t VIS = PTP + PC;             // u32 [64]       nodes discovered per topic (N_visit of SURVEY.md 8d, per row)
    static constexpr uint32_t REG = VIS + 64 * 4;         // uint2 [64]     MIXED only: (region base, buckets) of each topic's tenant
    static constexpr uint32_t BYTES = (REG + (MIXED ? 64 * 8 : 0) + 15) & ~15u;
    static constexpr uint32_t STAGE = BYTES - STK;        // bytes available for staging
    // waves per SIMD the LDS slice allows (160 KB per CU, one wave per workgroup, 512-byte allocation granules assumed), at most 8
    static constexpr uint32_t ALLOC = (BYTES + 511) & ~511u;
    // (the MIXED instantiation carries 64-bit addresses and per-topic regions through the loop: at more than 5 waves -- <= 96 VGPRs --
    // the compiler spills 50-80 registers to scratch)
#ifndef BMQ_WALK_MAX_WAVES
#define BMQ_WALK_MAX_WAVES 8 // (variant builds: tools/build_variant.sh w7 -DBMQ_WALK_MAX_WAVES=7 gives the compiler 72 vector registers)
#endif
    static constexpr uint32_t WAVES_LDS = (163840 / ALLOC) / 4 > BMQ_WALK_MAX_WAVES ? BMQ_WALK_MAX_WAVES : (163840 / ALLOC) / 4;
    static constexpr uint32_t WAVES = MIXED && WAVES_LDS > 5 ? 5 : WAVES_LDS;
    static_assert(3 * 64 * 4 <= STK && PC % 8 == 0 && QC % 8 == 0 && STK % 16 == 0 && WAVES >= 1 && TC >= 2 * FAST_LEVELS && TC <= 1023, "layout");
    static_assert(FAST_LEVELS <= 32, "walk_meta keeps the levels behind an item in 5 bits");
    static_assert(QC >= 128 && PC >= 128, "one sink pushes up to 128 items and emits up to 128 ranges into an empty list");
    static_assert(STK + (QC + 64) * 8 <= BYTES, "a round reads 64 stack slots from `tail` on, whatever is there");
};

// (emulator builds only: which of the cold paths of the work stack and the range buffer a harness has driven through)
#ifdef BMQ_WAVE_EMU
struct WalkCoverage {
    unsigned long long flushes = 0, parks = 0, restores = 0;
};
inline WalkCoverage walk_cov;
#define BMQ_WALK_COV(field) do { if (threadIdx.x == 0) walk_cov.field++; } while (0)
#else
#define BMQ_WALK_COV(field) do { } while (0)
#endif

// The lane id behind an opaque copy: address arithmetic derived from it cannot be hoisted out of the walk loop (the compiler otherwise
// precomputes a dozen lane-dependent LDS addresses at the kernel's entry, keeps them alive across all phases and spills them to scratch
// -- and a kernel that uses ANY scratch got 5 wave slots per SIMD instead of 8 in this process, measured with the census: BMQ_DEBUG=8).
// (Reading the kernel's arguments through an opaque copy of the kernarg pointer, to keep the compiler from loading all 328 bytes at the
// entry, was tried too: it made the register allocation worse -- 60 VGPRs spilled -- and was dropped.)
#ifndef BMQ_WAVE_EMU
__device__ __forceinline__ uint32_t lane_here() {
    uint32_t v = threadIdx.x;
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#endif

// The body of k_walk as a function of (argument block, block index, the wave's LDS): the kernel below is this and nothing else; the
// persistent matcher of the batching front (k_poll, bmq_poll_kernel.h) runs the very same code on batches it finds in its request ring.
template <int TC, int QC, int PC, bool MIXED>
__device__ __forceinline__ void walk_wave(const BatchArgs& a, const uint32_t block_x, uint32_t* const lds) {
    using G = WalkLds<TC, QC, PC, MIXED>;
    uint32_t* const tokens = lds + G::TOK / 4;
    uint2* const stk = reinterpret_cast<uint2*>(lds + G::STK / 4);
    uint2* const p_rng = reinterpret_cast<uint2*>(lds + G::PRG / 4);
    uint8_t* const p_topic = reinterpret_cast<uint8_t*>(lds) + G::PTP;
    uint32_t* const cnt_visit = lds + G::VIS / 4;
    uint2* const t_region = reinterpret_cast<uint2*>(lds + G::REG / 4); // MIXED only
    uint32_t* const cnt_pairs = lds + G::TOK / 4; // phase 3 (the token table is dead by then)
    uint32_t* const cnt_routes = cnt_pairs + 64;
    uint32_t* const cursor = cnt_routes + 64;

    const uint32_t lane = threadIdx.x;
    uint32_t blk;
    if (BMQ_DBG(a, 32u)) { // (experiment) every XCD -- workgroup b runs on XCD b % 8 -- takes ONE contiguous eighth of the batch: a tenant's region is cached by one L2, not by eight
        const uint32_t per = (a.n_blocks + 7u) >> 3;
        blk = (block_x & 7u) * per + (per - 1u - (block_x >> 3));
        if ((block_x >> 3) >= per || blk >= a.n_blocks) return;
    } else {
        if (block_x >= a.n_blocks) return;
        // last blocks first: batches arrive grouped by tenant with the hot tenants (L2-resident regions, fast waves) in
        // front; starting with the cold ones leaves the fast waves for the tail of the launch (measured: -4 % on C3)
        blk = BMQ_DBG(a, 256u) ? block_x : a.n_blocks - 1 - block_x; // (256: experiment, first blocks first)
    }
    // A wave owns TPW = 2^tpw_shift consecutive topics.  64 for large batches; a small batch is spread over more waves (16 or 4
    // topics each): the walk phase is a chain of dependent line fetches whose length is ~ max(depth, items / 64), so a wave with
    // fewer topics finishes sooner and a 10 k-topic batch fills the chip instead of 157 waves on 256 CUs.
    const uint32_t tpw = 1u << a.tpw_shift;
    const uint32_t t = (blk << a.tpw_shift) + lane;
    const bool valid = lane < tpw && t < a.n_topics;
    const bool dbg_w = BMQ_DBG(a, 2u) && a.dbg_wave;
    const unsigned long long clk0 = dbg_w ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long clk0c = (BMQ_DBG(a, 8u) && a.dbg_wave) ? __builtin_amdgcn_s_memtime() : 0ull;

    // ---- phase 1: tokenise ---------------------------------------------------------------------------------------------
    const uint32_t t_first = blk << a.tpw_shift, t_end = min(t_first + tpw, a.n_topics);
    const uint32_t s_beg = scalar_words(a.topic_off)[t_first], s_end = scalar_words(a.topic_off)[t_end]; // wave-uniform: scalar loads
    const uint32_t a0 = s_beg & ~15u;
    const bool staged = (s_end - a0) + 32u <= G::STAGE;
    if (staged) { // coalesced 16-byte copies of the wave's contiguous topic bytes into LDS
        uint4* dst = reinterpret_cast<uint4*>(stk);
        const uint4* src = reinterpret_cast<const uint4*>(a.topics + a0);
        const uint32_t n16 = (s_end - a0 + 15) >> 4;
        for (uint32_t o = lane; o < n16; o += 64) dst[o] = src[o];
    }
    uint32_t nlev = 0, tbytes = 0, pos = 0, end = 0, ti = 0xFFFFFFFFu;
    if (valid) {
        pos = a.topic_off[t];
        end = a.topic_off[t + 1];
        tbytes = end - pos;
        ti = a.topic_tenant[t];
    }
    // walked here: rows of tenants the batch names -- with in-batch de-duplication (k_dedup) only the representative of every (tenant, topic)
    // (a topic of a tenant the index turns out not to know is tokenised for nothing: rare)
    const bool asked = ti < a.n_tenants && (a.rep == nullptr || a.rep[t] == t);
    wave_sync();
    const uint8_t* lbytes = reinterpret_cast<const uint8_t*>(stk);
    const uint32_t* lwords = reinterpret_cast<const uint32_t*>(stk);
    const uint8_t* gbytes = a.topics;
    // level count = '/' bytes + 1 (UTIL/TopicUtil.java:206-225: empty levels count), four bytes per step over the ALIGNED words the topic lies in
    if (asked) {
        uint32_t cnt = 1;
        for (uint32_t p = pos & ~3u; p < end; p += 4) {
            const uint32_t w = staged ? lwords[(p - a0) >> 2] : *reinterpret_cast<const uint32_t*>(gbytes + p);
            const uint32_t x = w ^ 0x2F2F2F2Fu;
            uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); // 0x80 in every byte that is '/' (exact: no carries across bytes)
            if (p < pos) z &= 0xFFFFFFFFu << (8u * (pos - p));        // bytes in front of the topic
            if (p + 4 > end) z &= 0xFFFFFFFFu >> (8u * (p + 4 - end)); // ... behind it
            cnt += (uint32_t)__popc(z);
        }
        nlev = cnt;
    }
    const bool deep = nlev > (uint32_t)FAST_LEVELS; // Setting.MaxTopicLevels is 16: deeper topics take the per-lane walk of k_walk_slow
    const bool sys = asked && end > pos && (staged ? (uint32_t)lbytes[pos - a0] : (uint32_t)gbytes[pos]) == '$';
    const unsigned long long clkA = dbg_clock(dbg_w); // topic bytes staged, offsets loaded, levels counted
    uint32_t tok_base = 0; // where this topic's tokens start in the table (of the chunk that holds it)
    // the topics of `chunk` -> tokens[tok_base ..]; from_lds: the staged bytes are still there (first chunk only)
    auto tokenise = [&](bool mine, bool from_lds) {
        auto byte_at = [&](uint32_t i) -> uint32_t { return from_lds ? (uint32_t)lbytes[i - a0] : (uint32_t)gbytes[i]; };
        auto word_at = [&](uint32_t i) -> uint32_t {
            if (!from_lds) return global_word_at(gbytes, i);
            const uint32_t rel = i - a0;
            return __builtin_amdgcn_alignbyte(lwords[(rel >> 2) + 1], lwords[rel >> 2], rel & 3u);
        };
        bool more = mine;
        uint32_t p = pos;
        for (uint32_t l = 0; ballot64(more) != 0; l++) {
            LevelHash h = level_hash_init();
            uint32_t inl[4] = {0, 0, 0, 0}, len = 0;
            const uint32_t start = p;
            const bool now = more;
            if (more) {
                bool last;
                scan_level(p, end, true, word_at, h, inl, len, last);
                more = !last;
            }
            const uint32_t tok = now ? (BMQ_DBG(a, 128u) ? dict_lookup(a.ix, h, len, inl, start, byte_at) : dict_lookup_by_slot(a.ix, h, len, inl, start, byte_at))
                                     : TOK_UNKNOWN;
            if (now) tokens[tok_base + l] = tok;
        }
    };

    // ---- phase 2: drain the work stack ----------------------------------------------------------------------------
    // The work list is a STACK (newest items first): depth-first order keeps it at a few pending siblings per topic,
    // where breadth-first order would have to hold a whole frontier level of all 64 topics.
    // Neither LDS list bounds the walk: a full range buffer is flushed to, and a full stack parked in, the global spill
    // area, as chunks {header record, payload records}; the header links to the wave's previous chunk of the same kind
    // (base, length; length 0 ends the chain), so the bookkeeping is two wave-uniform registers per chain.
    // Everything below that is named s_* or is a list fill level is WAVE-UNIFORM and kept so explicitly (sgpr()).
    uint32_t tail = 0, pcount = 0, rounds = 0, items = 0;
    uint32_t fl_base = 0, fl_len = 0; // last flushed range chunk
    uint32_t qs_base = 0, qs_len = 0; // last parked stack chunk (LIFO)
    auto spill_alloc = [&](const BatchArgs& c, uint32_t ln, uint32_t n, uint32_t& base) -> bool { // wave-uniform; n payload records + header
        unsigned long long sb = 0;
        uint32_t ok = 1;
        if (ln == 0) ok = pair_alloc(c.subs + N_SUB, c.spill_cap, blk, n + 1, sb) ? 1u : 0u;
        const uint32_t lo = sgpr((uint32_t)sb), hi = sgpr((uint32_t)(sb >> 32));
        const bool fits = sgpr(ok) != 0 && hi == 0 && lo + n + 1 < 0xFFFFFFFFu && lo + n + 1 > lo;
        if (!fits && ln == 0) atomicOr(&c.ctr->status, ST_NEED_SPILL); // the batch is re-run with a larger area
        base = lo;
        return fits;
    };
    const unsigned long long clk1 = dbg_clock(dbg_w);
    // what a resolved item leaves behind: its matched ranges go into the LDS range buffer, its children onto the stack
    // (The four predicates arrive as INTEGERS -- a count that is zero unless its range is to be emitted, a flag, the Bloom word's sign --
    // so that every ballot below is one v_cmp on a vector register: a ballot of a boolean that was combined on the scalar unit costs a
    // v_cndmask + v_cmp to bring it back.)
    auto sink = [&](uint32_t own_begin, uint32_t own_count /* 0: no such range */, uint32_t hash_begin, uint32_t hash_count /* 0: none */,
                    uint32_t lit /* != 0: push the literal child probe */, uint32_t bloom /* bit 31: push the '+' child probe */, uint32_t child,
                    uint32_t cmeta /* the children's meta, kind L */, uint32_t tl, uint32_t ln /* lane_here() */) {
        const bool emit_own = own_count != 0, emit_hash = hash_count != 0, push_l = lit != 0, push_h = (int32_t)bloom < 0;
        const unsigned long long m_own = ballot64(emit_own), m_hash = ballot64(emit_hash);
        const unsigned long long m_l = ballot64(push_l), m_h = ballot64(push_h);
        if ((m_own | m_hash) != 0) {
            const uint32_t n_own = (uint32_t)__popcll(m_own), n_emit = n_own + (uint32_t)__popcll(m_hash);
            if (__builtin_expect(pcount + n_emit > (uint32_t)PC, 0)) { // (cold) this round's matches do not fit: the buffer is flushed to the spill area first
                const BatchArgs& c = a;
                BMQ_WALK_COV(flushes);
                uint32_t cb;
                if (spill_alloc(c, ln, pcount, cb)) {
                    if (ln == 0) c.spill[cb] = make_uint4(fl_base, fl_len, 0u, 0u);
                    for (uint32_t i = ln; i < pcount; i += 64) c.spill[cb + 1 + i] = make_uint4(p_rng[i].x, p_rng[i].y, p_topic[i], 0u);
                    fl_base = cb;
                    fl_len = pcount;
                }
                pcount = 0;
                wave_sync();
            }
            if (emit_own) {
                const uint32_t p = pcount + rank_below(m_own);
                p_rng[p] = make_uint2(own_begin, own_count);
                p_topic[p] = (uint8_t)tl;
            }
            if (emit_hash) {
                const uint32_t p = pcount + n_own + rank_below(m_hash);
                p_rng[p] = make_uint2(hash_begin, hash_count);
                p_topic[p] = (uint8_t)tl;
            }
            pcount += n_emit;
        }
        // children -> stack; if they do not fit, the pending (older) items are parked and the walk goes on with the children
        if ((m_l | m_h) != 0) {
            const uint32_t n_l = (uint32_t)__popcll(m_l), n_push = n_l + (uint32_t)__popcll(m_h);
            if (__builtin_expect(tail + n_push > (uint32_t)QC, 0)) { // (cold)
                const BatchArgs& c = a;
                BMQ_WALK_COV(parks);
                uint32_t cb;
                if (spill_alloc(c, ln, tail, cb)) {
                    if (ln == 0) c.spill[cb] = make_uint4(qs_base, qs_len, 0u, 0u);
                    for (uint32_t i = ln; i < tail; i += 64) c.spill[cb + 1 + i] = make_uint4(stk[i].x, stk[i].y, 0u, 0u);
                    qs_base = cb;
                    qs_len = tail;
                }
                tail = 0;
                wave_sync();
            }
            if (push_l) stk[tail + rank_below(m_l)] = make_uint2(child, cmeta);
            if (push_h) stk[tail + n_l + rank_below(m_h)] = make_uint2(child, cmeta | KIND_P);
            tail += n_push;
        }
        wave_sync();
    };
    // One drain = the walk of everything on the stack.  MODE 0: the wave's items belong to ONE tenant whose region (s_rptr, s_rbuckets)
    // sits in scalar registers -- the bucket index is one v_mul_hi against an SGPR, the line's address an SGPR base + a 32-bit lane offset;
    // MODE 1: the same for a region of 2^25 buckets and more (64-bit addresses); MODE 2 (MIXED): (base, buckets) per topic from LDS.
    auto drain = [&](auto mode_tag, const TrieSlot* s_rptr, uint32_t s_rbuckets) {
        constexpr int MODE = decltype(mode_tag)::value;
        while ((tail | qs_len) != 0) {
            const uint32_t ln = lane_here();
            if (__builtin_expect(tail == 0, 0)) { // (cold) the stack ran dry: take the most recently parked chunk back
                const BatchArgs& c = a;
                BMQ_WALK_COV(restores);
                const uint32_t hx = sgpr(c.spill[qs_base].x), hy = sgpr(c.spill[qs_base].y);
                for (uint32_t i = ln; i < qs_len; i += 64) {
                    const uint4 r = c.spill[qs_base + 1 + i];
                    stk[i] = make_uint2(r.x, r.y);
                }
                tail = qs_len;
                qs_base = hx;
                qs_len = hy;
                wave_sync();
            }
            const uint32_t take = tail < 64u ? tail : 64u;
            tail -= take;
            rounds++;
            items += take;
            const bool live = ln < take;
            // (all lanes read: a lane without an item reads whatever lies above the stack's top and turns it into a probe of the root's
            // '+' edge -- some bucket of the region, the same for all of them: one line, harmless)
            const uint2 raw = stk[tail + ln];
            const uint32_t node = live ? raw.x : 0u, meta = live ? raw.y : KIND_P;
            const uint32_t tl = meta & 63u;
            const uint32_t tcur = tokens[(meta >> 6) & 1023u];        // the topic's token at the item's level
            const uint32_t tnext = tokens[((meta >> 6) & 1023u) + 1]; // ... at the next one (whatever follows the topic's last: unused then)
            const uint32_t tnext2 = tokens[((meta >> 6) & 1023u) + 2]; // ... and at the one after it (the literal child of a '+' child found in the same line)
            const uint32_t tok = (meta & KIND_P) ? TOK_PLUS : tcur;
            uint2 reg = make_uint2(0u, s_rbuckets);
            if (MODE == 2) reg = t_region[tl];
            uint32_t bk = edge_bucket(node, tok, MODE == 2 ? reg.y : s_rbuckets);
            Line64 line;
            if (MODE == 0) load_line64_s(s_rptr, bk * 64u, line);
            else if (MODE == 1) load_line64(s_rptr + 2 * (size_t)bk, line);
            else load_line64(s_rptr + (live ? (size_t)reg.x + 2 * (size_t)bk : (size_t)0), line); // (MODE 2: s_rptr = the slot table)
            wave_sync(); // every ln holds its item in registers: the stack above `tail` may be overwritten by the pushes below
            bool m0 = line.a0.x == node && line.a0.y == tok;
            bool m1 = line.b0.x == node && line.b0.y == tok;
            if (__builtin_expect(ballot64(live && !m0 && !m1 && line.a0.x != NONE && line.b0.x != NONE) != 0, 0)) {
                // (cold) a home bucket full of other edges (rare at load factor 1/2): first-free probing continues.  Bounded by the region
                // size so that not even a damaged image can hang the GPU.
                const uint32_t nb = MODE == 2 ? reg.y : s_rbuckets;
                const TrieSlot* rp = MODE == 2 ? s_rptr + reg.x : s_rptr;
                bool again = live && !m0 && !m1 && line.a0.x != NONE && line.b0.x != NONE;
                for (uint32_t probes = 1; again && probes < nb; probes++) {
                    bk = (bk + 1 == nb) ? 0 : bk + 1;
                    load_line64(rp + 2 * (size_t)bk, line);
                    m0 = line.a0.x == node && line.a0.y == tok;
                    m1 = line.b0.x == node && line.b0.y == tok;
                    again = !m0 && !m1 && line.a0.x != NONE && line.b0.x != NONE;
                }
            }
            const bool found = live && (m0 || m1);
            const uint32_t own_begin = m1 ? line.b0.z : line.a0.z, own_count = m1 ? line.b0.w : line.a0.w;
            const uint32_t hash_begin = m1 ? line.b1.x : line.a1.x, hash_count = m1 ? line.b1.y : line.a1.y;
            const uint32_t child = m1 ? line.b1.z : line.a1.z, bloom = m1 ? line.b1.w : line.a1.w;
            const uint32_t cmeta = (meta & ~KIND_P) + WALK_META_CHILD; // the same topic, one level on
            const uint32_t rem = (meta >> 16) & 31u;                   // levels of the topic behind this item's
            const bool last = rem == 0;
            const uint32_t bloom_in = (found && !last) ? bloom : 0u;   // children only below an inner level
            // Layout v3: the OTHER slot of the line holds the found node's '+' child whenever it was free when that child came into being
            // (bmq_build_core.h, trie_child).  Then the child P is resolved right here -- it consumes the topic's next level whatever its
            // token --: no item, no line, no round of its own.  Otherwise (the slot holds some other edge) the '+' child sits at its hashed home.
            const uint32_t o_parent = m1 ? line.a0.x : line.b0.x, o_token = m1 ? line.a0.y : line.b0.y;
            const bool plus_here = (int32_t)bloom_in < 0 && o_parent == child && o_token == TOK_PLUS;
            // (P's payload leaves the line's registers HERE, in front of the first sink: with the whole line alive across it the compiler
            // spilled 30 vector registers of the 64 that 8 waves per SIMD allow)
            const uint32_t p_own_begin = m1 ? line.a0.z : line.b0.z, p_own_count = m1 ? line.a0.w : line.b0.w;
            const uint32_t p_hash_begin = m1 ? line.a1.x : line.b1.x, p_hash_count = m1 ? line.a1.y : line.b1.y;
            const uint32_t p_node = m1 ? line.a1.z : line.b1.z, p_bloom = m1 ? line.a1.w : line.b1.w;
            if (found) atomicAdd(&cnt_visit[tl], plus_here ? 2u : 1u); // (LDS, no return value)
            // the literal child: the next token is known to the dictionary (TOK_UNKNOWN = 0: min() drops it) and the node's Bloom word has its bit
            // ONE sink body run once or twice (not two copies of it: the second copy cost 11 spilled vector registers): first the found node,
            // then -- if some lane of the wave has one -- the '+' child P beside it.  (P's own '+' child cannot lie beside P -- that slot
            // holds P's parent --: it is at its hashed home, an ordinary '+' probe.)
            uint32_t e_own_begin = own_begin, e_own_count = (found && last) ? own_count : 0u, e_hash_begin = hash_begin;
            uint32_t e_hash_count = found ? hash_count : 0u; // "<path>/#" matches whatever follows, also nothing
            uint32_t e_lit = min((bloom_in >> bloom_bit(tnext)) & 1u, tnext), e_bloom = plus_here ? 0u : bloom_in, e_child = child, e_meta = cmeta;
            const bool second = ballot64(plus_here) != 0;
#pragma clang loop unroll(disable)
            for (uint32_t part = 0;; part++) {
                sink(e_own_begin, e_own_count, e_hash_begin, e_hash_count, e_lit, e_bloom, e_child, e_meta, tl, ln);
                if (part == 1 || !second) break;
                const bool p_last = rem == 1; // P's level is the topic's last
                const uint32_t p_bloom_in = (plus_here && !p_last) ? p_bloom : 0u;
                e_own_begin = p_own_begin, e_own_count = (plus_here && p_last) ? p_own_count : 0u, e_hash_begin = p_hash_begin;
                e_hash_count = plus_here ? p_hash_count : 0u;
                e_lit = min((p_bloom_in >> bloom_bit(tnext2)) & 1u, tnext2), e_bloom = p_bloom_in, e_child = p_node, e_meta = cmeta + WALK_META_CHILD;
            }
        }
    };
    // Round 0 visits the tenant roots: their slot payload comes with the directory entry, no line is fetched.  Layout v3: the entry also says
    // where the root's '+' child P0 lies; the grouped instantiations read P0's whole line on the scalar unit (the wave shares the tenant) and
    // resolve P0 -- and P0's own '+' child PP0 if it lies beside P0 -- right here.  rp: what was read: has (0: nothing, 1: P0, 2: P0 and PP0),
    // then per node (own_begin, own_count, hash_begin, hash_count) + (node id, Bloom word).  MIXED passes has = 0: P0 is then probed for at
    // its hashed home like any other node and the drain finds PP0 beside it.
    struct RootPlus {
        uint32_t has;
        uint4 p0;
        uint2 p0n;
        uint4 pp0;
        uint2 pp0n;
    };
    auto boot = [&](bool mine, uint32_t r_hash_begin, uint32_t r_hash_count, uint32_t r_bloom, const RootPlus& rp) {
        const uint32_t ln = lane_here();
        bool actp = mine; // the lane's topic reaches the node of this part
        // part 0: the root; 1: P0 = "+" (consumes the first level); 2: PP0 = "+/+".  ONE sink body (see the drain); the per-part values are
        // wave-uniform selects on the scalar unit.
#pragma clang loop unroll(disable)
        for (uint32_t part = 0; part <= rp.has; part++) {
            const uint4 pl = part == 0 ? make_uint4(0u, 0u, r_hash_begin, r_hash_count) : (part == 1 ? rp.p0 : rp.pp0);
            const uint32_t nd = part == 0 ? 0u : (part == 1 ? rp.p0n.x : rp.pp0n.x), bl = part == 0 ? r_bloom : (part == 1 ? rp.p0n.y : rp.pp0n.y);
            const uint32_t tk = tokens[actp ? tok_base + part : 0u]; // (a lane that is not active reads entry 0: its own tok_base may lie beyond the table)
            const bool root_sys = part == 0 && sys;                  // a first-level wildcard never matches a '$' topic
            const uint32_t bloom_in = (actp && nlev > part) ? bl : 0u; // children only while the topic has a level left
            if (part != 0 && actp) cnt_visit[ln] += 1u; // (the lane's own topic; the drain's atomics come later)
            sink(pl.x, (actp && nlev == part) ? pl.y : 0u /* (part 0: pl.y = 0: a topic has at least one level) */,
                 pl.z, (actp && !root_sys) ? pl.w : 0u, min((bloom_in >> bloom_bit(tk)) & 1u, tk),
                 (root_sys || part < rp.has) ? (bloom_in & ~BLOOM_PLUS) : bloom_in, // ('+' child resolved by the next part, or probed for at its hashed home)
                 nd, walk_meta(ln, tok_base + part, nlev - 1 - part), ln, ln);
            actp = actp && !root_sys && (int32_t)bloom_in < 0;
            if (ballot64(actp) == 0) break;
        }
    };
    {
    const BatchArgs& b = a;
    const bool run = !BMQ_DBG(b, 1u);
    uint2 t_region_early = make_uint2(0u, 1u);
    // MIXED: every topic resolves its own tenant (three dependent requests per lane); the region goes to LDS, the root's payload stays in
    // three registers (a topic of a tenant the index does not know: an empty root, nothing is pushed)
    uint32_t m_hash_begin = 0, m_hash_count = 0, m_bloom = 0;
    if (MIXED) {
        uint2 reg = make_uint2(0u, 1u);
        if (asked) {
            const TenantSlot rg = resolve_tenant(b, ti);
            if (tenant_known(rg)) reg = make_uint2(rg.base, rg.buckets), m_hash_begin = rg.root_hash_begin, m_hash_count = rg.root_hash_count, m_bloom = rg.root_lit_bloom;
        }
        t_region_early = reg;
    }
    // The wave's topics in CHUNKS that fit the token table -- one chunk unless the depths of the 64 topics add up to more than TC levels.
    // A chunk = the longest prefix of the topics still to be walked whose tokens fit; its topics are tokenised (the first chunk from the
    // staged bytes, later ones from global memory: the staging area is stack and range buffer by then), then walked.
    unsigned long long left = ballot64(asked && !deep);
    if (left == 0) cnt_visit[lane] = 0; // (nothing to walk: the counters are read below all the same)
    for (bool first = true; left != 0; first = false) {
        const bool cand = (left >> lane) & 1ull;
        uint32_t total;
        tok_base = wave_excl_scan(cand ? nlev : 0u, lane, total);
        const unsigned long long chunk = ballot64(cand && tok_base + nlev <= (uint32_t)TC); // (never empty: a topic has <= FAST_LEVELS <= TC levels)
        left &= ~chunk;
        const bool mine = (chunk >> lane) & 1ull;
        tokenise(mine, first && staged);
        if (first) cnt_visit[lane] = 0; // (first chunk: the staged bytes are dead from here on: the area becomes stack + range buffer + counters)
        wave_sync();
        if (!run) continue;
        if (MIXED) {
            if (first) { // (the staging area is free now)
                t_region[lane] = t_region_early;
                wave_sync();
            }
            boot(mine, m_hash_begin, m_hash_count, m_bloom, RootPlus{0u, make_uint4(0u, 0u, 0u, 0u), make_uint2(0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint2(0u, 0u)});
            drain(std::integral_constant<int, 2>{}, b.ix.trie, 1u);
        } else {
            // the chunk's tenants, one after the other (batches arrive grouped by tenant: one, sometimes two)
            unsigned long long todo = chunk;
            for (uint32_t n_seg = 0; todo != 0; n_seg++) {
                if (n_seg == WALK_MAX_TENANTS) { // a batch that is not grouped by tenant: it runs again through the MIXED instantiation
                    if (lane == 0) atomicOr(&b.ctr->status, ST_WANT_MIXED);
                    left = 0;
                    break;
                }
                const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)ti, (int)__builtin_ctzll(todo));
                const unsigned long long seg = ballot64(mine && ti == cur);
                todo &= ~seg;
                const TenantSlot rg = resolve_tenant_uniform(b, cur); // on the scalar unit
                if (!tenant_known(rg)) continue;                      // no such tenant: no routes
                const uint32_t s_rbase = sgpr(rg.base), s_rbuckets = sgpr(rg.buckets);
                RootPlus rp{0u, make_uint4(0u, 0u, 0u, 0u), make_uint2(0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint2(0u, 0u)};
                const uint32_t s_rplus = sgpr(rg.root_plus);
                if ((int32_t)sgpr(rg.root_lit_bloom) < 0 && s_rplus != NONE) { // the line of the root's '+' child: 64 bytes through the scalar cache
                    ScalarWords sp = scalar_words(b.ix.trie + (size_t)s_rbase + (s_rplus & ~1u));
                    const uint32_t o0 = (s_rplus & 1u) * 8u, o1 = 8u - o0; // word offsets of P0's slot and of the other one
                    rp.has = 1u, rp.p0 = make_uint4(sp[o0 + 2], sp[o0 + 3], sp[o0 + 4], sp[o0 + 5]), rp.p0n = make_uint2(sp[o0 + 6], sp[o0 + 7]);
                    if (sp[o1] == rp.p0n.x && sp[o1 + 1] == TOK_PLUS)
                        rp.has = 2u, rp.pp0 = make_uint4(sp[o1 + 2], sp[o1 + 3], sp[o1 + 4], sp[o1 + 5]), rp.pp0n = make_uint2(sp[o1 + 6], sp[o1 + 7]);
                }
                boot(mine && ti == cur, sgpr(rg.root_hash_begin), sgpr(rg.root_hash_count), sgpr(rg.root_lit_bloom), rp);
                if (s_rbuckets < (1u << 25)) drain(std::integral_constant<int, 0>{}, b.ix.trie + s_rbase, s_rbuckets);
                else drain(std::integral_constant<int, 1>{}, b.ix.trie + s_rbase, s_rbuckets);
            }
        }
    }
    }

    // ---- phase 3: ranges grouped by topic -> global; per-topic bookkeeping --------------------------------------
    const unsigned long long clk2 = dbg_clock(dbg_w);
    const BatchArgs& d = a;
    const uint32_t l3 = lane_here();
    cnt_pairs[l3] = 0;
    cnt_routes[l3] = 0;
    wave_sync();
    for (uint32_t cb = fl_base, cl = fl_len; cl != 0;) { // (cold) what was flushed
        const uint32_t hx = sgpr(d.spill[cb].x), hy = sgpr(d.spill[cb].y);
        for (uint32_t i = l3; i < cl; i += 64) {
            const uint4 r = d.spill[cb + 1 + i];
            atomicAdd(&cnt_pairs[r.z], 1u);
            atomicAdd(&cnt_routes[r.z], r.y & ~RANGE_INDIRECT);
        }
        cb = hx;
        cl = hy;
    }
    for (uint32_t i = l3; i < pcount; i += 64) { // counted here, once per range, instead of two LDS atomics per match
        const uint32_t tp = p_topic[i];
        atomicAdd(&cnt_pairs[tp], 1u);
        atomicAdd(&cnt_routes[tp], p_rng[i].y & ~RANGE_INDIRECT);
    }
    wave_sync();
    const bool flagged = deep;
    const uint32_t np = flagged ? 0u : cnt_pairs[l3];
    const uint32_t nr = flagged ? 0u : cnt_routes[l3];
    uint32_t total_pairs;
    const uint32_t excl = wave_excl_scan(np, l3, total_pairs);
    total_pairs = sgpr(total_pairs);
    unsigned long long base = 0;
    uint32_t fits_l = 1;
    if (l3 == 0 && total_pairs) fits_l = pair_alloc(d.subs, d.pair_cap, blk, total_pairs, base) ? 1u : 0u;
    base = ((unsigned long long)sgpr((uint32_t)(base >> 32)) << 32) | sgpr((uint32_t)base);
    const bool fits = sgpr(fits_l) != 0;
    if (!fits && l3 == 0) atomicOr(&d.ctr->status, ST_NEED_PAIRS);
    cursor[l3] = excl;
    wave_sync();
    if (fits && total_pairs) {
        // flushed chunks first, OLDEST first (the chain runs newest to oldest: it is laid out in the dead stack area and
        // replayed backwards), then what is still in LDS: every topic's ranges stay in discovery order, which is close to
        // ascending id order and keeps the ordering work of k_expand small
        auto copy_chunk = [&](uint32_t cb, uint32_t cl) {
            for (uint32_t i = l3; i < cl; i += 64) {
                const uint4 r = d.spill[cb + 1 + i];
                const uint32_t dst = atomicAdd(&cursor[r.z], 1u);
                d.pairs[base + dst] = MatchRange{r.x, r.y};
            }
        };
        uint32_t n_ch = 0;
        for (uint32_t cb = fl_base, cl = fl_len; cl != 0;) {
            const uint32_t hx = sgpr(d.spill[cb].x), hy = sgpr(d.spill[cb].y);
            if (n_ch < (uint32_t)QC) {
                stk[n_ch] = make_uint2(cb, cl); // all lanes store the same values
                n_ch++;
            } else copy_chunk(cb, cl); // a chain longer than the stack area: order is only a matter of speed
            cb = hx;
            cl = hy;
        }
        wave_sync();
        while (n_ch != 0) {
            n_ch--;
            copy_chunk(sgpr(stk[n_ch].x), sgpr(stk[n_ch].y));
        }
        for (uint32_t i = l3; i < pcount; i += 64) {
            const uint32_t dst = atomicAdd(&cursor[p_topic[i]], 1u);
            const uint2 r = p_rng[i];
            d.pairs[base + dst] = MatchRange{r.x, r.y};
        }
    }
    if (valid) {
        d.pair_off[t] = (uint32_t)(base + excl); // pair_cap < 2^32 is enforced by the host
        d.pair_cnt[t] = np;
        d.route_cnt[t] = nr;
        if (flagged && ti < d.n_tenants) { // deeper than FAST_LEVELS: the per-lane walk of k_walk_slow
            const uint32_t sp = atomicAdd(&d.ctr->slow_count, 1u);
            if (sp < d.slow_cap) d.slow_list[sp] = t;
            else atomicOr(&d.ctr->status, ST_NEED_SLOW);
        }
    }
    const uint32_t vis = cnt_visit[l3];
    if (valid && d.visit_cnt) d.visit_cnt[t] = vis;
    const unsigned long long wsum = wave_sum_u64(nr);
    const unsigned long long wvis = wave_sum_u64(vis);
    const unsigned long long wbytes = wave_sum_u64(tbytes);
    if (l3 == 0) {
        if (d.rep == nullptr) { // (with in-batch de-duplication k_fill writes these, every duplicate row counted with its representative's figures)
            d.wave_sums[blk] = wsum;
            if (wsum) atomicAdd(&d.super_sums[(size_t)(blk >> SUPER_SHIFT) * SUPER_STRIDE], wsum);
            // statistics: a plain store per wave.  (Atomics were measured twice: on the batch counters they set the kernel's duration,
            // three more per wave on the super-block's line still cost +20 us per 1 M topics and +7 us per 10 k.)
            d.blk_stats[blk] = make_uint4((uint32_t)wvis, total_pairs, (uint32_t)wbytes, heavy_mark(d, blk, total_pairs, wsum)); // (.w: k_expand splits the block)
        }
        if (BMQ_DBG(d, 8u) && d.dbg_wave) { // residency census: when and where this wave ran (tools: BMQ_DEBUG=8)
            const unsigned long long clk3 = __builtin_amdgcn_s_memtime();
            const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, all 32 bits
            const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
            d.dbg_wave[blk] = make_uint4((uint32_t)clk0c, (uint32_t)(clk0c >> 32), (uint32_t)(clk3 - clk0c), (hw & 0xFFFFu) | (xcc << 16));
        }
        if (dbg_w) {
            const unsigned long long clk3 = __builtin_amdgcn_s_memtime();
            d.dbg_wave[blk] = make_uint4((uint32_t)(clk1 - clk0), (uint32_t)(clk2 - clk1), (uint32_t)(clk3 - clk2), rounds | (items << 8));
            d.dbg_wave[d.n_blocks + blk] = make_uint4((uint32_t)(clkA - clk0), 0u, (uint32_t)(clk1 - clkA), 0u); // phase 1 in detail
        }
    }
}

template <int TC, int QC, int PC, bool MIXED>
__global__ __launch_bounds__(64, (WalkLds<TC, QC, PC, MIXED>::WAVES)) void k_walk(BatchArgs a) {
    __shared__ __align__(16) uint32_t lds[WalkLds<TC, QC, PC, MIXED>::BYTES / 4];
    walk_wave<TC, QC, PC, MIXED>(a, blockIdx.x, lds);
}

#if BMQ_EXPERIMENTS
// Residency probe (BMQ_DEBUG=16, profiling experiments only): a kernel with k_walk's launch shape and resource footprint (one-wave
// workgroups, 5024 B of LDS, 64 VGPRs, 78 SGPRs, the same argument block) that only waits ~100 k shader ticks and reports where it ran.
__global__ __launch_bounds__(64, 8) void k_occ_probe(BatchArgs a) {
    __shared__ uint32_t lds[5024 / 4];
    asm volatile("s_mov_b32 s71, 0" ::: "s71");
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    lds[threadIdx.x] = a.n_topics;
    while (__builtin_amdgcn_s_memtime() - t0 < 100000u) __builtin_amdgcn_s_sleep(8);
    const uint32_t acc = lds[(threadIdx.x + 1) & 63];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x < a.n_blocks) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        a.dbg_wave[blockIdx.x] = make_uint4((uint32_t)t0, (uint32_t)(t0 >> 32), (uint32_t)(t1 - t0) | (acc == 0x1234567u), (hw & 0xFFFFu) | (xcc << 16));
    }
}

#endif // BMQ_EXPERIMENTS

} // namespace bmq
