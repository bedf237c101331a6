// bmq_retain_kernels.h -- gfx950 kernel of the retain-direction match: wildcard FILTERS against the index of
// retained TOPICS (bmq_retain.h).  Semantics: RetainMatcher (RS/index/RetainTopicIndex.java:36-124) driven by
// TopicLevelTrie.lookup (UTIL/index/TopicLevelTrie.java:190-249); '$' children are skipped by a wildcard in the first
// topic level (currentLevel == 1 there, because level 0 is the tenant).
//
// k_retain_walk: persistent waves, one filter per wave at a time (grid-stride).  The frontier is a list of NODE RANGES
// in LDS (overflowing into a per-wave global scratch).  '+' maps a range to the range of its children with two node
// reads; a literal level looks every node of the frontier up in the edge hash (one 64-byte line each, lanes in
// parallel); '#' and end-of-filter emit topic-id RANGES.  The ranges of a filter are buffered in LDS and copied out once
// their number is known; only a filter with more than R_OUT ranges is walked a second time (count, then write).  The CSR is then produced by the dist direction's k_scan_blocks /
// k_expand / k_sort_rows (same MatchRange plumbing).
#pragma once
#include <hip/hip_runtime.h>

#include "bmq_dist_kernels.h"
#include "bmq_retain.h"
#include "bmq_retain_args.h"
#include "bmq_retain_core.h"

namespace bmq {

constexpr uint32_t R_MAXL = 64;        // filter levels supported by the kernel
#ifndef BMQ_R_FRONT
#define BMQ_R_FRONT 256
#endif
#ifndef BMQ_R_OUT
#define BMQ_R_OUT 256
#endif
#ifndef BMQ_OV_OUT
#define BMQ_OV_OUT 128
#endif
// The LDS lists of a wave decide how many filters a CU works on at a time -- and the walk is a chain of dependent reads per filter, so
// filters in flight are what hides its latency: 768 / 512 / 256 entries (20 KB: 8 waves per CU) -> 256 / 256 / 128 (9 KB: 16 waves per CU).
constexpr uint32_t R_FRONT = BMQ_R_FRONT; // frontier ranges kept in LDS (per buffer); more: the wave's global scratch
constexpr uint32_t R_OUT = BMQ_R_OUT;     // matched ranges of one filter buffered in LDS (more: a second, writing walk)
constexpr uint32_t OV_OUT = BMQ_OV_OUT;   // matched OVERLAY topic ids of one filter ordered in LDS (more: flushed unordered, the row is repaired)
#ifndef BMQ_R_WAVES_PER_CU
#define BMQ_R_WAVES_PER_CU 16
#endif
constexpr uint32_t R_WAVES_PER_CU = BMQ_R_WAVES_PER_CU; // persistent waves per CU (LDS: 16 B per frontier entry + 8 B per range / overlay id + ~1.7 KB)
static_assert(R_WAVES_PER_CU * (16u * BMQ_R_FRONT + 8u * BMQ_R_OUT + 8u * BMQ_OV_OUT + 1700u) <= 160u * 1024u, "the retain walk's LDS lists do not fit that many waves");
struct Frontier {
    uint32_t* lb;
    uint32_t* lc;
    uint2* g;
    __device__ __forceinline__ uint2 get(uint32_t i) const { return i < R_FRONT ? make_uint2(lb[i], lc[i]) : g[i - R_FRONT]; }
    __device__ __forceinline__ void put(uint32_t i, uint32_t b, uint32_t c) const {
        if (i < R_FRONT) {
            lb[i] = b;
            lc[i] = c;
        } else g[i - R_FRONT] = make_uint2(b, c);
    }
};

// (parent, token) -> child inside the tenant's edge region (all ids tenant-local); NONE if absent
__device__ __forceinline__ uint32_t redge_lookup(const RetainIndexView& ix, uint32_t edge_base, uint32_t mask, uint32_t parent,
                                                 uint32_t token) {
    uint32_t bk = redge_bucket(parent, token, mask);
    for (uint32_t probes = 0; probes <= mask; probes++) { // bounded: a damaged table must not hang the GPU
        Line64 ln; // the whole bucket in one request (see load_line64)
        load_line64(ix.edges + edge_base + 4 * (size_t)bk, ln);
        const uint4 e0 = ln.a0, e1 = ln.a1, e2 = ln.b0, e3 = ln.b1;
        if (e0.x == parent && e0.y == token) return e0.z & ~RE_OVERFLOW;
        if (e1.x == parent && e1.y == token) return e1.z & ~RE_OVERFLOW;
        if (e2.x == parent && e2.y == token) return e2.z & ~RE_OVERFLOW;
        if (e3.x == parent && e3.y == token) return e3.z & ~RE_OVERFLOW;
        if (e0.x == NONE || e1.x == NONE || e2.x == NONE || e3.x == NONE) return NONE;
        bk = (bk + 1) & mask;
    }
    return NONE;
}
__device__ __forceinline__ RNode load_rnode(const RetainIndexView& ix, uint32_t i) {
    const uint4 v = *reinterpret_cast<const uint4*>(ix.nodes + i);
    return RNode{v.x, v.y, v.z, v.w};
}

// DEEP = false: every filter of the batch, the per-level arrays in LDS (filters deeper than R_MAXL are listed and left empty);
// DEEP = true: the listed filters, the per-level arrays in global memory (launched only while batches hold such filters).
// OVONLY: the overlay trie alone (the bulk-loaded index is walked by k_retain_walk at the same time, on another stream): the filters that
// kernel answers, i.e. those of at most RW_LV levels; `a` carries the overlay's own range list
template <bool DEEP, bool OVONLY = false>
__device__ __forceinline__ void retain_walk_body(const RetainArgs& r, const BatchArgs& a) {
    __shared__ uint32_t s_lev_start[R_MAXL + 1], s_lev_end[R_MAXL + 1], s_ftok[R_MAXL + 1];
    __shared__ uint32_t fb0[R_FRONT], fc0[R_FRONT], fb1[R_FRONT], fc1[R_FRONT];
    __shared__ uint32_t sh[12];
    __shared__ uint32_t ob[R_OUT], oc[R_OUT];
    __shared__ uint32_t s_lh1[R_MAXL + 1], s_lh2[R_MAXL + 1], s_llen[R_MAXL + 1]; // level hashes: the overlay is keyed by them
    const uint32_t maxl = DEEP ? r.deep_maxl : R_MAXL;
    uint32_t* const gl = DEEP ? r.deep_levels + (size_t)blockIdx.x * 6 * ((size_t)r.deep_maxl + 1) : nullptr;
    uint32_t* const lev_start = DEEP ? gl : s_lev_start;
    uint32_t* const lev_end = DEEP ? gl + (maxl + 1) : s_lev_end;
    uint32_t* const ftok = DEEP ? gl + 2 * (maxl + 1) : s_ftok;
    uint32_t* const lh1 = DEEP ? gl + 3 * (maxl + 1) : s_lh1;
    uint32_t* const lh2 = DEEP ? gl + 4 * (maxl + 1) : s_lh2;
    uint32_t* const llen = DEEP ? gl + 5 * (maxl + 1) : s_llen;
    __shared__ uint32_t ovb[OV_OUT], ovs[OV_OUT];
    const RetainDynView dyn = r.ix.dyn;
    const uint32_t lane = threadIdx.x;
    uint2* gs = r.gscratch + (size_t)blockIdx.x * 2 * r.gcap;
    const uint32_t cap = R_FRONT + r.gcap;
    unsigned long long visits = 0, wranges = 0, wbytes = 0;

    const uint32_t n_work = DEEP ? min(a.ctr->slow_count, r.n_filters) : r.n_filters;
    for (uint32_t fi = blockIdx.x; fi < n_work; fi += gridDim.x) {
        const uint32_t f = DEEP ? r.deep_list[fi] : fi;
        // ---- tokenise the filter: '/' positions by ballot, then one lane per level for hash + dictionary ----------------
        const uint32_t beg = r.filter_off[f], end = r.filter_off[f + 1];
        uint32_t nsep = 0;
        if (lane == 0) lev_start[0] = beg;
        bool deep = false;
        for (uint32_t cb = beg; cb < end; cb += 64) {
            const uint32_t i = cb + lane;
            const bool sep = i < end && r.filters[i] == '/';
            const unsigned long long m = __ballot(sep);
            if (sep) {
                const uint32_t k = nsep + rank_below(m);
                if (k < maxl) {
                    lev_end[k] = i;
                    lev_start[k + 1] = i + 1;
                }
            }
            nsep += (uint32_t)__popcll(m);
        }
        const uint32_t nlev = nsep + 1;
        if (nlev > maxl || (OVONLY && nlev > RW_LV)) deep = true;
        if (lane == 0 && !deep) lev_end[nlev - 1] = end;
        if (DEEP) __threadfence_block(); // (the arrays live in global memory: other lanes read what this lane wrote)
        __syncthreads();
        RTenantSlot ten{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0}};
        if (!deep) {
            const uint8_t* fbytes = r.filters;
            auto fbyte = [&](uint32_t k) -> uint32_t { return fbytes[k]; };
            auto fword = [&](uint32_t k) -> uint32_t { return global_word_at(fbytes, k); };
            for (uint32_t lv = lane; lv < nlev; lv += 64) { // one lane per level (one round unless the filter is deep)
                uint32_t pos = lev_start[lv];
                const uint32_t start = pos, e = lev_end[lv];
                LevelHash h;
                uint32_t inl[4], len;
                bool last;
                scan_level(pos, e, false, fword, h, inl, len, last);
                uint32_t tok;
                if (len == 1 && inl[0] == '+') tok = RT_PLUS;
                else if (len == 1 && inl[0] == '#' && lv == nlev - 1) tok = RT_HASH;
                else {
                    DistIndexView dv{};
                    dv.dict = r.ix.dict;
                    dv.dict_group_mask = r.ix.dict_group_mask;
                    dv.pool = r.ix.pool;
                    tok = dict_lookup(dv, h, len, inl, start, fbyte);
                }
                ftok[lv] = tok;
                lh1[lv] = h.h1;
                lh2[lv] = h.h2;
                llen[lv] = len;
            }
            if (lane == 63) { // tenant -> root
                const uint32_t ti = r.filter_tenant[f];
                uint32_t tok = TOK_UNKNOWN;
                sh[9] = NONE; // the tenant's node in the overlay trie
                if (ti < r.n_tenants) {
                    const uint8_t* tb = r.tenants;
                    auto tbyte = [&](uint32_t k) -> uint32_t { return tb[k]; };
                    auto tword = [&](uint32_t k) -> uint32_t { return global_word_at(tb, k); };
                    uint32_t pos = r.tenant_off[ti];
                    const uint32_t start = pos, e = r.tenant_off[ti + 1];
                    LevelHash h;
                    uint32_t inl[4], len;
                    bool last;
                    scan_level(pos, e, false, tword, h, inl, len, last);
                    DistIndexView dv{};
                    dv.dict = r.ix.dict;
                    dv.dict_group_mask = r.ix.dict_group_mask;
                    dv.pool = r.ix.pool;
                    tok = dict_lookup(dv, h, len, inl, start, tbyte);
                    if (r.ix.dyn.ov_live) sh[9] = ov_find(r.ix.dyn.onodes, r.ix.dyn.oedges, r.ix.dyn.oedge_mask, r.ix.dyn.opool, 0u, h.h1, h.h2, len, tb, start);
                }
                uint32_t root = NONE;
                if (tok != TOK_UNKNOWN) {
                    uint32_t d = tenant_hash(tok) & r.ix.tenant_mask;
                    for (uint32_t probes = 0; probes <= r.ix.tenant_mask; probes++) {
                        const uint4* p = reinterpret_cast<const uint4*>(r.ix.tenants + d);
                        const uint4 t0 = p[0];
                        if (t0.x == tok) {
                            const uint4 t1 = p[1];
                            const uint4 t2 = p[2];
                            sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w;                   // node_base, edge_base, edge mask
                            sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;     // id_base, sys nodes lo/hi, sys id lo
                            sh[8] = t2.x;                                               // sys id hi
                            root = 0; // tenant-local id of the root
                            break;
                        }
                        if (t0.x == 0) break;
                        d = (d + 1) & r.ix.tenant_mask;
                    }
                }
                sh[0] = root;
            }
        }
        if (DEEP) __threadfence_block(); // (the level arrays in global memory: written by one lane each, read by all below)
        __syncthreads();
        const uint32_t root = (deep || OVONLY) ? NONE : sh[0];
        const uint32_t tnode = (deep || !dyn.ov_live) ? NONE : sh[9];
        ten.node_base = sh[1]; ten.edge_base = sh[2]; ten.edge_bucket_mask = sh[3]; ten.id_base = root != NONE ? sh[4] : 0u;
        ten.sys_node_lo = sh[5]; ten.sys_node_hi = sh[6]; ten.sys_id_lo = sh[7]; ten.sys_id_hi = sh[8];
        if (deep && lane == 0) {
            if (DEEP) atomicOr(&a.ctr->status, ST_RETAIN_DEEP); // (deeper than a 64 KB filter can be: cannot happen)
            else if (!OVONLY) r.deep_list[atomicAdd(&a.ctr->slow_count, 1u)] = f; // left empty here; the deep pass answers it
        }

        // ---- two passes: count, then write -------------------------------------------------------------------------------------
        unsigned long long base = 0;
        uint32_t np_total = 0, nr_total = 0;
        for (int pass = 0; pass < 2; pass++) {
            uint32_t wp = 0, nr = 0; // ranges / ids emitted so far (wave-uniform)
            const uint32_t vinc = pass == 0 ? 1u : 0u; // nodes touched are counted once, in the counting pass
            auto emit = [&](bool pred, uint32_t b, uint32_t c) {
                const unsigned long long m = __ballot(pred);
                uint32_t live = c;
                if (pred) { // b is a tenant-local topic rank: ids are global
                    const uint32_t p = wp + rank_below(m), g = ten.id_base + b;
                    if (pass == 1) a.pairs[base + p] = MatchRange{g, c};
                    else if (p < R_OUT) { // first walk: keep the ranges in LDS, most filters never need the second walk
                        ob[p] = g;
                        oc[p] = c;
                    }
                    // topics removed since the bulk load keep their ids: the range stays whole, its live count comes from the rank
                    // directory of the dead bitmap (two lookups), and the expansion skips the dead ids
                    if (dyn.use_dead && g < dyn.base_n) live = c - (dead_before(dyn.dead_bits, dyn.dead_rank, g + c) - dead_before(dyn.dead_bits, dyn.dead_rank, g));
                }
                wp += (uint32_t)__popcll(m);
                unsigned long long s = pred ? live : 0u;
                nr += (uint32_t)wave_sum_u64(s);
            };
            Frontier cur{fb0, fc0, gs}, nxt{fb1, fc1, gs + r.gcap};
            uint32_t ncur = 0;
            bool overflow = false;
            if (root != NONE) {
                if (lane == 0) cur.put(0, root, 1);
                ncur = 1;
            }
            __syncthreads();
            bool done = false;
            for (uint32_t l = 0; l < nlev && ncur && !done; l++) {
                const uint32_t kind = ftok[l];
                uint32_t nn = 0;
                if (kind == RT_HASH) { // last level: whole subtrees (the parent level matches too)
                    for (uint32_t i0 = 0; i0 < ncur; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        uint2 rg = make_uint2(0, 0);
                        if (i < ncur) rg = cur.get(i);
                        const bool mixed = __any(i < ncur && rg.y > 1); // keep frontier order: ids must come out ascending
                        if (!mixed) { // singletons in parallel
                            const bool one = i < ncur && rg.y == 1;
                            RNode n{0, 0, 0, 0};
                            if (one) n = load_rnode(r.ix, ten.node_base + rg.x);
                            if (l == 0) { // filter "#": everything of the tenant except what lies below '$' children
                                emit(one && ten.sys_id_lo > n.sub_begin && ten.sys_id_hi > ten.sys_id_lo, n.sub_begin, ten.sys_id_lo - n.sub_begin);
                                const uint32_t lo2 = ten.sys_id_hi > ten.sys_id_lo ? ten.sys_id_hi : n.sub_begin;
                                emit(one && n.sub_end > lo2, lo2, n.sub_end - lo2);
                            } else {
                                emit(one && n.sub_end > n.sub_begin, n.sub_begin, n.sub_end - n.sub_begin);
                            }
                            if (one) visits += vinc;
                        }
                        // a chunk holding ranges of several nodes: the wave streams its items one after the other
                        for (uint32_t src = 0; mixed && src < min(64u, ncur - i0); src++) {
                            const uint32_t b = __shfl(rg.x, (int)src), c = __shfl(rg.y, (int)src);
                            for (uint32_t j0 = 0; j0 < c; j0 += 64) {
                                const uint32_t j = j0 + lane;
                                RNode n{0, 0, 0, 0};
                                if (j < c) {
                                    n = load_rnode(r.ix, ten.node_base + b + j);
                                    visits += vinc;
                                }
                                emit(j < c && n.sub_end > n.sub_begin, n.sub_begin, n.sub_end - n.sub_begin);
                            }
                        }
                    }
                    done = true;
                    break;
                }
                if (kind == RT_PLUS) { // every child of every frontier node: a node range maps to ONE node range
                    for (uint32_t i0 = 0; i0 < ncur; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        uint32_t b1 = 0, c1 = 0, b2 = 0, c2 = 0;
                        if (i < ncur) {
                            const uint2 rg = cur.get(i);
                            const RNode first = load_rnode(r.ix, ten.node_base + rg.x);
                            const RNode last = rg.y > 1 ? load_rnode(r.ix, ten.node_base + rg.x + rg.y - 1) : first;
                            visits += 2 * vinc;
                            const uint32_t cb = first.child_begin, ce = last.child_begin + (last.child_count & ~RN_TERM);
                            if (l == 0 && ten.sys_node_hi > ten.sys_node_lo) { // skip the '$' children of the tenant root
                                b1 = cb; c1 = ten.sys_node_lo > cb ? ten.sys_node_lo - cb : 0;
                                b2 = ten.sys_node_hi; c2 = ce > ten.sys_node_hi ? ce - ten.sys_node_hi : 0;
                            } else {
                                b1 = cb; c1 = ce - cb;
                            }
                        }
                        const unsigned long long m1 = __ballot(c1 != 0), m2 = __ballot(c2 != 0);
                        const uint32_t p1 = nn + rank_below(m1), p2 = nn + (uint32_t)__popcll(m1) + rank_below(m2);
                        if (c1) { if (p1 < cap) nxt.put(p1, b1, c1); else overflow = true; }
                        if (c2) { if (p2 < cap) nxt.put(p2, b2, c2); else overflow = true; }
                        nn += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
                    }
                } else { // literal level: look every frontier node up in the edge hash
                    for (uint32_t i0 = 0; i0 < ncur && kind != TOK_UNKNOWN; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        uint2 rg = make_uint2(0, 0);
                        if (i < ncur) rg = cur.get(i);
                        const bool mixed = __any(i < ncur && rg.y > 1);
                        if (!mixed) {
                            uint32_t child = NONE;
                            if (i < ncur && rg.y == 1) {
                                child = redge_lookup(r.ix, ten.edge_base, ten.edge_bucket_mask, rg.x, kind);
                                visits += vinc;
                            }
                            const unsigned long long m = __ballot(child != NONE);
                            const uint32_t p = nn + rank_below(m);
                            if (child != NONE) { if (p < cap) nxt.put(p, child, 1); else overflow = true; }
                            nn += (uint32_t)__popcll(m);
                        }
                        for (uint32_t src = 0; mixed && src < min(64u, ncur - i0); src++) {
                            const uint32_t b = __shfl(rg.x, (int)src), c = __shfl(rg.y, (int)src);
                            for (uint32_t j0 = 0; j0 < c; j0 += 64) {
                                const uint32_t j = j0 + lane;
                                uint32_t child = NONE;
                                if (j < c) {
                                    child = redge_lookup(r.ix, ten.edge_base, ten.edge_bucket_mask, b + j, kind);
                                    visits += vinc;
                                }
                                const unsigned long long m = __ballot(child != NONE);
                                const uint32_t p = nn + rank_below(m);
                                if (child != NONE) { if (p < cap) nxt.put(p, child, 1); else overflow = true; }
                                nn += (uint32_t)__popcll(m);
                            }
                        }
                    }
                }
                if (__any(overflow)) {
                    if (lane == 0) atomicOr(&a.ctr->status, ST_RETAIN_FRONT);
                    nn = 0;
                    done = true;
                }
                __syncthreads();
                const Frontier t = cur;
                cur = nxt;
                nxt = t;
                ncur = nn;
            }
            if (!done) { // the filter ended without '#': topics that end exactly at a frontier node
                for (uint32_t i0 = 0; i0 < ncur; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    uint2 rg = make_uint2(0, 0);
                    if (i < ncur) rg = cur.get(i);
                    const bool mixed = __any(i < ncur && rg.y > 1);
                    if (!mixed) {
                        const bool one = i < ncur && rg.y == 1;
                        RNode n{0, 0, 0, 0};
                        if (one) {
                            n = load_rnode(r.ix, ten.node_base + rg.x);
                            visits += vinc;
                        }
                        emit(one && (n.child_count & RN_TERM), n.sub_begin, 1);
                    }
                    for (uint32_t src = 0; mixed && src < min(64u, ncur - i0); src++) {
                        const uint32_t b = __shfl(rg.x, (int)src), c = __shfl(rg.y, (int)src);
                        for (uint32_t j0 = 0; j0 < c; j0 += 64) {
                            const uint32_t j = j0 + lane;
                            RNode n{0, 0, 0, 0};
                            if (j < c) {
                                n = load_rnode(r.ix, ten.node_base + b + j);
                                visits += vinc;
                            }
                            emit(j < c && (n.child_count & RN_TERM), n.sub_begin, 1);
                        }
                    }
                }
            }
            __syncthreads();
            // ---- the overlay: topics added since the bulk load (bmq_retain_core.h).  A small hashed trie keyed (parent node, level
            // hash), labels verified in its string pool, walked with a frontier of single nodes; '+' and '#' follow child lists.  Its
            // topics carry ids above every bulk-loaded id, so ordering them among themselves keeps the row ascending. ----------------
            if (tnode != NONE) {
                Frontier ocur{fb0, fc0, gs}, onxt{fb1, fc1, gs + r.gcap};
                uint32_t on = 1, n_ov = 0;
                bool oflow = false;
                if (lane == 0) ocur.put(0, tnode, 1);
                __syncthreads();
                auto take = [&](bool pred, uint32_t id) { // a matched overlay topic (wave-uniform call sites only)
                    const unsigned long long m = __ballot(pred);
                    const uint32_t cnt = (uint32_t)__popcll(m);
                    if (n_ov + cnt > OV_OUT) { // the buffer is full: what it holds goes out unordered (k_sort_rows repairs the row)
                        for (uint32_t i0 = 0; i0 < n_ov; i0 += 64) {
                            const bool in = i0 + lane < n_ov;
                            emit(in, (in ? ovb[i0 + lane] : 0u) - ten.id_base, 1);
                        }
                        n_ov = 0;
                        wave_sync();
                    }
                    if (pred) ovb[n_ov + rank_below(m)] = id;
                    n_ov += cnt;
                };
                auto live_topic = [&](uint32_t node) -> uint32_t { // id of the topic retained at an overlay node, or NONE
                    const uint32_t id = dyn.onodes[node].topic_id;
                    return (id != NONE && !id_dead(dyn.dead_bits, id)) ? id : NONE;
                };
                auto push_children = [&](uint32_t node, bool skip_sys) { // per lane: the child list of one node into the next frontier
                    for (uint32_t c = dyn.onodes[node].first_child; c != NONE; c = dyn.onodes[c].next_sibling) {
                        if (skip_sys && (dyn.onodes[c].str_len & ON_SYS)) continue; // a first-level wildcard never matches a '$' level
                        const uint32_t p = atomicAdd(&sh[10], 1u);
                        if (p < cap) onxt.put(p, c, 1);
                        else oflow = true;
                    }
                };
                bool odone = false;
                for (uint32_t l = 0; l < nlev && on && !odone; l++) {
                    const uint32_t kind = ftok[l];
                    if (kind == RT_HASH) { // the frontier's nodes (from level 1 on: "a/#" matches "a") and everything below, level by level
                        bool first = true;
                        while (on) {
                            if (lane == 0) sh[10] = 0;
                            __syncthreads();
                            for (uint32_t i0 = 0; i0 < on; i0 += 64) {
                                const bool in = i0 + lane < on;
                                const uint32_t node = in ? ocur.get(i0 + lane).x : 0u;
                                uint32_t id = NONE;
                                if (in) {
                                    if (!(first && l == 0)) id = live_topic(node); // (at level 0 the first frontier is the tenant itself)
                                    visits += vinc;
                                    push_children(node, first && l == 0);
                                }
                                take(id != NONE, id);
                            }
                            __syncthreads();
                            on = min(sh[10], cap);
                            const Frontier t = ocur;
                            ocur = onxt;
                            onxt = t;
                            first = false;
                            __syncthreads();
                        }
                        odone = true;
                        break;
                    }
                    if (lane == 0) sh[10] = 0;
                    __syncthreads();
                    for (uint32_t i0 = 0; i0 < on; i0 += 64) {
                        if (i0 + lane >= on) continue;
                        const uint32_t node = ocur.get(i0 + lane).x;
                        visits += vinc;
                        if (kind == RT_PLUS) push_children(node, l == 0);
                        else {
                            const uint32_t c = ov_find(dyn.onodes, dyn.oedges, dyn.oedge_mask, dyn.opool, node, lh1[l], lh2[l], llen[l], r.filters, lev_start[l]);
                            if (c != NONE) {
                                const uint32_t p = atomicAdd(&sh[10], 1u);
                                if (p < cap) onxt.put(p, c, 1);
                                else oflow = true;
                            }
                        }
                    }
                    __syncthreads();
                    on = min(sh[10], cap);
                    const Frontier t = ocur;
                    ocur = onxt;
                    onxt = t;
                    __syncthreads();
                }
                if (!odone) // the filter ended without '#': topics that end exactly at a frontier node
                    for (uint32_t i0 = 0; i0 < on; i0 += 64) {
                        const bool in = i0 + lane < on;
                        uint32_t id = NONE;
                        if (in) {
                            id = live_topic(ocur.get(i0 + lane).x);
                            visits += vinc;
                        }
                        take(id != NONE, id);
                    }
                if (__any(oflow) && lane == 0) atomicOr(&a.ctr->status, ST_RETAIN_FRONT); // the batch is re-run with a larger frontier
                wave_sync();
                for (uint32_t j = lane; j < n_ov; j += 64) { // order what the buffer holds: rank of every id among the others
                    const uint32_t x = ovb[j];
                    uint32_t rk = 0;
                    for (uint32_t k = 0; k < n_ov; k++) rk += ovb[k] < x ? 1u : 0u;
                    ovs[rk] = x;
                }
                wave_sync();
                for (uint32_t i0 = 0; i0 < n_ov; i0 += 64) {
                    const bool in = i0 + lane < n_ov;
                    emit(in, (in ? ovs[i0 + lane] : 0u) - ten.id_base, 1);
                }
                __syncthreads();
            }
            if (pass == 0) {
                np_total = wp;
                nr_total = nr;
                uint32_t fits_l = 1;
                if (lane == 0 && wp) fits_l = pair_alloc(a.subs, a.pair_cap, f, wp, base) ? 1u : 0u;
                base = __shfl(base, 0);
                if (!__shfl(fits_l, 0)) {
                    if (lane == 0) atomicOr(&a.ctr->status, OVONLY ? (uint32_t)ST_NEED_SPILL : (uint32_t)ST_NEED_PAIRS); // (the overlay's list: the host grows that one)
                    break;
                }
                if (wp == 0) break;
                if (wp <= R_OUT) { // everything is in LDS: copy it out, no second walk
                    for (uint32_t i = lane; i < wp; i += 64) a.pairs[base + i] = MatchRange{ob[i], oc[i]};
                    break;
                }
            }
        }
        if (lane == 0) {
            a.pair_off[f] = (uint32_t)base;
            a.pair_cnt[f] = np_total;
            a.route_cnt[f] = nr_total;
            // (the per-block sums k_expand wants are added up by k_retain_sums)
        }
        wranges += np_total;
        if (!OVONLY && (DEEP || !deep)) wbytes += end - beg; // (a deep filter is counted by the pass that answers it)
        __syncthreads();
    }
    const unsigned long long wv = wave_sum_u64(visits);
    if (lane == 0) { // once per persistent wave
        if (wv) atomicAdd(&a.ctr->n_visit, wv);
        if (wranges) atomicAdd(&a.ctr->n_ranges, wranges);
        if (wbytes) atomicAdd(&a.ctr->topic_bytes, wbytes);
    }
}

// one filter per wave: the walk of the filters k_retain_walk lists (more than RW_LV levels: DEEP) and of the overlay trie (OVONLY).  (The
// round-2..4 instantiation that walked both tries for every filter -- k_retain_walk_v1 -- is gone: round 6.)
__global__ __launch_bounds__(64) void k_retain_walk_deep(RetainArgs r, BatchArgs a) { retain_walk_body<true>(r, a); }
__global__ __launch_bounds__(64) void k_retain_overlay(RetainArgs r, BatchArgs a) { retain_walk_body<false, true>(r, a); }

} // namespace bmq
#include "bmq_rwalk_kernel.h"
namespace bmq {

template <int G, bool DYN> __global__ __launch_bounds__(64, BMQ_RW_MIN_WAVES) void k_retain_walk(RetainArgs r, BatchArgs a) {
    __shared__ RwLds<G> L;
    retain_walk_rounds<G, DYN>(r, a, L);
}

// k_retain_sums: ids per 64-row block (wave_sums) and per 2^SUPER_SHIFT blocks (super_sums) from the rows' id counts, behind the walk
// kernels and in front of k_expand / k_retain_rowptr_dyn.  (Rounds 2-4: two atomics per filter inside the walk -- 100 k filters on the
// seven super-block words are 14 k serialised atomics per word, ~0.2 ms of L2 atomic-unit time that a faster walk would wait for.)
__global__ __launch_bounds__(64) void k_retain_sums(BatchArgs a, RetainOvList ov) {
    const uint32_t blk = blockIdx.x, t = (blk << a.tpw_shift) + threadIdx.x;
    const bool in = threadIdx.x < (1u << a.tpw_shift) && t < a.n_topics;
    const unsigned long long v = in ? (unsigned long long)a.route_cnt[t] + (ov.route_cnt ? ov.route_cnt[t] : 0u) : 0ull;
    const unsigned long long s = wave_sum_u64(v);
    if (threadIdx.x == 0) {
        a.wave_sums[blk] = s;
        if (s) atomicAdd(&a.super_sums[(size_t)(blk >> SUPER_SHIFT) * SUPER_STRIDE], s);
    }
}

// ------------------------------------------------------------------------------------------------------------
// RetainStoreCoProc.match(limit, now) without expanding anything (RS/RetainStoreCoProc.java:167-190): the reference walks the
// FULL match set and keeps the first `limit` messages that have not expired.  Its iteration order is that of a HashSet
// (UTIL/index/StrategySet.java:31), i.e. unspecified; here it is ascending topic id.  One wave per filter works on the
// filter's matched id RANGES (disjoint subtree intervals, straight from k_retain_walk): repeatedly take the range with the
// smallest first id at or behind the cursor and scan its ids 64 at a time for live ones (expire_at > now) until the quota is
// met.  With the default limit of 10 (Setting.java:77) a filter matching 5000 topics touches ~10 ids instead of 5000.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t LIM_FAST = 64; // per-filter limits up to this go through k_limit_select; larger ones through the full CSR
// Round 6: TWO range lists per filter -- the walk of the bulk-loaded index leaves one (BatchArgs), the overlay's walk, once topics were
// added since the load, another (RetainOvList) -- taken one after the other: overlay ids lie above every bulk-loaded id, so "ascending
// topic id" is the first list, then the second.  (Rounds 2-5 read ONE list, and match(limit, now) on an index with an overlay therefore
// ran the retired one-filter-per-wave walk that filled one list from both tries: k_retain_walk_v1, gone.)
__global__ __launch_bounds__(64) void k_limit_select(BatchArgs a, RetainOvList ov, const uint32_t* limit, const unsigned long long* expire_at,
                                                     unsigned long long now, uint32_t n, uint32_t* tmp_ids, uint32_t* kept, uint32_t* counts) {
    const uint32_t lane = threadIdx.x;
    const bool blocked = (a.ctr->status & (ST_NEED_PAIRS | ST_NEED_SPILL | ST_RETAIN_FRONT | ST_RETAIN_LIST | ST_RETAIN_DEEP)) != 0; // the walk is re-run anyway
    for (uint32_t f = blockIdx.x; f < n; f += gridDim.x) {
        const uint32_t lim = min(limit[f], LIM_FAST);
        uint32_t taken = 0;
        for (uint32_t list = 0; list < (ov.pairs ? 2u : 1u) && taken < lim; list++) {
            const MatchRange* const pairs = list ? ov.pairs : a.pairs;
            const uint32_t po = list ? ov.pair_off[f] : a.pair_off[f], np = blocked ? 0u : (list ? ov.pair_cnt[f] : a.pair_cnt[f]);
            uint32_t cursor = 0;
            while (taken < lim) {
                uint32_t bb = 0xFFFFFFFFu, bc = 0;
                for (uint32_t k = lane; k < np; k += 64) {
                    const MatchRange r = pairs[po + k];
                    if (r.begin >= cursor && r.begin < bb && r.count) {
                        bb = r.begin;
                        bc = r.count;
                    }
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const uint32_t ob = __shfl_xor(bb, d), oc = __shfl_xor(bc, d);
                    if (ob < bb) {
                        bb = ob;
                        bc = oc;
                    }
                }
                if (bb == 0xFFFFFFFFu) break;
                for (uint32_t o = 0; o < bc && taken < lim; o += 64) {
                    const uint32_t id = bb + o + lane;
                    const bool live = o + lane < bc && expire_at[id] > now;
                    const unsigned long long m = __ballot(live);
                    const uint32_t slot = taken + rank_below(m);
                    if (live && slot < lim) tmp_ids[(size_t)f * LIM_FAST + slot] = id;
                    taken = min(lim, taken + (uint32_t)__popcll(m));
                }
                cursor = bb + bc; // ranges of one filter are disjoint intervals
            }
        }
        if (lane == 0) {
            kept[f] = taken;
            counts[f] = blocked ? 0u : a.route_cnt[f] + (ov.route_cnt ? ov.route_cnt[f] : 0u);
        }
    }
}
// row_ptr = exclusive scan of kept[0 .. n] (kept[n] = 0, so row_ptr[n] = the total): a device-wide scan (hipcub, two launches).  Rounds 2-5: ONE
// workgroup whose threads each walked a chunk of ~100 consecutive entries -- uncoalesced, 173 us for 100 k filters, more than the selection itself.
// k_limit_total carries the total where the batch's counters expect it.
__global__ void k_limit_total(const uint32_t* row_ptr, uint32_t n, unsigned long long* total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = row_ptr[n];
}
__global__ __launch_bounds__(256) void k_limit_compact(const uint32_t* tmp_ids, const uint32_t* row_ptr, uint32_t n, uint32_t* out_ids) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t dst = row_ptr[i], c = row_ptr[i + 1] - dst;
    for (uint32_t k = 0; k < c; k++) out_ids[dst + k] = tmp_ids[(size_t)i * LIM_FAST + k];
}
// limits above LIM_FAST: the finished CSR (rows ascending) is filtered row by row -- kept[i] = number of the first ids of row i that
// are needed to collect limit[i] live ones; the live ones among them are compacted by k_limit_copy_live
__global__ __launch_bounds__(256) void k_limit_count_live(const uint32_t* row_ptr, const uint32_t* ids, const uint32_t* limit,
                                                          const unsigned long long* expire_at, unsigned long long now, uint32_t n, uint32_t* kept,
                                                          uint32_t* counts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = row_ptr[i], e = row_ptr[i + 1], lim = limit[i];
    uint32_t t = 0;
    for (uint32_t k = b; k < e && t < lim; k++) t += expire_at[ids[k]] > now ? 1u : 0u;
    kept[i] = t;
    counts[i] = e - b;
}
__global__ __launch_bounds__(256) void k_limit_copy_live(const uint32_t* row_ptr, const uint32_t* new_row_ptr, const uint32_t* ids,
                                                         const unsigned long long* expire_at, unsigned long long now, uint32_t n, uint32_t* out_ids) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = row_ptr[i], e = row_ptr[i + 1], dst = new_row_ptr[i], c = new_row_ptr[i + 1] - dst;
    uint32_t t = 0;
    for (uint32_t k = src; k < e && t < c; k++)
        if (expire_at[ids[k]] > now) out_ids[dst + t++] = ids[k];
}

// ------------------------------------------------------------------------------------------------------------
// k_retain_rowptr_dyn + k_retain_expand_dyn -- CSR of a retain batch when bulk-loaded ids have been removed since the load
// (RetainDynView.use_dead): the matched ranges still cover the removed ids (they keep their place in the id space), route_cnt holds the
// LIVE counts, and the expansion drops the dead ids on the way out.  First the row pointers (one wave per 64 rows: k_expand's blocking
// and row-base arithmetic), then ONE WAVE PER ROW streams the row's ranges 64 ids at a time, live lanes compacted by ballot: reads one
// bitmap word per 64 ids, writes whole lines.  (A first version let one wave work through its 64 rows one after the other: 5.1 ms for
// the C4 batch against 0.7 ms of k_expand -- rows of 5 000 ids want a wave each.)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_retain_rowptr_dyn(BatchArgs a, RetainOvList ov) {
    const uint32_t lane = threadIdx.x, blk = blockIdx.x;
    if (blk >= a.n_blocks) return;
    const uint32_t t = (blk << a.tpw_shift) + lane;
    const bool valid = t < a.n_topics;
    const uint32_t nr = valid ? a.route_cnt[t] + (ov.route_cnt ? ov.route_cnt[t] : 0u) : 0u;
    uint32_t wtotal;
    const uint32_t excl = wave_excl_scan(nr, lane, wtotal);
    unsigned long long wbase;
    {
        const uint32_t sb = blk >> SUPER_SHIFT;
        unsigned long long acc = 0;
        for (uint32_t i = lane; i < sb; i += 64) acc += a.super_sums[(size_t)i * SUPER_STRIDE];
        for (uint32_t i = (sb << SUPER_SHIFT) + lane; i < blk; i += 64) acc += a.wave_sums[i];
        wbase = wave_sum_u64(acc);
    }
    const unsigned long long row = wbase + excl, wend = wbase + wtotal;
    const bool range_err = wend >= 0xFFFFFFFFull, no_space = wend > a.out_capacity;
    if (blk == a.n_blocks - 1 && lane == 0) {
        a.ctr->total_ids = wend;
        *a.out_total = wend;
    }
    if ((range_err || no_space) && lane == 0) atomicOr(&a.ctr->status, range_err ? (uint32_t)ST_RANGE : (uint32_t)ST_NOSPACE);
    if (valid && !range_err) {
        a.out_row_ptr[t] = (uint32_t)row;
        if (t == a.n_topics - 1) a.out_row_ptr[a.n_topics] = (uint32_t)(row + nr);
    }
}
constexpr uint32_t RXD_WAVES = 4;  // rows per workgroup of k_retain_expand_dyn (independent waves)
constexpr uint32_t RXD_SHORT = 16; // ranges up to this length are written by ONE lane each (64 ranges per step), longer ones streamed by the wave
// One wave per row, 64 ranges per step: every lane takes a range and learns its LIVE length -- a single id from its bit, a longer range
// from the rank directory of the DEAD bitmap (two words + two popcounts) --, a wave scan turns the lengths into output offsets; short
// ranges -- a filter like a/+/c matches thousands of single topics -- are written by their lanes (the one or two bitmap words they span
// are read once), long ones (a '#' subtree) are streamed by the whole wave one 64-id WORD of the bitmap at a time: the word's address is
// wave-uniform (scalar loads, four words per request), its live bits are the lanes that store -- a dozen instructions per 64 ids.
// (Round 3: 64 ids per step through per-lane bitmap tests whatever the alignment, 1.4 ms for the churned C4 batch against 0.6 ms for the
// plain expansion; first version, one range after the other: 1.69 ms.)
__global__ __launch_bounds__(RXD_WAVES * 64) void k_retain_expand_dyn(BatchArgs a, RetainOvList ov, const unsigned long long* dead_bits,
                                                                      const uint32_t* dead_rank, uint32_t base_n) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t = blockIdx.x * RXD_WAVES + (threadIdx.x >> 6);
    if (t >= a.n_topics) return;
    // (the status word is complete: k_retain_rowptr_dyn ran in front)
    if (a.ctr->status & (ST_NEED_PAIRS | ST_NEED_SPILL | ST_RETAIN_FRONT | ST_RETAIN_DEEP | ST_RETAIN_LIST | ST_RANGE | ST_NOSPACE)) return;
    const uint32_t np1 = a.pair_cnt[t], np2 = ov.pair_cnt ? ov.pair_cnt[t] : 0u;
    const uint32_t nr = a.route_cnt[t] + (ov.route_cnt ? ov.route_cnt[t] : 0u);
    if (np1 + np2 == 0 || nr == 0) return;
    uint32_t* out = a.out_ids + a.out_row_ptr[t];
    uint32_t done = 0, carry_last = 0;
    bool bad = false, first_range = true;
    // the row's two range lists one after the other: the bulk-loaded ids, then the overlay's (above every bulk-loaded id)
    for (uint32_t part = 0; part < 2; part++) {
    const uint32_t np = part ? np2 : np1;
    if (np == 0) continue;
    const MatchRange* const prs = part ? ov.pairs + ov.pair_off[t] : a.pairs + a.pair_off[t];
    for (uint32_t k0 = 0; k0 < np; k0 += 64) {
        const uint32_t k = k0 + lane;
        const bool have = k < np;
        const MatchRange rg = have ? prs[k] : MatchRange{0u, 0u};
        // ranges out of order (overlay ids flushed unordered): the row is sorted afterwards
        const uint32_t last = rg.begin + rg.count - 1;
        uint32_t prev_last = __shfl_up(last, 1);
        if (lane == 0) prev_last = carry_last;
        if (have && rg.count && !(first_range && k == 0) && rg.begin <= prev_last) bad = true;
        carry_last = __shfl(last, (int)(min(np - k0, 64u) - 1u));
        // dead ids of the range: only bulk-loaded ids (< base_n) can be dead (overlay topics were checked by the walk); the words a SHORT
        // range spans are kept for the write below
        const bool is_long = have && rg.count > RXD_SHORT, is_short = have && rg.count != 0 && !is_long;
        const uint32_t lo = rg.begin < base_n ? rg.begin : base_n, hi = rg.begin + rg.count < base_n ? rg.begin + rg.count : base_n;
        uint32_t dead = 0;
        unsigned long long w0 = 0, w1 = 0;
        if (is_short && hi > lo) {
            w0 = dead_bits[lo >> 6];
            if (((hi - 1) >> 6) != (lo >> 6)) w1 = dead_bits[(lo >> 6) + 1];
            // bit i of `span`: id lo + i is dead (i < 16)
            const unsigned long long span = (w0 >> (lo & 63u)) | ((lo & 63u) ? (w1 << (64u - (lo & 63u))) : 0ull);
            w0 = span & ((1ull << (hi - lo)) - 1ull);
            dead = (uint32_t)__popcll(w0);
        } else if (is_long && hi > lo) dead = dead_before(dead_bits, dead_rank, hi) - dead_before(dead_bits, dead_rank, lo);
        const uint32_t live_n = rg.count - dead;
        uint32_t tot;
        const uint32_t excl = wave_excl_scan(have ? live_n : 0u, lane, tot);
        if (is_short && live_n) {
            uint32_t p = done + excl;
            for (uint32_t o = 0; o < rg.count; o++)
                if (!((w0 >> o) & 1ull)) out[p++] = rg.begin + o; // (ids from base_n on have no bit: w0 covers [lo, hi) only, o beyond it is live)
        }
        for (unsigned long long m_long = __ballot(is_long); m_long; m_long &= m_long - 1ull) {
            const int l = __ffsll((long long)m_long) - 1;
            const uint32_t b = sgpr(__shfl(rg.begin, l)), c = sgpr(__shfl(rg.count, l)), d = sgpr(__shfl(dead, l));
            uint32_t at = sgpr(done + __shfl(excl, l));
            if (d == 0) { // clean: nothing to look up
                for (uint32_t o = lane; o < c; o += 64) out[at + o] = b + o;
                continue;
            }
            // word by word: [b, e) meets the bitmap words wb .. we; ids from base_n on are live
            const uint32_t e = b + c, eb = e < base_n ? e : base_n;
            uint32_t id0 = b;
            if (b < eb) {
                const uint32_t wb = b >> 6, we = (eb - 1) >> 6;
                // the live lanes of one bitmap word store: `keep` = the word's ids inside the range that are not dead
                auto put_word = [&](uint32_t w, unsigned long long keep) {
                    if ((keep >> lane) & 1ull) out[at + rank_below(keep)] = (w << 6) + lane;
                    at += (uint32_t)__popcll(keep);
                };
                unsigned long long edge = ~0ull << (b & 63u); // the first word: from b on (and, if it is the last one too, up to eb)
                if (wb == we && (eb & 63u)) edge &= ~0ull >> (64u - (eb & 63u));
                put_word(wb, ~dead_bits[wb] & edge);
                uint32_t w = wb + 1;
                for (; w + 4 <= we; w += 4) { // whole words, four per request (a word's address is wave-uniform: scalar loads)
                    const unsigned long long d0 = dead_bits[w], d1 = dead_bits[w + 1], d2 = dead_bits[w + 2], d3 = dead_bits[w + 3];
                    put_word(w, ~d0), put_word(w + 1, ~d1), put_word(w + 2, ~d2), put_word(w + 3, ~d3);
                }
                for (; w < we; w++) put_word(w, ~dead_bits[w]);
                if (we > wb) put_word(we, ~dead_bits[we] & ((eb & 63u) ? ~0ull >> (64u - (eb & 63u)) : ~0ull)); // the last word: up to eb
                id0 = eb;
            }
            for (uint32_t o = id0 + lane; o < e; o += 64) out[at + (o - id0)] = o; // (the part of the range above the bulk-loaded ids)
        }
        done += tot;
    }
    first_range = false;
    }
    if (__any(bad) && nr > 1 && lane == 0) {
        const uint32_t sp = atomicAdd(&a.ctr->sort_count, 1u);
        if (sp < a.sort_cap) a.sort_list[sp] = t;
        else atomicOr(&a.ctr->status, ST_NEED_SORTLIST);
    }
}

} // namespace bmq
