// bmq_retain_core.h -- MUTATION of the retained-topic index where it lives (HBM), with STABLE topic ids: the per-item functions
// behind bmq_retain_apply* / bmq_retain_expired / bmq_retain_topic*.  Every function is BMQ_HD: bmq_exec_dev.h wraps them in gfx950
// kernels on the engine stream (one lane per op / per id), bmq_exec_host.h runs the very same code on host threads for host-only
// engines and the sanitizer fuzzers (tools/retain_fuzz.cpp) -- there is no second implementation.
//
// What it replaces: TopicLevelTrie.add / remove (UTIL/index/TopicLevelTrie.java:49-182: lock-free CAS on INode.main, CNode copies,
// tomb contraction) as RetainTopicIndex drives it (RS/index/RetainTopicIndex.java:126-134) from the post-commit closure of
// RetainStoreCoProc.batchRetain / gc (RS/RetainStoreCoProc.java:240-255,270-275).
//
// Design.  The bulk-loaded index (bmq_retain.h: topic id = rank, every subtree ONE id range, the children of a node range ONE
// node range) stays immutable between rebuilds -- that layout is what makes '+' and '#' O(1) per node range -- and is overlaid by
//   * a DEAD bitmap over the ids (+ a rank directory: dead ids in front of every 64-id word), so that a removed topic keeps its
//     id, matched id ranges give exact live counts in O(1) and the expansion just skips dead ids.  A re-added topic gets its id
//     back (the bit is cleared);
//   * an OVERLAY trie for topics the bulk load did not hold: nodes keyed (parent node, 64-bit level hash) in one open-addressing
//     table, labels verified byte for byte in a string pool, child lists for '+' / '#'; a new topic gets the next unused id.
//     One trie for all tenants, the tenant id is level 0 (as in the reference: RetainTopicIndex keys [tenantId, levels...]).
// An id therefore never changes and is never reused until the next bmq_retain_rebuild* / bmq_retain_compact (a new generation),
// which merges the overlay into a fresh bulk-loaded index.
//
// One batch of ops = three kernels in stream order with the match batches (which therefore never see half a batch):
//   locate   one lane per op (the adds, then the removes): find the topic in the bulk-loaded index (dictionary + edge hash:
//            O(levels) line fetches), else find -- for an add: find or create, lock-free CAS claims -- its overlay node; remember
//            the target and bid for it (64-bit atomic max of the op's position: the LAST op on a topic decides, so ops apply "in order")
//   commit   one lane per op: the winning op of every target sets / clears the dead bit, stores timestamp / expiry / expireAt,
//            hands out the id of a brand-new topic
//   rank     dead ids in front of every 64-id word (exclusive prefix popcount)
#pragma once
#include <stdint.h>

#include "bmq_build_core.h"
#include "bmq_retain.h"

namespace bmq {

#if defined(__HIP_DEVICE_COMPILE__)
template <class T> BMQ_HD T atom_max(T* p, T v) { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> BMQ_HD T atom_and(T* p, T v) { return __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> BMQ_HD T atom_xchg(T* p, T v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BMQ_HD void atom_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } // this lane's stores have left for memory
#else
template <class T> BMQ_HD T atom_max(T* p, T v) {
    T cur = __atomic_load_n(p, __ATOMIC_ACQUIRE);
    while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
    }
    return cur;
}
template <class T> BMQ_HD T atom_and(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_ACQ_REL); }
template <class T> BMQ_HD T atom_xchg(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_ACQ_REL); }
BMQ_HD void atom_drain() {}
#endif

constexpr uint32_t RT_OVERLAY = 0x80000000u; // op target: an overlay node (low bits) instead of a bulk-loaded id
constexpr uint32_t ID_SYS = 0x80000000u;     // id_tnode flag: the topic's first level starts with '$'
enum : uint32_t { RERR_NODES = 1u, RERR_POOL = 2u, RERR_IDS = 4u, RERR_STUCK = 8u, RERR_BAD_OP = 16u };

struct RetainCounters { // device memory; the host reads it back after every batch
    uint32_t ov_nodes;   // overlay nodes handed out (node 0 = the root)
    uint32_t opool_used; // bytes of the overlay string pool handed out
    uint32_t next_id;    // ids handed out (bulk-loaded + overlay, live or not)
    uint32_t err;
    // per batch, N_CTR_LANES copies each (a single hot word serialises in the L2 atomic unit: bmq_build_core.h)
    uint32_t went_live[N_CTR_LANES], went_dead[N_CTR_LANES];           // topics that became retained / stopped being retained
    uint32_t base_went_live[N_CTR_LANES], base_went_dead[N_CTR_LANES]; // ... among the bulk-loaded ids (the dead-aware paths switch on them)
};

// Everything the mutation kernels touch (device pointers; host pointers under HostExec).
struct RetainMut {
    RetainIndexView base; // the bulk-loaded index (not written here)
    uint32_t base_n;
    ONode* onodes;
    uint32_t ov_cap;
    uint32_t* oedges;
    uint32_t oedge_mask;
    uint8_t* opool;
    uint32_t opool_cap;
    // per id
    unsigned long long* expire_at; // ms; 0 while the id is dead (so that "expire_at > now" alone rejects removed topics)
    unsigned long long* ts;        // Message.timestamp (HLC) given to IRetainTopicIndex.add
    uint32_t* expiry;              // Message.expiryInterval; 0xFFFFFFFF with ts 0: never expires
    uint32_t* id_node;             // overlay node of an overlay id (NONE for bulk-loaded ids)
    uint32_t* id_tnode;            // overlay node of its tenant | ID_SYS (NONE for bulk-loaded ids)
    uint32_t id_cap;
    unsigned long long* dead_bits;
    uint32_t* dead_rank;
    unsigned long long* last_op;   // [id_cap + ov_cap] bids of the running batch: (batch sequence << 32) | (op position + 1)
    RetainCounters* ctr;
};

struct RetainOps {
    const uint8_t* tenants;       // packed tenant ids
    const uint32_t* tenant_off;   // [n_tenants + 1]
    uint32_t n_tenants;
    const uint32_t* op_tenant;    // [n] index into the tenant table; null: every op belongs to tenant 0
    const uint8_t* topics;        // packed topics (readable 16 bytes past the end)
    const uint32_t* topic_off;    // [n + 1]
    const uint8_t* op;            // [n] 0 = add, 1 = remove
    const unsigned long long* ts; // [n] or null
    const uint32_t* expiry;       // [n] or null (together with ts)
    uint32_t n;
    unsigned long long seq;       // batch sequence number (> 0, ascending)
    uint32_t* target;             // [n] out of locate: id, RT_OVERLAY | node, or NONE
    uint32_t* out_ids;            // [n] out of commit: the topic's id, or NONE (absent topic removed / op superseded by a later one)
};

// ------------------------------------------------------------------------------------------------------------
// the bulk-loaded index, read with plain loads (it is immutable between rebuilds)
// ------------------------------------------------------------------------------------------------------------
BMQ_HD uint32_t rdict_find(const RetainIndexView& ix, const LevelHash& h, uint32_t len, const uint32_t inl[4], const uint8_t* bytes,
                           unsigned long long start) {
    const uint32_t tag = level_hash_tag(h);
    uint32_t g = level_hash_slot(h, len) & ix.dict_group_mask;
    for (uint32_t probes = 0; probes <= ix.dict_group_mask; probes++) {
        bool free_slot = false;
        for (uint32_t j = 0; j < DICT_GROUP; j++) {
            const DictSlot& s = ix.dict[DICT_GROUP * (size_t)g + j];
            if (s.tag == 0) {
                free_slot = true;
                continue;
            }
            if (s.tag != tag || s.len != len || s.inl[0] != inl[0] || s.inl[1] != inl[1] || s.inl[2] != inl[2] || s.inl[3] != inl[3]) continue;
            bool eq = true;
            for (uint32_t i = 16; i < len && eq; i++) eq = ix.pool[s.pool_off + i] == bytes[start + i];
            if (eq) return s.token;
        }
        if (free_slot) return TOK_UNKNOWN;
        g = (g + 1) & ix.dict_group_mask;
    }
    return TOK_UNKNOWN;
}
BMQ_HD uint32_t redge_find(const RetainIndexView& ix, uint32_t edge_base, uint32_t mask, uint32_t parent, uint32_t token) {
    uint32_t bk = redge_bucket(parent, token, mask);
    for (uint32_t probes = 0; probes <= mask; probes++) {
        bool free_slot = false;
        for (uint32_t j = 0; j < 4; j++) {
            const REdge& e = ix.edges[edge_base + 4 * (size_t)bk + j];
            if (e.parent == parent && e.token == token) return e.child & ~RE_OVERFLOW;
            free_slot = free_slot || e.parent == NONE;
        }
        if (free_slot) return NONE;
        bk = (bk + 1) & mask;
    }
    return NONE;
}
BMQ_HD const RTenantSlot* rtenant_find(const RetainIndexView& ix, uint32_t token) {
    uint32_t d = tenant_hash(token) & ix.tenant_mask;
    for (uint32_t probes = 0; probes <= ix.tenant_mask; probes++) {
        const RTenantSlot& t = ix.tenants[d];
        if (t.token == token) return &t;
        if (t.token == 0) return nullptr;
        d = (d + 1) & ix.tenant_mask;
    }
    return nullptr;
}
// The next level of a '/'-separated topic: [pos, level end); pos moves behind the separator.  `more` = another level follows
// (TopicUtil.parse keeps empty levels, UTIL/TopicUtil.java:206-225: "a/" has two).
struct LevelScan {
    LevelHash h;
    uint32_t inl[4];
    uint32_t len;
    unsigned long long start;
    bool more;
};
BMQ_HD void next_level(const uint8_t* bytes, unsigned long long& pos, unsigned long long end, LevelScan& lv) {
    lv.start = pos;
    scan_level_bytes<0x2F2F2F2Fu>(bytes, pos, end, lv.h, lv.inl, lv.len);
    lv.more = pos < end;
    if (lv.more) pos++; // the '/'
}
// id of (tenant, topic) in the bulk-loaded index, or NONE
BMQ_HD uint32_t base_find(const RetainIndexView& ix, uint32_t base_n, const uint8_t* tenants, unsigned long long tb, unsigned long long te,
                          const uint8_t* topics, unsigned long long pb, unsigned long long pe) {
    if (base_n == 0) return NONE;
    LevelScan lv;
    unsigned long long pos = tb;
    lv.start = pos;
    scan_level_bytes<0u>(tenants, pos, te, lv.h, lv.inl, lv.len); // the whole id is ONE level (SEP4 = NUL bytes: an id holds none)
    if (pos != te) return NONE;
    const uint32_t ttok = rdict_find(ix, lv.h, lv.len, lv.inl, tenants, tb);
    if (ttok == TOK_UNKNOWN) return NONE;
    const RTenantSlot* ten = rtenant_find(ix, ttok);
    if (!ten) return NONE;
    uint32_t node = 0;
    pos = pb;
    for (;;) {
        next_level(topics, pos, pe, lv);
        const uint32_t tok = rdict_find(ix, lv.h, lv.len, lv.inl, topics, lv.start);
        if (tok == TOK_UNKNOWN) return NONE;
        node = redge_find(ix, ten->edge_base, ten->edge_bucket_mask, node, tok);
        if (node == NONE) return NONE;
        if (!lv.more) break;
    }
    const RNode& n = ix.nodes[ten->node_base + node];
    return (n.child_count & RN_TERM) ? ten->id_base + n.sub_begin : NONE;
}

// ------------------------------------------------------------------------------------------------------------
// the overlay trie
// ------------------------------------------------------------------------------------------------------------
BMQ_HD uint32_t ov_slot(uint32_t parent, uint32_t h1, uint32_t h2, uint32_t mask) {
    uint32_t x = (parent * 0x9E3779B1u) ^ h1 ^ rotl32(h2, 13);
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x & mask;
}
// child of `parent` labelled bytes[start, start + len), or NONE.  Read side of the MATCH kernels and of lookups between batches:
// plain loads (nothing writes while they run).
BMQ_HD uint32_t ov_find(const ONode* onodes, const uint32_t* oedges, uint32_t mask, const uint8_t* opool, uint32_t parent, uint32_t h1, uint32_t h2,
                        uint32_t len, const uint8_t* bytes, unsigned long long start) {
    uint32_t s = ov_slot(parent, h1, h2, mask);
    for (uint32_t probes = 0; probes <= mask; probes++) {
        const uint32_t e = oedges[s];
        if (e == 0) return NONE;
        const ONode& n = onodes[e];
        if (n.parent == parent && n.h1 == h1 && n.h2 == h2 && (n.str_len & ~ON_SYS) == len && bytes_equal(opool, n.str_off, bytes, start, len)) return e;
        s = (s + 1) & mask;
    }
    return NONE;
}
// The same from INSIDE a mutation batch: other lanes insert concurrently, so table entries are claimed by CAS and a node is
// complete in memory (shared_store + drain) before its index is published; insert = false only looks.
BMQ_HD uint32_t ov_child(const RetainMut& m, uint32_t parent, const LevelScan& lv, const uint8_t* bytes, bool insert) {
    uint32_t s = ov_slot(parent, lv.h.h1, lv.h.h2, m.oedge_mask), mine = NONE;
    for (uint32_t probes = 0; probes <= m.oedge_mask; probes++) {
        uint32_t e = atom_load(&m.oedges[s]);
        if (e == 0) {
            if (!insert) return NONE;
            if (mine == NONE) { // build the node first, publish it with the CAS
                mine = atom_add(&m.ctr->ov_nodes, 1u);
                if (mine >= m.ov_cap) {
                    atom_or(&m.ctr->err, (uint32_t)RERR_NODES);
                    return NONE;
                }
                uint32_t off = 0;
                if (lv.len) {
                    off = atom_add(&m.ctr->opool_used, lv.len);
                    if ((unsigned long long)off + lv.len > m.opool_cap) {
                        atom_or(&m.ctr->err, (uint32_t)RERR_POOL);
                        return NONE;
                    }
                    for (uint32_t i = 0; i < lv.len; i++) shared_store(m.opool + off + i, bytes[lv.start + i]);
                }
                ONode* n = m.onodes + mine;
                shared_store(&n->parent, parent);
                shared_store(&n->h1, lv.h.h1);
                shared_store(&n->h2, lv.h.h2);
                shared_store(&n->str_off, off);
                shared_store(&n->str_len, lv.len | ((lv.len && bytes[lv.start] == '$') ? ON_SYS : 0u));
                shared_store(&n->first_child, NONE);
                shared_store(&n->next_sibling, NONE);
                shared_store(&n->topic_id, NONE);
                atom_drain();
            }
            e = atom_cas(&m.oedges[s], 0u, mine);
            if (e == 0) { // ours: hang it into the parent's child list (read by later kernels only)
                const uint32_t old = atom_xchg(&m.onodes[parent].first_child, mine);
                shared_store(&m.onodes[mine].next_sibling, old);
                return mine;
            }
            // somebody claimed the slot in between: e is their node -- maybe the very label we are after
        }
        const ONode* n = m.onodes + e;
        const uint32_t sl = shared_load(&n->str_len) & ~ON_SYS;
        if (shared_load(&n->parent) == parent && shared_load(&n->h1) == lv.h.h1 && shared_load(&n->h2) == lv.h.h2 && sl == lv.len) {
            const uint32_t so = shared_load(&n->str_off);
            bool eq = true;
            for (uint32_t i = 0; i < sl && eq; i++) eq = shared_load(m.opool + so + i) == bytes[lv.start + i];
            if (eq) return e; // (a node built for nothing stays behind as garbage until the next rebuild: rare)
        }
        s = (s + 1) & m.oedge_mask;
    }
    atom_or(&m.ctr->err, (uint32_t)RERR_STUCK);
    return NONE;
}
BMQ_HD bool id_dead(const unsigned long long* dead_bits, uint32_t id) { return (dead_bits[id >> 6] >> (id & 63u)) & 1ull; }
// dead ids in [0, x)
BMQ_HD uint32_t dead_before(const unsigned long long* dead_bits, const uint32_t* dead_rank, uint32_t x) {
    const uint32_t w = x >> 6, b = x & 63u;
    return dead_rank[w] + (b ? (uint32_t)__builtin_popcountll(dead_bits[w] & ((1ull << b) - 1ull)) : 0u);
}

// ------------------------------------------------------------------------------------------------------------
// one batch: locate, commit
// ------------------------------------------------------------------------------------------------------------
// Two launches per batch: phase 0 places the adds (creating overlay nodes), phase 1 the removes -- a remove must find the node an
// add of the SAME batch created for its topic, whatever the lanes' timing.
BMQ_HD void rlocate_one(const RetainMut& m, const RetainOps& ob, uint32_t i, uint32_t phase) {
    if (ob.op[i] > 1) {
        if (phase == 0) {
            ob.target[i] = NONE;
            atom_or(&m.ctr->err, (uint32_t)RERR_BAD_OP);
        }
        return;
    }
    if (ob.op[i] != phase) return;
    ob.target[i] = NONE;
    const bool add = ob.op[i] == 0;
    const uint32_t ti = ob.op_tenant ? ob.op_tenant[i] : 0u;
    if (ti >= ob.n_tenants) {
        atom_or(&m.ctr->err, (uint32_t)RERR_BAD_OP);
        return;
    }
    const unsigned long long tb = ob.tenant_off[ti], te = ob.tenant_off[ti + 1], pb = ob.topic_off[i], pe = ob.topic_off[i + 1];
    uint32_t target = base_find(m.base, m.base_n, ob.tenants, tb, te, ob.topics, pb, pe);
    if (target == NONE) { // not a bulk-loaded topic: the overlay holds it, or will
        LevelScan lv;
        unsigned long long pos = tb;
        lv.start = pos;
        scan_level_bytes<0u>(ob.tenants, pos, te, lv.h, lv.inl, lv.len);
        lv.more = false;
        uint32_t node = ov_child(m, 0u, lv, ob.tenants, add);
        pos = pb;
        while (node != NONE) {
            next_level(ob.topics, pos, pe, lv);
            node = ov_child(m, node, lv, ob.topics, add);
            if (!lv.more) break;
        }
        if (node != NONE) target = RT_OVERLAY | node;
    }
    ob.target[i] = target;
    if (target != NONE) {
        const size_t slot = (target & RT_OVERLAY) ? (size_t)m.id_cap + (target & ~RT_OVERLAY) : (size_t)target;
        atom_max(&m.last_op[slot], (ob.seq << 32) | (unsigned long long)(i + 1u));
    }
}

BMQ_HD void rcommit_one(const RetainMut& m, const RetainOps& ob, uint32_t i) {
    ob.out_ids[i] = NONE;
    const uint32_t target = ob.target[i];
    if (target == NONE) return; // removal of a topic nobody retains (or a refused op)
    const bool ov = (target & RT_OVERLAY) != 0;
    const uint32_t node = target & ~RT_OVERLAY;
    const size_t slot = ov ? (size_t)m.id_cap + node : (size_t)target;
    if (m.last_op[slot] != ((ob.seq << 32) | (unsigned long long)(i + 1u))) return; // a later op of this batch decides for the topic
    const bool add = ob.op[i] == 0;
    uint32_t id = target;
    if (ov) {
        id = m.onodes[node].topic_id;
        if (id == NONE) {
            if (!add) return; // the node is only a level on the way to other topics
            id = atom_add(&m.ctr->next_id, 1u);
            if (id >= m.id_cap) {
                atom_or(&m.ctr->err, (uint32_t)RERR_IDS);
                return;
            }
            m.onodes[node].topic_id = id;
            m.id_node[id] = node;
            uint32_t t = node, first = node; // the tenant's node and the topic's first level
            while (m.onodes[t].parent != 0u) {
                first = t;
                t = m.onodes[t].parent;
            }
            m.id_tnode[id] = t | ((m.onodes[first].str_len & ON_SYS) && first != t ? ID_SYS : 0u);
        }
    }
    const unsigned long long bit = 1ull << (id & 63u);
    const bool was_dead = (atom_load(&m.dead_bits[id >> 6]) & bit) != 0; // only this lane touches this id's BIT (others: the same word)
    const uint32_t c = i & (N_CTR_LANES - 1);
    if (add) {
        if (was_dead) {
            atom_and(&m.dead_bits[id >> 6], ~bit);
            atom_add(&m.ctr->went_live[c], 1u);
            if (!ov) atom_add(&m.ctr->base_went_live[c], 1u);
        }
        const bool has = ob.ts != nullptr;
        const unsigned long long ts = has ? ob.ts[i] : 0ull;
        const uint32_t ex = has ? ob.expiry[i] : 0xFFFFFFFFu;
        m.ts[id] = ts;
        m.expiry[id] = ex;
        m.expire_at[id] = (ts == 0 && ex == 0xFFFFFFFFu) ? RETAIN_NEVER : retain_expire_at(ts, ex);
    } else if (!was_dead) {
        atom_or(&m.dead_bits[id >> 6], bit);
        atom_add(&m.ctr->went_dead[c], 1u);
        if (!ov) atom_add(&m.ctr->base_went_dead[c], 1u);
        m.expire_at[id] = 0ull;
    }
    ob.out_ids[i] = id;
}

// re-insert overlay node i into a larger edge table (all nodes in parallel, after the table was cleared)
BMQ_HD void ov_rehash_one(const RetainMut& m, uint32_t i) {
    if (i == 0) return;
    const ONode& n = m.onodes[i];
    uint32_t s = ov_slot(n.parent, n.h1, n.h2, m.oedge_mask);
    for (uint32_t probes = 0; probes <= m.oedge_mask; probes++) {
        if (atom_cas(&m.oedges[s], 0u, i) == 0u) return;
        s = (s + 1) & m.oedge_mask;
    }
    atom_or(&m.ctr->err, (uint32_t)RERR_STUCK);
}

// ------------------------------------------------------------------------------------------------------------
// id -> retained topic, GC scan
// ------------------------------------------------------------------------------------------------------------
// overlay id -> bytes of tenant id + topic: lens[2 i] = tenant length, lens[2 i + 1] = total length (0 / 0: not an overlay topic)
BMQ_HD void ov_topic_len_one(const RetainMut& m, const uint32_t* ids, uint32_t i, uint32_t* lens) {
    const uint32_t id = ids[i];
    lens[2 * i] = lens[2 * i + 1] = 0;
    if (id < m.base_n || id >= m.id_cap || m.id_node[id] == NONE) return;
    uint32_t total = 0, levels = 0, t = m.id_node[id];
    while (m.onodes[t].parent != 0u) {
        total += m.onodes[t].str_len & ~ON_SYS;
        levels++;
        t = m.onodes[t].parent;
    }
    const uint32_t tl = m.onodes[t].str_len & ~ON_SYS;
    lens[2 * i] = tl;
    lens[2 * i + 1] = tl + total + (levels ? levels - 1 : 0);
}
BMQ_HD void ov_topic_write_one(const RetainMut& m, const uint32_t* ids, uint32_t i, const unsigned long long* offs, uint8_t* out) {
    const uint32_t id = ids[i];
    unsigned long long end = offs[i + 1];
    if (end == offs[i]) return;
    uint32_t t = m.id_node[id];
    for (;;) { // back to front: the last level first
        const ONode& n = m.onodes[t];
        const uint32_t l = n.str_len & ~ON_SYS;
        end -= l;
        for (uint32_t k = 0; k < l; k++) out[end + k] = m.opool[n.str_off + k];
        if (n.parent == 0u) break; // that was the tenant id
        if (m.onodes[n.parent].parent != 0u) out[--end] = '/';
        t = n.parent;
    }
}
// The scan of RetainStoreCoProc.gc (RS/RetainStoreCoProc.java:257-277): flag[id] = 1 for every retained id whose message has
// expired at `now` (expireTime <= now; override_expiry >= 0 replaces the stored interval: GCRequest.expirySeconds).
// tenant scan (has_tenant): only ids of that tenant -- bulk-loaded ranks [t_lo, t_hi), overlay ids below tenant node t_node -- and,
// because the reference obtains them with index.match(tenantId, "#"), NOT the topics whose first level starts with '$'
// (bulk-loaded ranks [sys_lo, sys_hi), overlay ids flagged ID_SYS).  live_only = list every retained id (findAll).
struct GcQuery {
    unsigned long long now;
    long long override_expiry; // < 0: none
    uint32_t has_tenant, t_lo, t_hi, sys_lo, sys_hi, t_node;
    uint32_t live_only;
    uint32_t skip_sys;         // tenant scan: leave out the topics whose first level starts with '$' (what match(tenant, "#") cannot reach)
    uint32_t n_ids;            // ids handed out
};
BMQ_HD uint32_t gc_flag_one(const RetainMut& m, const GcQuery& q, uint32_t id) {
    if (id >= q.n_ids || id_dead(m.dead_bits, id)) return 0u;
    if (q.has_tenant) {
        if (id < m.base_n) {
            if (id < q.t_lo || id >= q.t_hi || (q.skip_sys && id >= q.sys_lo && id < q.sys_hi)) return 0u;
        } else {
            const uint32_t tn = m.id_tnode[id];
            if ((tn & ~ID_SYS) != q.t_node || (q.skip_sys && (tn & ID_SYS))) return 0u;
        }
    }
    if (q.live_only) return 1u;
    unsigned long long at = m.expire_at[id];
    if (q.override_expiry >= 0) at = retain_expire_at(m.ts[id], (uint32_t)(q.override_expiry > 0xFFFFFFFFll ? 0xFFFFFFFFll : q.override_expiry));
    return at <= q.now ? 1u : 0u;
}

} // namespace bmq
