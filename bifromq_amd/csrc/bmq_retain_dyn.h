// bmq_retain_dyn.h -- control of the MUTABLE part of the retained-topic index (bmq_retain_core.h): owns the per-id arrays, the dead
// bitmap and the overlay trie in exec memory (HBM under DevExec, host memory under HostExec), sizes them ahead of every batch so that
// no lane can run out of room, and drives the locate / commit / rank kernels.  The bulk-loaded part (RetainIndexHost, bmq_retain.h)
// is handed in by the engine after every (re)build.
#pragma once
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "bmq_retain_core.h"

namespace bmq {

struct RetainDynInfo {
    uint64_t n_live = 0;      // retained topics now
    uint64_t id_bound = 0;    // ids handed out in this generation (live or not): every id is below
    uint64_t base_n = 0;      // ids below are ranks of the last bulk load
    uint64_t base_dead = 0;   // bulk-loaded ids whose topic has been removed
    uint64_t overlay_ids = 0; // ids handed out to topics added since the bulk load
    uint64_t overlay_nodes = 0;
    uint64_t batches = 0;
};

template <class Exec> class RetainDyn {
public:
    explicit RetainDyn(Exec& exec) : x(exec) {}
    ~RetainDyn() { drop(); }
    RetainDyn(const RetainDyn&) = delete;
    RetainDyn& operator=(const RetainDyn&) = delete;

    Exec& x;
    std::string error;
    bool tiny = false; // test knob: minimal capacities, every growth path runs all the time
    RetainDynInfo info;
    bool ready = false;

    // ---- after a bulk load: ids 0 .. n-1 are live ranks; everything dynamic starts empty ----
    bool reset(const RetainIndexView& base_view, const RetainIndexHost& h) {
        error.clear();
        drop();
        base = base_view;
        base_n = (uint32_t)h.n_topics;
        const uint32_t want_ids = base_n + (tiny ? 4u : std::max<uint32_t>(1024u, base_n / 8));
        if (!alloc_ids(want_ids)) return false;
        // per-id payload of the bulk load
        std::vector<unsigned long long> ts(base_n ? base_n : 1), ex_at(base_n ? base_n : 1);
        std::vector<uint32_t> ex(base_n ? base_n : 1);
        for (const RTenantState* t : h.order)
            for (size_t i = 0; i < t->topics.size(); i++) {
                ts[t->id_base + i] = t->ts[i];
                ex[t->id_base + i] = t->expiry[i];
                ex_at[t->id_base + i] = h.expire_at[t->id_base + i];
            }
        if (!x.copy_in_async(d_ts, ts.data(), sizeof(unsigned long long) * base_n) || !x.copy_in_async(d_expiry, ex.data(), sizeof(uint32_t) * base_n) ||
            !x.copy_in_async(d_expire_at, ex_at.data(), sizeof(unsigned long long) * base_n))
            return xfail();
        // dead bits: bulk-loaded ids live, everything behind them "not retained" until an add hands the id out
        {
            std::vector<unsigned long long> bits(n_words() + 1, ~0ull);
            for (uint32_t w = 0; w < base_n / 64; w++) bits[w] = 0ull;
            if (base_n & 63u) bits[base_n / 64] = ~0ull << (base_n & 63u);
            if (!x.copy_in_async(d_dead, bits.data(), sizeof(unsigned long long) * (n_words() + 1))) return xfail();
        }
        if (!alloc_overlay(tiny ? 4u : 4096u, tiny ? 16u : 1u << 16)) return false;
        RetainCounters c{};
        c.ov_nodes = 1; // the root
        c.next_id = base_n;
        ONode root{NONE, 0, 0, 0, 0, NONE, NONE, NONE};
        if (!x.copy_in_async(d_nodes, &root, sizeof(root)) || !x.copy_in_async(d_ctr, &c, sizeof(c))) return xfail();
        hc = c;
        if (!x.r_rank(mut(), n_words()) || !x.sync()) return xfail();
        info = RetainDynInfo{};
        info.n_live = base_n;
        info.id_bound = base_n;
        info.base_n = base_n;
        seq = 0;
        ready = true;
        return true;
    }

    RetainDynView view() const {
        RetainDynView v{};
        v.onodes = d_nodes;
        v.oedges = d_edges;
        v.oedge_mask = edge_slots - 1;
        v.opool = d_pool;
        v.dead_bits = d_dead;
        v.dead_rank = d_rank;
        v.base_n = base_n;
        v.use_dead = info.base_dead != 0;
        v.ov_live = (uint32_t)info.overlay_ids;
        return v;
    }
    const unsigned long long* expire_at() const { return d_expire_at; }

    // ---- one batch of ops (host buffers).  out_ids (may be null): the id of every op's topic, NONE for a no-op / superseded op ----
    bool apply(const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* op_tenant, const uint8_t* topics,
               const uint32_t* topic_off, const uint8_t* op, const unsigned long long* ts, const uint32_t* expiry, uint32_t n, uint32_t* out_ids) {
        error.clear();
        if (!ready) return fail("the retained-topic index has not been built");
        if (n == 0) return true;
        const size_t tb = tenant_off[n_tenants], pb = topic_off[n];
        // room for the worst case: every op adds a topic none of whose levels exists yet
        const size_t seps = (size_t)std::count(topics, topics + pb, (uint8_t)'/');
        uint64_t max_tenant_bytes = 0;
        for (uint32_t t = 0; t < n_tenants; t++) max_tenant_bytes = std::max<uint64_t>(max_tenant_bytes, tenant_off[t + 1] - tenant_off[t]);
        const uint64_t need_nodes = (uint64_t)hc.ov_nodes + seps + 2ull * n, need_pool = (uint64_t)hc.opool_used + pb + max_tenant_bytes * n;
        const uint64_t need_ids = (uint64_t)hc.next_id + n;
        if (need_nodes >= 0x7FFFFFF0ull || need_pool >= 0xFFFFFFF0ull || need_ids >= 0x7FFFFFF0ull) return fail("the retained-topic index is full: compact it");
        if (need_ids > id_cap && !grow_ids((uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, need_ids + need_ids / 2))) return false;
        if ((need_nodes > ov_cap || need_pool > pool_cap || need_nodes * 2 > edge_slots) &&
            !grow_overlay((uint32_t)std::max<uint64_t>(need_nodes + need_nodes / 2, ov_cap), (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, std::max<uint64_t>(need_pool + need_pool / 2, pool_cap))))
            return false;
        // staging
        const size_t off_t = 0, off_to = align16(off_t + tb + 16), off_ot = align16(off_to + 4 * ((size_t)n_tenants + 1)), off_p = align16(off_ot + 4 * (size_t)n),
                     off_po = align16(off_p + pb + 16), off_op = align16(off_po + 4 * ((size_t)n + 1)), off_ts = align16(off_op + n), off_ex = align16(off_ts + 8 * (size_t)n),
                     off_tg = align16(off_ex + 4 * (size_t)n), off_id = align16(off_tg + 4 * (size_t)n), total = align16(off_id + 4 * (size_t)n);
        if (total > stage_cap) {
            if (!x.sync()) return xfail();
            x.release(d_stage);
            d_stage = (uint8_t*)x.alloc(total + total / 4);
            stage_cap = d_stage ? total + total / 4 : 0;
            if (!d_stage) return fail("out of memory (retain op staging)");
        }
        if (!x.copy_in_async(d_stage + off_t, tenants, tb) || !x.copy_in_async(d_stage + off_to, tenant_off, 4 * ((size_t)n_tenants + 1)) ||
            (op_tenant && !x.copy_in_async(d_stage + off_ot, op_tenant, 4 * (size_t)n)) || !x.copy_in_async(d_stage + off_p, topics, pb) ||
            !x.copy_in_async(d_stage + off_po, topic_off, 4 * ((size_t)n + 1)) || !x.copy_in_async(d_stage + off_op, op, n) ||
            (ts && (!x.copy_in_async(d_stage + off_ts, ts, 8 * (size_t)n) || !x.copy_in_async(d_stage + off_ex, expiry, 4 * (size_t)n))))
            return xfail();
        RetainOps ob{};
        ob.tenants = d_stage + off_t;
        ob.tenant_off = (const uint32_t*)(d_stage + off_to);
        ob.n_tenants = n_tenants;
        ob.op_tenant = op_tenant ? (const uint32_t*)(d_stage + off_ot) : nullptr;
        ob.topics = d_stage + off_p;
        ob.topic_off = (const uint32_t*)(d_stage + off_po);
        ob.op = d_stage + off_op;
        ob.ts = ts ? (const unsigned long long*)(d_stage + off_ts) : nullptr;
        ob.expiry = ts ? (const uint32_t*)(d_stage + off_ex) : nullptr;
        ob.n = n;
        ob.seq = ++seq;
        ob.target = (uint32_t*)(d_stage + off_tg);
        ob.out_ids = (uint32_t*)(d_stage + off_id);
        // the per-batch counters start at zero (the persistent ones stay)
        if (!x.zero((uint8_t*)d_ctr + offsetof(RetainCounters, went_live), sizeof(RetainCounters) - offsetof(RetainCounters, went_live))) return xfail();
        const RetainMut m = mut();
        if (!x.r_locate(m, ob) || !x.r_commit(m, ob) || !x.r_rank(m, n_words())) return xfail();
        if (!x.copy_out(&hc, d_ctr, sizeof(hc))) return xfail(); // waits for the batch
        if (out_ids && !x.copy_out(out_ids, ob.out_ids, 4 * (size_t)n)) return xfail();
        if (hc.err) {
            const uint32_t err = hc.err;
            uint32_t zero32 = 0;
            (void)x.copy_in((uint8_t*)d_ctr + offsetof(RetainCounters, err), &zero32, 4);
            hc.err = 0;
            if (err & RERR_BAD_OP) return fail("malformed retain op (op code / tenant index)");
            return fail("retain overlay ran out of room despite the bound (" + std::to_string(err) + ")");
        }
        uint64_t live = 0, dead = 0, blive = 0, bdead = 0;
        for (uint32_t k = 0; k < N_CTR_LANES; k++) live += hc.went_live[k], dead += hc.went_dead[k], blive += hc.base_went_live[k], bdead += hc.base_went_dead[k];
        info.n_live += live;
        info.n_live -= dead;
        info.base_dead += bdead;
        info.base_dead -= blive;
        info.id_bound = hc.next_id;
        info.overlay_ids = hc.next_id - base_n;
        info.overlay_nodes = hc.ov_nodes - 1;
        info.batches++;
        return true;
    }

    // ---- id -> tenant id + topic of overlay topics (bulk-loaded ids are the host's: RetainIndexHost::topic) ----
    // lens[2 i] = tenant length, lens[2 i + 1] = total length (0: no such overlay topic), bytes = the strings back to back
    bool overlay_topics(const uint32_t* ids, uint32_t n, std::vector<uint32_t>& lens, std::vector<uint8_t>& bytes) {
        lens.assign(2 * (size_t)n, 0);
        bytes.clear();
        if (n == 0) return true;
        if (!ready) return fail("the retained-topic index has not been built");
        const size_t off_len = align16(4 * (size_t)n), off_off = align16(off_len + 8 * (size_t)n);
        if (!ensure(q_buf, q_cap, off_off + 8 * ((size_t)n + 1))) return false;
        uint32_t* d_ids = (uint32_t*)q_buf;
        uint32_t* d_lens = (uint32_t*)(q_buf + off_len);
        unsigned long long* d_offs = (unsigned long long*)(q_buf + off_off);
        const RetainMut m = mut();
        if (!x.copy_in_async(d_ids, ids, 4 * (size_t)n) || !x.r_topic_lens(m, d_ids, n, d_lens) || !x.copy_out(lens.data(), d_lens, 8 * (size_t)n)) return xfail();
        std::vector<unsigned long long> offs((size_t)n + 1, 0);
        for (uint32_t i = 0; i < n; i++) offs[i + 1] = offs[i] + lens[2 * i + 1];
        bytes.resize(offs[n]);
        if (offs[n] == 0) return true;
        if (!ensure(o_buf, o_cap, offs[n] + 16)) return false;
        if (!x.copy_in_async(d_offs, offs.data(), 8 * ((size_t)n + 1)) || !x.r_topic_write(m, d_ids, n, d_offs, o_buf) || !x.copy_out(bytes.data(), o_buf, offs[n]))
            return xfail();
        return true;
    }
    bool topic_info(uint32_t id, unsigned long long& ts, uint32_t& expiry, unsigned long long& expire_at_ms, bool& live) {
        if (!ready || id >= info.id_bound) return fail("no such retained topic");
        unsigned long long w = 0;
        if (!x.copy_out(&ts, d_ts + id, 8) || !x.copy_out(&expiry, d_expiry + id, 4) || !x.copy_out(&expire_at_ms, d_expire_at + id, 8) || !x.copy_out(&w, d_dead + (id >> 6), 8))
            return xfail();
        live = !((w >> (id & 63u)) & 1ull);
        return true;
    }
    // ids (ascending) the query accepts (GC scan / findAll); tenant_name: resolves GcQuery.t_node when q.has_tenant
    bool select(GcQuery q, const uint8_t* tenant_name, uint32_t tenant_len, std::vector<uint32_t>& out) {
        out.clear();
        if (!ready) return fail("the retained-topic index has not been built");
        q.n_ids = (uint32_t)info.id_bound;
        if (q.n_ids == 0) return true;
        const size_t off_cnt = align16((size_t)q.n_ids), off_name = align16(off_cnt + 16), off_ids = align16(off_name + tenant_len + 32);
        if (!ensure(q_buf, q_cap, off_ids + 4 * (size_t)q.n_ids)) return false;
        uint32_t* d_cnt = (uint32_t*)(q_buf + off_cnt);
        const RetainMut m = mut();
        if (q.has_tenant) {
            std::vector<uint8_t> padded((size_t)tenant_len + 16, 0);
            if (tenant_len) memcpy(padded.data(), tenant_name, tenant_len);
            if (!x.copy_in_async(q_buf + off_name, padded.data(), padded.size()) || !x.r_find_tenant(m, q_buf + off_name, tenant_len, d_cnt + 1) ||
                !x.copy_out(&q.t_node, d_cnt + 1, 4))
                return xfail();
        }
        uint32_t cnt = 0;
        if (!x.r_gc_select(m, q, q_buf, (uint32_t*)(q_buf + off_ids), d_cnt) || !x.copy_out(&cnt, d_cnt, 4)) return xfail();
        out.resize(cnt);
        if (cnt && !x.copy_out(out.data(), q_buf + off_ids, 4 * (size_t)cnt)) return xfail();
        return true;
    }

    // timestamp / expiry interval of every id handed out (compaction re-loads the live ones)
    bool payload(std::vector<unsigned long long>& ts, std::vector<uint32_t>& expiry) {
        ts.assign(info.id_bound ? info.id_bound : 1, 0);
        expiry.assign(info.id_bound ? info.id_bound : 1, 0);
        if (!ready) return fail("the retained-topic index has not been built");
        if (!x.copy_out(ts.data(), d_ts, 8 * (size_t)info.id_bound) || !x.copy_out(expiry.data(), d_expiry, 4 * (size_t)info.id_bound)) return xfail();
        return true;
    }

    void drop() {
        for (void* p : {(void*)d_expire_at, (void*)d_ts, (void*)d_expiry, (void*)d_id_node, (void*)d_id_tnode, (void*)d_dead, (void*)d_rank, (void*)d_last,
                        (void*)d_nodes, (void*)d_edges, (void*)d_pool, (void*)d_ctr, (void*)d_stage, (void*)q_buf, (void*)o_buf})
            if (p) x.release(p);
        d_expire_at = d_ts = nullptr;
        d_expiry = d_id_node = d_id_tnode = nullptr;
        d_dead = nullptr;
        d_rank = nullptr;
        d_last = nullptr;
        d_nodes = nullptr;
        d_edges = nullptr;
        d_pool = nullptr;
        d_ctr = nullptr;
        d_stage = q_buf = o_buf = nullptr;
        id_cap = ov_cap = pool_cap = edge_slots = 0;
        stage_cap = q_cap = o_cap = 0;
        ready = false;
    }

private:
    RetainIndexView base{};
    uint32_t base_n = 0;
    unsigned long long *d_expire_at = nullptr, *d_ts = nullptr;
    uint32_t *d_expiry = nullptr, *d_id_node = nullptr, *d_id_tnode = nullptr;
    unsigned long long* d_dead = nullptr;
    uint32_t* d_rank = nullptr;
    unsigned long long* d_last = nullptr;
    ONode* d_nodes = nullptr;
    uint32_t* d_edges = nullptr;
    uint8_t* d_pool = nullptr;
    RetainCounters* d_ctr = nullptr;
    RetainCounters hc{};
    uint32_t id_cap = 0, ov_cap = 0, pool_cap = 0, edge_slots = 0;
    uint8_t *d_stage = nullptr, *q_buf = nullptr, *o_buf = nullptr;
    size_t stage_cap = 0, q_cap = 0, o_cap = 0;
    unsigned long long seq = 0;

    static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
    uint32_t n_words() const { return (id_cap + 63) / 64; }
    bool fail(const std::string& m) {
        error = m;
        return false;
    }
    bool xfail() { return fail(x.err.empty() ? std::string("exec failure") : x.err); }
    bool ensure(uint8_t*& p, size_t& cap, size_t need) {
        if (need <= cap) return true;
        if (!x.sync()) return xfail();
        x.release(p);
        p = (uint8_t*)x.alloc(need + need / 4);
        cap = p ? need + need / 4 : 0;
        return p ? true : fail("out of memory (retain scratch)");
    }
    RetainMut mut() const {
        RetainMut m{};
        m.base = base;
        m.base_n = base_n;
        m.onodes = d_nodes;
        m.ov_cap = ov_cap;
        m.oedges = d_edges;
        m.oedge_mask = edge_slots - 1;
        m.opool = d_pool;
        m.opool_cap = pool_cap;
        m.expire_at = d_expire_at;
        m.ts = d_ts;
        m.expiry = d_expiry;
        m.id_node = d_id_node;
        m.id_tnode = d_id_tnode;
        m.id_cap = id_cap;
        m.dead_bits = d_dead;
        m.dead_rank = d_rank;
        m.last_op = d_last;
        m.ctr = d_ctr;
        return m;
    }
    template <class T> T* take(size_t n) { return (T*)x.alloc(sizeof(T) * (n ? n : 1)); }
    bool alloc_ids(uint32_t cap) {
        id_cap = cap;
        const size_t w = n_words() + 1;
        d_expire_at = take<unsigned long long>(cap);
        d_ts = take<unsigned long long>(cap);
        d_expiry = take<uint32_t>(cap);
        d_id_node = take<uint32_t>(cap);
        d_id_tnode = take<uint32_t>(cap);
        d_dead = take<unsigned long long>(w);
        d_rank = take<uint32_t>(w + 1);
        d_ctr = take<RetainCounters>(1);
        if (!d_expire_at || !d_ts || !d_expiry || !d_id_node || !d_id_tnode || !d_dead || !d_rank || !d_ctr) return fail("out of memory (retain ids)");
        if (!x.zero(d_expire_at, 8 * (size_t)cap) || !x.zero(d_ts, 8 * (size_t)cap) || !x.zero(d_expiry, 4 * (size_t)cap) ||
            !x.fill_bytes(d_id_node, 0xFF, 4 * (size_t)cap) || !x.fill_bytes(d_id_tnode, 0xFF, 4 * (size_t)cap))
            return xfail();
        return true;
    }
    bool alloc_last() { // bids: one per id and per overlay node
        if (d_last) {
            if (!x.sync()) return xfail();
            x.release(d_last);
        }
        d_last = take<unsigned long long>((size_t)id_cap + ov_cap);
        if (!d_last) return fail("out of memory (retain bids)");
        return x.zero(d_last, 8 * ((size_t)id_cap + ov_cap)) ? true : xfail();
    }
    bool alloc_overlay(uint32_t nodes, uint32_t pool) {
        ov_cap = nodes;
        pool_cap = pool;
        edge_slots = pow2_at_least((uint64_t)nodes * 2);
        if (edge_slots < 8) edge_slots = 8;
        d_nodes = take<ONode>(nodes);
        d_edges = take<uint32_t>(edge_slots);
        d_pool = (uint8_t*)x.alloc((size_t)pool + 16);
        if (!d_nodes || !d_edges || !d_pool) return fail("out of memory (retain overlay)");
        if (!x.zero(d_edges, 4 * (size_t)edge_slots)) return xfail();
        return alloc_last();
    }
    template <class T> bool regrow(T*& p, size_t old_n, size_t new_n, int fill) {
        T* np = take<T>(new_n);
        if (!np) return fail("out of memory (retain growth)");
        if (!x.copy(np, p, sizeof(T) * old_n) || !x.fill_bytes((uint8_t*)(np + old_n), fill, sizeof(T) * (new_n - old_n)) || !x.sync()) return xfail();
        x.release(p);
        p = np;
        return true;
    }
    bool grow_ids(uint32_t cap) {
        const uint32_t old = id_cap, old_w = n_words() + 1;
        id_cap = cap;
        const uint32_t new_w = n_words() + 1;
        if (!regrow(d_expire_at, old, cap, 0) || !regrow(d_ts, old, cap, 0) || !regrow(d_expiry, old, cap, 0) || !regrow(d_id_node, old, cap, 0xFF) ||
            !regrow(d_id_tnode, old, cap, 0xFF) || !regrow(d_dead, old_w, new_w, 0xFF))
            return false;
        if (!x.sync()) return xfail();
        x.release(d_rank);
        d_rank = take<uint32_t>((size_t)new_w + 1);
        if (!d_rank) return fail("out of memory (retain growth)");
        // (the old last word was allocated all-ones beyond id_cap: ids in it that are now in range are "not retained", as they must be)
        return alloc_last();
    }
    bool grow_overlay(uint32_t nodes, uint32_t pool) {
        if (nodes > ov_cap && !regrow(d_nodes, ov_cap, nodes, 0)) return false;
        if (pool > pool_cap) {
            uint8_t* np = (uint8_t*)x.alloc((size_t)pool + 16);
            if (!np) return fail("out of memory (retain growth)");
            if (!x.copy(np, d_pool, pool_cap) || !x.sync()) return xfail();
            x.release(d_pool);
            d_pool = np;
            pool_cap = pool;
        }
        ov_cap = std::max(ov_cap, nodes);
        const uint32_t want = pow2_at_least((uint64_t)ov_cap * 2);
        if (want > edge_slots) { // re-insert every node into the larger table
            if (!x.sync()) return xfail();
            x.release(d_edges);
            d_edges = take<uint32_t>(want);
            if (!d_edges) return fail("out of memory (retain growth)");
            edge_slots = want;
            if (!x.zero(d_edges, 4 * (size_t)edge_slots) || !x.r_rehash(mut(), hc.ov_nodes)) return xfail();
        }
        return alloc_last();
    }
};

} // namespace bmq
