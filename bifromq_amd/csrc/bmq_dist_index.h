// bmq_dist_index.h -- owner of the dist index: the arrays of bmq_layout.h in "exec memory" (HBM under DevExec, host memory
// under HostExec) and the host-side control of the builder pipeline of bmq_build_core.h.  The host side only moves bytes,
// sizes buffers and reads back counters; every key is parsed, every node inserted and every id set edited by the builder
// functions on the exec side.
//
//   rebuild(keys)  <- IKVRangeCoProc.reset(Boundary)           (KVAPI/IKVRangeCoProc.java:64, DW/DistWorkerCoProc.java:283-291)
//   apply(ops)     <- ISubscriptionCache.refresh(add/remove)   (DW/DistWorkerCoProc.java:188-209, DW/cache/SubscriptionCache.java:127-134)
//   route_key(id)  -> key bytes, so the Java side can materialise Matching objects (SCHEMA/KVSchemaUtil.java:73-89)
//
// Exec concept (bmq_exec_host.h: HostExec; bmq_engine.hip: DevExec):
//   void* alloc(size_t); void release(void*);
//   bool copy_in(dst, src, n)  [blocking]   bool copy_in_async(dst, src, n) [src valid until sync()]
//   bool copy_out(dst, src, n) [blocking, after everything queued before]   bool copy(dst, src, n)   bool zero(p, n)   bool sync()
//   bool fill_slots(TrieSlot*, n), iota(uint32_t*, n)
//   bool prepare(ix, ob), prepare_check(ix, ob, n_dir), bulk_prepare(ix, ob), scan_flags(in, out, n), bulk_tenants(ix, ob, scan),
//        locate(ix, ob), sort_targets(ob), group(ix, ob), rehash(ix, old_base, old_slots, new_base, new_buckets),
//        dict_rehash(old, old_slots, ix), find(ix, query, tenant_len, filter_len, out, cap), gather_refs(ix, ids, n, out_refs),
//        gather_bytes(ix, refs, offs, n, out)
//   std::string err;
#pragma once
#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "bmq_build_core.h"

// A tenant's region holds nodes * (1 + slack / 4) two-slot buckets (DistIndex::slack_num, bmq_config.region_slack): 1 -> load factor 0.4
// (rounds 2-5), 3 -> 0.29, 6 -> 0.2 (the default since round 6).  What the slack buys is fewer SECOND probes: at 0.4 a publish of the survey's
// workload pays 1.24 line fetches beyond the home buckets of its 10.2 probes (a home bucket full of other edges; a Bloom false positive walks on
// to the first bucket with a free slot), at 0.29 0.55, at 0.2 0.28, at 0.125 0.12 -- and k_walk, which is bound by the rate of its line requests,
// takes 0.212 / 0.198 / 0.195 / 0.194 ms (profiles/r06/extras/ab_region_slack.txt).  The price is memory: 64 bytes x (1 + slack / 4) per node.
#ifndef BMQ_REGION_SLACK_NUM
#define BMQ_REGION_SLACK_NUM 6
#endif

namespace bmq {

struct DistIndexStats {
    uint64_t n_routes = 0, n_tenants = 0, n_nodes = 0, n_tokens = 0;
    uint64_t trie_slots = 0, dict_slots = 0, bytes = 0;
    uint64_t id_list_words = 0, id_list_garbage = 0, trie_garbage_slots = 0, key_bytes = 0;
    uint32_t next_id = 0;
};
struct ApplyResult {
    uint32_t added = 0, removed = 0, dups = 0;
};

template <class Exec> class DistIndex {
    // BMQ_TIMING=1: phase times on stderr (each lap synchronises the exec first, so the numbers include the kernels)
    struct PhaseTimer {
        Exec& x;
        bool on;
        std::chrono::steady_clock::time_point t0;
        explicit PhaseTimer(Exec& ex) : x(ex), on(bmq_env("BMQ_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
        void lap(const char* what) {
            if (!on) return;
            (void)x.sync();
            const auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[bmq index] %-34s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        }
    };

public:
    explicit DistIndex(Exec& exec) : x(exec) {}
    ~DistIndex() { drop(); }
    DistIndex(const DistIndex&) = delete;
    DistIndex& operator=(const DistIndex&) = delete;

    Exec& x;
    std::string error;
    bool built = false, broken = false;
    uint64_t generation = 0; // bumped by every rebuild: ids of different generations are unrelated
    uint32_t slack_num = BMQ_REGION_SLACK_NUM; // region size = nodes * (1 + slack_num / 4) buckets (bmq_config.region_slack)
    bool tiny = false;       // test knob (tools/host_fuzz.cpp): minimal initial capacities, so that every growth path runs all the time

    // ---- arrays in exec memory ----
    TrieSlot* trie = nullptr;
    uint64_t trie_cap = 0, trie_used = 0;
    TenantSlot* dir = nullptr;
    uint32_t dir_slots = 0;
    uint8_t* names = nullptr;
    uint32_t names_cap = 0, names_used = 0;
    DictSlot* dict = nullptr;
    uint32_t dict_slots = 0;
    uint8_t* dpool = nullptr;
    uint32_t dpool_cap = 0;
    uint32_t* route_pos = nullptr;
    uint64_t rp_cap = 0;
    unsigned long long* kref = nullptr;
    uint32_t* khash = nullptr;
    uint32_t id_cap = 0, next_id = 0;
    uint64_t n_routes = 0; // live routes (host count: rebuild sets it, every apply adds what the groups report)
    uint8_t* kpool = nullptr;
    uint64_t kpool_cap = 0, kpool_used = 0;
    BuildCounters* bc = nullptr;
    BuildCounters hbc{}; // last read-back
    uint32_t* fo_dgroup = nullptr; // registered by the fan-out grouping (bmq_fanout.h): deletes mark their ids dead in its per-id cache
    uint32_t fo_cap = 0;

    // ---- host bookkeeping ----
    std::unordered_map<std::string, uint32_t> tenant_slot; // tenant id -> directory slot
    std::vector<std::pair<uint64_t, uint64_t>> free_regions; // (base, slots) abandoned by region growth
    uint64_t trie_garbage = 0;

    DistIndexView view() const {
        DistIndexView v{};
        v.trie = trie;
        v.tenants = dir;
        v.tenant_mask = dir_slots - 1;
        v.tenant_names = names;
        v.dict = dict;
        v.dict_group_mask = dict_slots / DICT_GROUP - 1;
        v.pool = dpool;
        v.route_pos = route_pos;
        return v;
    }
    DistIndexMut mut() const {
        DistIndexMut m{};
        m.trie = trie;
        m.tenants = dir;
        m.tenant_mask = dir_slots - 1;
        m.tenant_names = names;
        m.dict = dict;
        m.dict_group_mask = dict_slots / DICT_GROUP - 1;
        m.dpool = dpool;
        m.dpool_cap = dpool_cap;
        m.route_pos = route_pos;
        m.rp_cap = rp_cap;
        m.kref = kref;
        m.khash = khash;
        m.id_cap = id_cap;
        m.kpool = kpool;
        m.bc = bc;
        m.fo_dgroup = fo_dgroup;
        m.fo_cap = fo_cap;
        return m;
    }

    // ---- full (re)load.  keys: packed route keys, any order (a KV scan yields them sorted: that is the fast path) ----
    bool rebuild(const uint8_t* keys, const uint32_t* key_off, uint32_t n) {
        error.clear();
        std::vector<uint8_t> sorted_bytes; // only used when the input was not strictly ascending
        std::vector<uint32_t> sorted_off;
        for (int attempt = 0; attempt < 2; attempt++) {
            uint32_t err = 0;
            if (!rebuild_sorted(keys, key_off, n, err)) {
                if (err & ERR_BAD_KEY) return false;
                if ((err & ERR_UNSORTED) && attempt == 0) { // not a KV scan: order + de-duplicate on the host, once
                    sort_unique(keys, key_off, n, sorted_bytes, sorted_off);
                    keys = sorted_bytes.data();
                    key_off = sorted_off.data();
                    n = (uint32_t)sorted_off.size() - 1;
                    continue;
                }
                return false;
            }
            built = true;
            broken = false;
            generation++;
            return true;
        }
        return fail("rebuild: input could not be ordered");
    }

    // ---- post-commit mutations, applied in order.  op[i]: 0 = put, 1 = delete ----
    // apply = apply_begin + apply_end.  On the device apply_begin only ENQUEUES: the ops are uploaded on the executor's upload stream (beside
    // whatever the engine stream still runs), the stages follow on the engine stream behind the gate, the counters travel back into
    // page-locked memory; apply_end waits for them, lets the stage-by-stage loops finish what the gate stopped and does the bookkeeping.
    // The caller's buffers must stay valid in between; at most one batch is open.
    struct OpenApply {
        bool open = false;
        const uint8_t* keys = nullptr;
        const uint32_t* key_off = nullptr;
        const uint8_t* op = nullptr;
        uint32_t n = 0, n_put = 0;
        uint64_t kb = 0;
        OpBatch ob{};
        std::vector<uint32_t> put_rank;
    } open_apply;
    bool apply(const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n, ApplyResult* res = nullptr) {
        return apply_begin(keys, key_off, op, n) && apply_end(res);
    }
    // sized: a batch of the carry-over behind reserve_like(old) (import_apply) -- all puts of keys `old` holds, `keys` is device memory
    bool apply_begin(const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n, bool sized = false) {
        error.clear();
        OpenApply& oa = open_apply;
        if (oa.open) return fail("an apply batch is open: apply_end first");
        if (n == 0) return true;
        if (broken) return fail("the index is inconsistent after a failed batch: bmq_rebuild is required");
        if (!built && !reset_empty()) return false;
        const uint64_t kb = key_off[n];
        oa.put_rank.resize(n);
        uint32_t n_put = 0;
        for (uint32_t i = 0; i < n; i++) {
            oa.put_rank[i] = n_put;
            n_put += op[i] == 0;
        }
        if ((uint64_t)next_id + n_put >= 0xFFFFFFF0ull) return fail("route id space exhausted: bmq_rebuild re-numbers the routes");
        if (!ensure_keys(kb) || !ensure_ids(next_id + n_put) || !ensure_scratch(n, true)) return false;
        OpBatch ob = batch(n);
        ob.key_base = kpool_used;
        ob.id_base = next_id;
        ob.put_rank = s_put_rank;
        ob.op = s_op;
        ob.bulk = 0;
        ob.sized = sized ? 1 : 0;
        if (!x.upload_async(kpool + kpool_used, keys, kb) || !x.upload_async(s_key_off, key_off, sizeof(uint32_t) * ((size_t)n + 1)) ||
            !x.upload_async(s_op, op, n) || !x.upload_async(s_put_rank, oa.put_rank.data(), sizeof(uint32_t) * (size_t)n) || !x.uploads_done())
            return xfail();
        oa.keys = keys, oa.key_off = key_off, oa.op = op, oa.n = n, oa.n_put = n_put, oa.kb = kb, oa.ob = ob;
        if (Exec::gated) { // the stages back to back behind the gate, the counters on their way back: nothing waits here
            DistIndexMut ix = mut();
            // (the gate word and the per-batch counters are one memset; nothing is zeroed BETWEEN the stages any more: a stage that leaves a
            // counter non-zero closes the gate, and behind an open gate the counters are still zero -- round 5 spent three fill kernels on it)
            static_assert(offsetof(BuildCounters, err) == offsetof(BuildCounters, gate) + 8, "gate, its pad and the per-batch block are contiguous");
            if (!x.zero(&bc->gate, sizeof(BuildCounters) - offsetof(BuildCounters, gate)) || !x.prepare(ix, ob) || !x.prepare_check(ix, ob, dir_slots) || !x.gate(bc, 1) ||
                !x.locate(ix, ob) || !x.prepare_check(ix, ob, dir_slots) || !x.gate(bc, 2) || !x.zero(ob.group_done, n) ||
                !x.sort_targets(ob) || !x.group(ix, ob) || !x.gate(bc, 3) || !x.read_back_async(&hbc, bc, sizeof(BuildCounters)))
                return xfail();
        }
        oa.open = true;
        return true;
    }
    bool apply_end(ApplyResult* res = nullptr) {
        OpenApply& oa = open_apply;
        if (!oa.open) return true;
        oa.open = false;
        const uint32_t* const key_off = oa.key_off;
        const uint32_t n = oa.n, n_put = oa.n_put;
        const uint64_t kb = oa.kb;
        OpBatch ob = oa.ob;
        PhaseTimer pt(x);
        auto count_groups = [&]() { // what the group step of this round did (the counters are zeroed in front of every round)
            uint64_t a = 0, r = 0, du = 0;
            for (uint32_t c = 0; c < N_CTR_LANES; c++) a += hbc.n_added[c], r += hbc.n_removed[c], du += hbc.n_dups[c];
            n_routes += a;
            n_routes -= r;
            if (res) {
                res->added += (uint32_t)a;
                res->removed += (uint32_t)r;
                res->dups += (uint32_t)du;
            }
        };
        // ---- the stages back to back, ONE read-back (round 5; rounds 1-4: a read-back behind each of the three stages).  A stage that
        // leaves work for the host closes the gate (BuildCounters.gate): the stages behind it do nothing and the loops below -- the
        // stage-by-stage form, idempotent -- take over from that stage.  The common batch needs nothing from the host.
        uint32_t from = 1; // the stage the stage-by-stage form starts at; 4: nothing left to do
        bool groups_pending = false;
        if (Exec::gated) {
            if (!x.read_back_wait(&hbc, sizeof(BuildCounters))) return xfail();
            pt.lap("apply: prepare, locate, sort, groups (one read-back)");
            from = hbc.gate ? hbc.gate : 4;
            if (from >= 3) { // the group step ran (to the end, or up to the groups that found the id-list pool full)
                if (hbc.err) return broke("apply: group step failed");
                count_groups();
                groups_pending = hbc.n_deferred != 0;
                if (groups_pending && !grow_route_pos(hbc.rp_used + 2 * hbc.rp_need + 1024)) return false;
            }
            if (hbc.gate && !x.zero(&bc->gate, sizeof(uint32_t))) return xfail();
        }
        // ---- prepare: validate + tenants + growth bounds (re-run once if tenants had to be created) ----
        for (int round = 0; from <= 1; round++) {
            if (!zero_batch_counters()) return false;
            DistIndexMut ix = mut();
            if (!x.prepare(ix, ob) || !x.prepare_check(ix, ob, dir_slots) || !read_counters()) return xfail();
            if (hbc.err & ERR_BAD_KEY) return fail("malformed route key or op in apply batch"); // nothing was changed
            if (hbc.n_unknown == 0) break;
            if (round == 1) return fail("apply: tenant creation did not converge");
            std::vector<uint32_t> unk(hbc.n_unknown);
            if (!x.copy_out(unk.data(), ob.unknown_list, sizeof(uint32_t) * unk.size())) return xfail();
            std::unordered_map<std::string, uint64_t> need; // new tenant -> nodes its puts may add
            // (from this index's key pool, where apply_begin put the batch: the caller's pointer may be device memory -- the builder of the
            // next generation hands the keys over where they live.  A few keys one by one, many with one copy of the batch.)
            std::vector<uint8_t> kbuf;
            const bool whole = unk.size() > 8;
            if (whole) {
                kbuf.resize(kb);
                if (!x.copy_out(kbuf.data(), kpool + ob.key_base, kb)) return xfail();
            }
            for (uint32_t i : unk) {
                const size_t klen = key_off[i + 1] - key_off[i];
                if (!whole) {
                    kbuf.resize(klen);
                    if (!x.copy_out(kbuf.data(), kpool + ob.key_base + key_off[i], klen)) return xfail();
                }
                const std::string_view k((const char*)kbuf.data() + (whole ? key_off[i] : 0u), klen);
                const size_t tlen = ((size_t)(uint8_t)k[1] << 8) | (uint8_t)k[2]; // validated by prepare
                uint64_t levels = 1;
                for (size_t q = 3 + tlen; q < k.size(); q++) levels += k[q] == 0;
                need[std::string(k.substr(3, tlen))] += levels;
            }
            for (auto& e : need)
                if (!create_tenant(e.first, e.second, 0)) return false;
            if (!flush_directory()) return false;
            ob.grow_list = s_grow; // the directory may have grown, and its scratch with it
        }
        if (from <= 1) {
            pt.lap("apply: prepare (+ tenants)");
            if (!grow_flagged_regions(ob)) return false;
            pt.lap("apply: region growth");
        }
        // ---- locate (idempotent: re-run after growing what it ran out of) ----
        for (int attempt = 0; from <= 2; attempt++) {
            if (attempt == 8) return broke("apply: growth did not converge");
            if (!zero_batch_counters()) return false;
            DistIndexMut ix = mut();
            if (!x.locate(ix, ob) || !x.prepare_check(ix, ob, dir_slots) || !read_counters()) return xfail();
            if (hbc.err & ERR_STUCK) return broke("apply: a probe loop did not terminate");
            if (!(hbc.err & (ERR_DICT_FULL | ERR_REGION_FULL)) && hbc.n_grow == 0) break;
            if ((hbc.err & ERR_DICT_FULL) && !grow_dict(std::max<uint64_t>((uint64_t)dict_slots * 4, 4096), (uint64_t)dpool_cap * 2 + kb)) return false;
            if (!grow_flagged_regions(ob)) return false;
        }
        // ---- sort by filter node, apply the groups ----
        if (from <= 2) {
            pt.lap("apply: locate");
            if (!x.zero(ob.group_done, n) || !x.sort_targets(ob)) return xfail();
            pt.lap("apply: sort by filter node");
        }
        for (int attempt = 0; from <= 2 || groups_pending; attempt++) {
            if (attempt == 8) return broke("apply: id-list pool growth did not converge");
            if (!zero_batch_counters()) return false;
            DistIndexMut ix = mut();
            if (!x.group(ix, ob) || !read_counters()) return xfail();
            if (hbc.err) return broke("apply: group step failed");
            count_groups();
            if (hbc.n_deferred == 0) break;
            if (!grow_route_pos(hbc.rp_used + 2 * hbc.rp_need + 1024)) return false;
        }
        pt.lap("apply: groups");
        kpool_used += (kb + 15) & ~15ull;
        next_id += n_put;
        if ((uint64_t)hbc.n_tokens * 4 > dict_slots && !grow_dict((uint64_t)dict_slots * 4, dpool_cap)) return false;
        return true;
    }

    // ---- maintenance: re-build the index from its own live routes.  Drops abandoned regions and id lists, dead trie nodes and
    // dictionary tokens nobody uses any more; the key store shrinks to the live keys.  Route ids are re-numbered (ranks again):
    // a new generation, exactly like a bmq_rebuild with the current key set.
    bool compact() {
        error.clear();
        if (!built) return true;
        std::vector<uint32_t> ids(next_id);
        for (uint32_t i = 0; i < next_id; i++) ids[i] = i;
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> off;
        if (!route_keys(ids.data(), next_id, bytes, off)) return false;
        if (off[next_id] >= 0xFFFFFFF0ull) return fail("compact: more than 4 GB of route keys");
        std::vector<uint8_t> kb;
        std::vector<uint32_t> ko{0};
        kb.reserve(bytes.size() + 16);
        for (uint32_t i = 0; i < next_id; i++)
            if (off[i + 1] > off[i]) { // live
                kb.insert(kb.end(), bytes.begin() + (long)off[i], bytes.begin() + (long)off[i + 1]);
                ko.push_back((uint32_t)kb.size());
            }
        kb.resize(kb.size() + 16, 0);
        return rebuild(kb.data(), ko.data(), (uint32_t)ko.size() - 1); // not ascending any more after churn: sorted on the host
    }

    // ---- id -> key ----
    // false + empty error: no such route (never existed or deleted)
    bool route_key(uint32_t id, std::string& out) {
        out.clear();
        if (!built || id >= next_id) return false;
        unsigned long long r = 0;
        if (!x.copy_out(&r, kref + id, sizeof(r))) return xfail();
        if (r == 0) return false;
        out.resize((size_t)(r >> KREF_LEN_SHIFT));
        if (!x.copy_out(out.data(), kpool + (r & KREF_OFF_MASK), out.size())) return xfail();
        return true;
    }
    // keys of many ids at once: out_off[n + 1], bytes appended to out (a dead id gives an empty key)
    bool route_keys(const uint32_t* ids, uint32_t n, std::vector<uint8_t>& out, std::vector<uint64_t>& out_off) {
        out.clear();
        out_off.assign((size_t)n + 1, 0);
        if (n == 0) return true;
        if (!built) return fail("no index");
        if (!ensure_buf(g_ids, g_ids_cap, n) || !ensure_buf(g_refs, g_refs_cap, n) || !ensure_buf(g_offs, g_offs_cap, (size_t)n + 1)) return false;
        std::vector<unsigned long long> refs(n);
        if (!x.copy_in(g_ids, ids, sizeof(uint32_t) * (size_t)n) || !x.gather_refs(mut(), g_ids, n, next_id, g_refs) ||
            !x.copy_out(refs.data(), g_refs, sizeof(unsigned long long) * (size_t)n))
            return xfail();
        for (uint32_t i = 0; i < n; i++) out_off[i + 1] = out_off[i] + (refs[i] >> KREF_LEN_SHIFT);
        const uint64_t total = out_off[n];
        out.resize(total);
        if (total == 0) return true;
        if (!ensure_buf(g_bytes, g_bytes_cap, total)) return false;
        if (!x.copy_in(g_offs, out_off.data(), sizeof(uint64_t) * ((size_t)n + 1)) || !x.gather_bytes(mut(), g_refs, g_offs, n, g_bytes) ||
            !x.copy_out(out.data(), g_bytes, total))
            return xfail();
        return true;
    }
    // ---- generation change: THIS index (the generation being built, behind reserve_like(old)) takes the live keys among old's ids [lo, hi).
    // Two steps, so that the caller needs to keep `old` still only for the first.  import_snapshot ENQUEUES a copy of old's key references
    // kref[lo, hi) on this index's executor (the caller ordered it behind whatever old was told so far) and notes where old's key bytes
    // lie -- the key pool is append-only, and while a generation change runs old keeps the blocks it outgrows (defer_release), so the
    // pointer stays good whatever old is told meanwhile.  import_apply does the rest on this index's executor alone: the references come
    // back (8 bytes per id: length + liveness), the live keys are gathered -- HBM to HBM, the bytes never visit the host -- and go through
    // the apply path as puts (no order required, so no sort) into regions that are already sized.
    struct Import {
        uint32_t n = 0;
        const uint8_t* old_kpool = nullptr;
    } imp;
    bool import_snapshot(DistIndex& old, uint32_t lo, uint32_t hi) {
        error.clear();
        imp = Import{};
        if (!old.built) return true;
        if (hi > old.next_id) hi = old.next_id;
        if (hi <= lo) return true;
        const uint32_t n = hi - lo;
        if (!ensure_buf(g_refs, g_refs_cap, n)) return false;
        if (!x.copy(g_refs, old.kref + lo, sizeof(unsigned long long) * (size_t)n)) return xfail();
        imp.n = n;
        imp.old_kpool = old.kpool;
        return true;
    }
    bool import_apply(uint32_t& n_live) {
        n_live = 0;
        const uint32_t n = imp.n;
        if (n == 0) return true;
        imp.n = 0;
        std::vector<unsigned long long> refs(n);
        if (!x.copy_out(refs.data(), g_refs, sizeof(unsigned long long) * (size_t)n)) return xfail();
        std::vector<uint64_t> offs((size_t)n + 1, 0);
        std::vector<uint32_t> live_off(1, 0u);
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t len = refs[i] >> KREF_LEN_SHIFT; // 0: no such route (never existed / deleted)
            offs[i + 1] = offs[i] + len;
            if (len) live_off.push_back((uint32_t)offs[i + 1]);
        }
        const uint64_t total = offs[n];
        if (total == 0) return true;
        if (total >= 0xFFFFFFF0ull) return fail("generation change: more than 4 GB of route keys in one chunk");
        if (!ensure_buf(g_offs, g_offs_cap, (size_t)n + 1) || !ensure_buf(g_bytes, g_bytes_cap, total + 16)) return false;
        DistIndexMut m = mut();
        m.kpool = const_cast<uint8_t*>(imp.old_kpool); // (gather_bytes reads the key pool and nothing else)
        if (!x.copy_in_async(g_offs, offs.data(), sizeof(uint64_t) * ((size_t)n + 1)) || !x.gather_bytes(m, g_refs, g_offs, n, g_bytes) || !x.sync()) return xfail();
        n_live = (uint32_t)live_off.size() - 1;
        const std::vector<uint8_t> puts(n_live, 0);
        return apply_begin(g_bytes, live_off.data(), puts.data(), n_live, true) && apply_end(nullptr);
    }
    // The buffers a chunk of `n` ids / `bytes` key bytes of a generation change goes through, taken NOW: the first bmq_compact_poll used to
    // allocate them -- a dozen device allocations beside a saturated matcher, the one batch of a compaction that took 10 ms instead of 0.35.
    bool reserve_import(uint32_t n, uint64_t bytes) {
        return ensure_buf(g_refs, g_refs_cap, n) && ensure_buf(g_offs, g_offs_cap, (size_t)n + 1) && ensure_buf(g_bytes, g_bytes_cap, bytes + 16) &&
               ensure_scratch(n, true);
    }
    // blocks this index outgrew while another generation was being built from it
    bool defer_release = false;
    std::vector<void*> graveyard;
    void release_deferred() {
        for (void* q : graveyard) x.release(q);
        graveyard.clear();
    }
    uint32_t id_bound() const { return next_id; } // ids handed out so far (live or not)
    // Room for everything a generation change is about to hand over, taken once: every tenant of `old` that has routes gets its region at
    // the size its trie had there (dead nodes included: an upper bound), the trie pool, the dictionary, the id-list pool, the key store
    // and the id tables likewise.  The batches that carry the keys over then meet no unknown tenant and grow nothing.
    bool reserve_like(DistIndex& old) {
        error.clear();
        if (!built && !reset_empty()) return false;
        if (!old.built) return true;
        if (!old.read_counters()) return fail(old.x.err);
        std::vector<TenantSlot> d(old.dir_slots);
        if (old.dir_slots && !old.x.copy_out(d.data(), old.dir, sizeof(TenantSlot) * (size_t)old.dir_slots)) return fail(old.x.err);
        uint64_t slots = 0, n_ten = 0;
        for (auto& t : d)
            if ((t.hash_lo | t.hash_hi) && t.n_routes) slots += 2ull * buckets_for((uint64_t)t.n_nodes + 1), n_ten++;
        if (trie_used + slots > trie_cap) {
            if (trie_used + slots >= 0xFFFFFFF0ull) return fail("trie too large (2^32 slots)");
            const uint64_t cap = std::min<uint64_t>(trie_used + slots + (tiny ? 0 : slots / 8 + (1u << 16)), 0xFFFFFFF0ull);
            if (!regrow(trie, trie_used, cap)) return false;
            if (!x.fill_slots(trie + trie_used, cap - trie_used)) return xfail();
            trie_cap = cap;
        }
        if (!ensure_directory(tenant_slot.size() + n_ten)) return false;
        for (auto& t : d)
            if ((t.hash_lo | t.hash_hi) && t.n_routes &&
                !create_tenant(std::string((const char*)old.names_h.data() + t.name_off, t.name_len), (uint64_t)t.n_nodes + 1, 0))
                return false;
        if (!flush_directory()) return false;
        if (old.dict_slots > dict_slots && !grow_dict(old.dict_slots, old.dpool_cap)) return false;
        if (!grow_route_pos(old.hbc.rp_used + 1024)) return false; // (lists are handed out in blocks: the old pool's size, not the live words)
        return ensure_keys(old.kpool_used) && ensure_ids(std::min<uint64_t>((uint64_t)next_id + old.next_id + 64, 0xFFFFFFF0ull));
    }
    // exact lookup (inspection only -- never used for matching): ids stored under (tenant, MQTT filter)
    bool find(std::string_view tenant, std::string_view filter, std::vector<uint32_t>& ids) {
        ids.clear();
        if (!built) return true;
        const size_t qn = tenant.size() + filter.size();
        std::vector<uint8_t> q(qn + 32, 0);
        if (!tenant.empty()) memcpy(q.data(), tenant.data(), tenant.size());
        if (!filter.empty()) memcpy(q.data() + tenant.size(), filter.data(), filter.size());
        if (!ensure_buf(g_bytes, g_bytes_cap, q.size())) return false;
        uint32_t cap = 1024;
        for (;;) {
            if (!ensure_buf(g_ids, g_ids_cap, (size_t)cap + 1)) return false;
            if (!x.copy_in(g_bytes, q.data(), q.size()) ||
                !x.find(mut(), g_bytes, (uint32_t)tenant.size(), (uint32_t)filter.size(), g_ids, cap))
                return xfail();
            uint32_t cnt = 0;
            if (!x.copy_out(&cnt, g_ids, sizeof(cnt))) return xfail();
            if (cnt > cap) {
                cap = cnt;
                continue;
            }
            ids.resize(cnt);
            if (cnt && !x.copy_out(ids.data(), g_ids + 1, sizeof(uint32_t) * (size_t)cnt)) return xfail();
            return true;
        }
    }
    bool stats(DistIndexStats& st) {
        st = DistIndexStats{};
        if (!built) return true;
        if (!read_counters()) return xfail();
        std::vector<TenantSlot> d(dir_slots);
        if (!x.copy_out(d.data(), dir, sizeof(TenantSlot) * (size_t)dir_slots)) return xfail();
        for (auto& t : d)
            if (t.hash_lo | t.hash_hi) {
                st.n_tenants += t.n_routes != 0 ? 1 : 0;
                st.n_nodes += t.n_nodes;
            }
        st.n_routes = n_routes;
        st.n_tokens = hbc.n_tokens;
        st.trie_slots = trie_used;
        st.dict_slots = dict_slots;
        st.id_list_words = hbc.rp_used;
        st.id_list_garbage = hbc.rp_garbage;
        st.trie_garbage_slots = trie_garbage;
        st.key_bytes = kpool_used;
        st.next_id = next_id;
        st.bytes = trie_cap * sizeof(TrieSlot) + (uint64_t)dir_slots * sizeof(TenantSlot) + names_cap + (uint64_t)dict_slots * sizeof(DictSlot) +
                   dpool_cap + rp_cap * 4 + (uint64_t)id_cap * 12 + kpool_cap;
        return true;
    }

    void drop() {
        auto rel = [&](auto*& p) {
            if (p) x.release((void*)p);
            p = nullptr;
        };
        release_deferred();
        rel(trie); rel(dir); rel(names); rel(dict); rel(dpool); rel(route_pos); rel(kref); rel(khash); rel(kpool); rel(bc);
        rel(s_key_off); rel(s_op); rel(s_put_rank); rel(s_dir_slot); rel(s_nn); rel(s_flag); rel(s_target); rel(s_order);
        rel(s_sorted_target); rel(s_group_done); rel(s_unknown); rel(s_grow); rel(s_bt_first); rel(s_bt_nodes); rel(s_bt_keys); rel(s_bt_dir);
        rel(g_ids); rel(g_refs); rel(g_offs); rel(g_bytes);
        trie_cap = trie_used = 0; dir_slots = 0; names_cap = names_used = 0; dict_slots = 0; dpool_cap = 0; rp_cap = 0; id_cap = next_id = 0;
        kpool_cap = kpool_used = 0; s_cap = 0; s_bt_cap = 0; s_grow_cap = 0; g_ids_cap = g_refs_cap = g_offs_cap = g_bytes_cap = 0;
        tenant_slot.clear(); free_regions.clear(); dir_h.clear(); names_h.clear(); trie_garbage = 0;
        built = false;
    }

private:
    // ---- scratch in exec memory ----
    uint32_t *s_key_off = nullptr, *s_put_rank = nullptr, *s_dir_slot = nullptr, *s_nn = nullptr, *s_flag = nullptr, *s_order = nullptr,
             *s_unknown = nullptr, *s_grow = nullptr, *s_bt_first = nullptr, *s_bt_nodes = nullptr, *s_bt_keys = nullptr, *s_bt_dir = nullptr;
    uint8_t *s_op = nullptr, *s_group_done = nullptr;
    unsigned long long *s_target = nullptr, *s_sorted_target = nullptr;
    size_t s_cap = 0, s_bt_cap = 0, s_grow_cap = 0;
    uint32_t* g_ids = nullptr;
    unsigned long long* g_refs = nullptr;
    uint64_t* g_offs = nullptr;
    uint8_t* g_bytes = nullptr;
    size_t g_ids_cap = 0, g_refs_cap = 0, g_offs_cap = 0, g_bytes_cap = 0;
    // host mirror of the directory (placement only; live counters stay on the exec side) and of the tenant names
    std::vector<TenantSlot> dir_h;
    std::vector<uint8_t> names_h;
    bool dir_dirty = false;

    bool fail(const std::string& m) {
        error = m;
        return false;
    }
    bool xfail() { return fail(x.err.empty() ? "exec failure" : x.err); }
    bool broke(const std::string& m) {
        broken = true;
        return fail(m);
    }
    template <class T> bool ensure_buf(T*& p, size_t& cap, size_t need) { // grow-only, contents dropped
        if (need <= cap) return true;
        if (p) x.release(p);
        const size_t want = need + need / 4 + 64;
        p = (T*)x.alloc(want * sizeof(T));
        cap = p ? want : 0;
        return p ? true : fail("out of memory");
    }
    template <class T> bool regrow(T*& p, uint64_t old_n, uint64_t new_n) { // contents [0, old_n) survive
        T* q = (T*)x.alloc(new_n * sizeof(T) + 16);
        if (!q) return fail("out of memory");
        if (p && old_n && !x.copy(q, p, old_n * sizeof(T))) return xfail();
        if (p && defer_release) graveyard.push_back((void*)p); // (somebody may still read it: see import_snapshot)
        else if (p) {
            if (!x.sync()) return xfail();
            x.release(p);
        }
        p = q;
        return true;
    }
    bool read_counters() { return x.copy_out(&hbc, bc, sizeof(BuildCounters)); }
    bool zero_batch_counters() {
        const size_t off = offsetof(BuildCounters, err);
        return x.zero((uint8_t*)bc + off, sizeof(BuildCounters) - off) ? true : xfail();
    }
    OpBatch batch(uint32_t n) {
        OpBatch ob{};
        ob.key_off = s_key_off;
        ob.n = n;
        ob.dir_slot = s_dir_slot;
        ob.nn = s_nn;
        ob.flag = s_flag;
        ob.target = s_target;
        ob.order = s_order;
        ob.sorted_target = s_sorted_target;
        ob.group_done = s_group_done;
        ob.unknown_list = s_unknown;
        ob.grow_list = s_grow;
        ob.bt_first = s_bt_first;
        ob.bt_nodes = s_bt_nodes;
        ob.bt_keys = s_bt_keys;
        ob.bt_dir = s_bt_dir;
        return ob;
    }
    bool ensure_scratch(uint32_t n, bool incremental) {
        const size_t need = (size_t)n + 1;
        if (need > s_cap) {
            auto rel = [&](auto*& p) {
                if (p) x.release((void*)p);
                p = nullptr;
            };
            rel(s_key_off); rel(s_op); rel(s_put_rank); rel(s_dir_slot); rel(s_nn); rel(s_flag); rel(s_target); rel(s_order);
            rel(s_sorted_target); rel(s_group_done); rel(s_unknown);
            const size_t want = need + need / 4 + 64;
            s_key_off = (uint32_t*)x.alloc(4 * want);
            s_op = (uint8_t*)x.alloc(want);
            s_put_rank = (uint32_t*)x.alloc(4 * want);
            s_dir_slot = (uint32_t*)x.alloc(4 * want);
            s_nn = (uint32_t*)x.alloc(4 * want);
            s_flag = (uint32_t*)x.alloc(4 * want);
            s_target = (unsigned long long*)x.alloc(8 * want);
            s_order = (uint32_t*)x.alloc(4 * want);
            s_sorted_target = (unsigned long long*)x.alloc(8 * want);
            s_group_done = (uint8_t*)x.alloc(want);
            s_unknown = (uint32_t*)x.alloc(4 * want);
            if (!s_key_off || !s_op || !s_put_rank || !s_dir_slot || !s_nn || !s_flag || !s_target || !s_order || !s_sorted_target ||
                !s_group_done || !s_unknown) {
                s_cap = 0;
                return fail("out of memory (builder scratch)");
            }
            s_cap = want;
        }
        (void)incremental;
        const size_t gneed = 2 * (size_t)std::max<uint32_t>(dir_slots, 64);
        if (gneed > s_grow_cap) {
            if (s_grow) x.release(s_grow);
            s_grow = (uint32_t*)x.alloc(4 * gneed);
            s_grow_cap = s_grow ? gneed : 0;
            if (!s_grow) return fail("out of memory (builder scratch)");
        }
        return true;
    }
    void release_scratch() { // after a bulk load: 10M-key scratch is not worth keeping
        auto rel = [&](auto*& p) {
            if (p) x.release((void*)p);
            p = nullptr;
        };
        rel(s_key_off); rel(s_op); rel(s_put_rank); rel(s_dir_slot); rel(s_nn); rel(s_flag); rel(s_target); rel(s_order);
        rel(s_sorted_target); rel(s_group_done); rel(s_unknown); rel(s_bt_first); rel(s_bt_nodes); rel(s_bt_keys); rel(s_bt_dir);
        s_cap = 0;
        s_bt_cap = 0;
    }
    bool ensure_keys(uint64_t more) {
        const uint64_t need = kpool_used + more + 32;
        if (need <= kpool_cap) return true;
        if (need >= (1ull << KREF_LEN_SHIFT)) return fail("key store exceeds 1 TB");
        const uint64_t cap = tiny ? need : std::max<uint64_t>(need + need / 2, 1u << 20);
        if (!regrow(kpool, kpool_used, cap)) return false;
        kpool_cap = cap;
        return true;
    }
    bool ensure_ids(uint64_t need) {
        if (need <= id_cap) return true;
        const uint64_t cap = tiny ? need : std::min<uint64_t>(std::max<uint64_t>(need + need / 2, 1u << 16), 0xFFFFFFF0ull);
        if (!regrow(kref, id_cap, cap) || !regrow(khash, id_cap, cap)) return false;
        if (!x.zero(kref + id_cap, (cap - id_cap) * sizeof(unsigned long long))) return xfail();
        id_cap = (uint32_t)cap;
        return true;
    }
    bool grow_route_pos(uint64_t need) {
        if (need <= rp_cap) return true;
        if (need >= 0x7FFFFFF0ull) return fail("id-list pool exceeds 2^31 words: bmq_rebuild compacts it");
        const uint64_t cap = tiny ? need : std::min<uint64_t>(std::max<uint64_t>(need + need / 2, 1u << 16), 0x7FFFFFF0ull);
        if (!regrow(route_pos, std::min<uint64_t>(rp_cap, hbc.rp_used ? hbc.rp_used : rp_cap), cap)) return false;
        rp_cap = cap;
        return true;
    }
    // dictionary of `slots` slots (power of two) + string pool of `pool` bytes; existing entries are re-inserted
    bool grow_dict(uint64_t slots, uint64_t pool) {
        if (slots > (1ull << 30)) return fail("level dictionary exceeds 2^30 slots");
        uint32_t ns = tiny ? 4 : 64;
        while (ns < slots) ns <<= 1;
        DictSlot* nd = (DictSlot*)x.alloc((size_t)ns * sizeof(DictSlot));
        if (!nd) return fail("out of memory (dictionary)");
        if (!x.zero(nd, (size_t)ns * sizeof(DictSlot))) return xfail();
        const uint32_t np = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(pool, dpool_cap), 0xFFFFFF00ull);
        if (np > dpool_cap) {
            if (!read_counters()) return xfail();
            if (!regrow(dpool, std::min<uint32_t>(hbc.dpool_used, dpool_cap), np)) return false;
            dpool_cap = np;
        }
        DictSlot* old = dict;
        const uint32_t old_slots = dict_slots;
        dict = nd;
        dict_slots = ns;
        if (old) {
            if (!x.dict_rehash(old, old_slots, mut()) || !x.sync()) return xfail();
            x.release(old);
        }
        return true;
    }
    // ---- tenants ----
    static uint64_t hash_name(std::string_view s) {
        uint64_t h = TENANT_HASH_INIT;
        for (unsigned char c : s) h = tenant_hash_step(h, c);
        return tenant_hash_final(h);
    }
    bool ensure_directory(size_t n_tenants) { // load factor <= 1/2; growing re-places every entry (live counters come back first)
        if (dir_slots && n_tenants * 2 <= dir_slots) return true;
        if (dir_dirty && !flush_directory()) return false; // entries created but not uploaded yet must survive the re-placement
        uint32_t ns = tiny ? 2 : 64;
        while ((size_t)ns < n_tenants * 2 + 2) ns <<= 1;
        std::vector<TenantSlot> old;
        if (dir) {
            old.resize(dir_slots);
            if (!x.copy_out(old.data(), dir, sizeof(TenantSlot) * (size_t)dir_slots)) return xfail();
            x.release(dir);
            dir = nullptr;
        }
        dir_h.assign(ns, TenantSlot{});
        for (auto& t : old)
            if (t.hash_lo | t.hash_hi) place(t, ns);
        dir = (TenantSlot*)x.alloc(sizeof(TenantSlot) * (size_t)ns);
        if (!dir) return fail("out of memory (tenant directory)");
        dir_slots = ns;
        tenant_slot.clear();
        for (uint32_t d = 0; d < ns; d++)
            if (dir_h[d].hash_lo | dir_h[d].hash_hi)
                tenant_slot[std::string((const char*)names_h.data() + dir_h[d].name_off, dir_h[d].name_len)] = d;
        if (!x.copy_in(dir, dir_h.data(), sizeof(TenantSlot) * (size_t)ns)) return xfail();
        dir_dirty = false;
        if (s_grow_cap < 2 * (size_t)ns) {
            if (s_grow) x.release(s_grow);
            s_grow = (uint32_t*)x.alloc(8 * (size_t)ns);
            s_grow_cap = s_grow ? 2 * (size_t)ns : 0;
            if (!s_grow) return fail("out of memory (builder scratch)");
        }
        return true;
    }
    uint32_t place(const TenantSlot& t, uint32_t ns) {
        uint32_t d = (t.hash_lo ^ t.hash_hi) & (ns - 1);
        while (dir_h[d].hash_lo | dir_h[d].hash_hi) d = (d + 1) & (ns - 1);
        dir_h[d] = t;
        return d;
    }
    // a region of `slots` free (FREE_SLOT-filled) slots; NONE64 on failure
    bool alloc_region(uint64_t slots, uint64_t& base) {
        for (size_t i = 0; i < free_regions.size(); i++)
            if (free_regions[i].second >= slots && free_regions[i].second <= slots + slots / 2) { // reuse a region of similar size
                base = free_regions[i].first;
                const uint64_t got = free_regions[i].second;
                free_regions.erase(free_regions.begin() + (long)i);
                trie_garbage -= got;
                if (!x.fill_slots(trie + base, got)) return xfail();
                return true;
            }
        if (trie_used + slots > trie_cap) {
            if (trie_used + slots >= 0xFFFFFFF0ull) return fail("trie too large (2^32 slots): bmq_rebuild compacts it");
            const uint64_t cap = tiny ? trie_used + slots : std::min<uint64_t>(std::max<uint64_t>((trie_used + slots) * 2, 1u << 16), 0xFFFFFFF0ull);
            if (!regrow(trie, trie_used, cap)) return false;
            if (!x.fill_slots(trie + trie_used, cap - trie_used)) return xfail();
            trie_cap = cap;
        }
        base = trie_used;
        trie_used += slots;
        return true;
    }
    uint32_t buckets_for(uint64_t nodes) const { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(nodes + (tiny ? 0 : nodes * slack_num / 4 + 2), tiny ? 1 : 8), 0x7FFFFFF0ull); }
    // new tenant with room for `nodes` nodes; written to the host mirror (flush_directory uploads)
    bool create_tenant(const std::string& name, uint64_t nodes, uint32_t n_routes) {
        if (tenant_slot.count(name)) return true;
        if (!ensure_directory(tenant_slot.size() + 1)) return false;
        const uint32_t buckets = buckets_for(nodes);
        uint64_t base;
        if (!alloc_region(2ull * buckets, base)) return false;
        const uint64_t h = hash_name(name);
        TenantSlot t{};
        t.hash_lo = (uint32_t)h;
        t.hash_hi = (uint32_t)(h >> 32);
        t.name_off = (uint32_t)names_h.size();
        t.name_len = (uint32_t)name.size();
        memcpy(t.name12, name.data(), std::min<size_t>(name.size(), 12));
        t.root_plus = NONE;
        t.base = (uint32_t)base;
        t.buckets = buckets;
        t.n_routes = n_routes;
        names_h.insert(names_h.end(), name.begin(), name.end());
        const uint32_t d = place(t, dir_slots);
        tenant_slot[name] = d;
        new_dir_slots.push_back(d);
        dir_dirty = true;
        return true;
    }
    std::vector<uint32_t> new_dir_slots;
    bool flush_directory() { // upload the entries created since the last flush + the name pool
        if (names_h.size() + 16 > names_cap) {
            const uint32_t cap = (uint32_t)std::max<size_t>(names_h.size() * 2 + 64, 4096);
            if (names) {
                if (!x.sync()) return xfail();
                x.release(names);
            }
            names = (uint8_t*)x.alloc(cap);
            if (!names) return fail("out of memory (tenant names)");
            names_cap = cap;
            names_used = 0;
        }
        if (names_h.size() > names_used) {
            if (!x.copy_in(names + names_used, names_h.data() + names_used, names_h.size() - names_used)) return xfail();
            names_used = (uint32_t)names_h.size();
        }
        if (new_dir_slots.size() == tenant_slot.size()) { // bulk load: every entry is new, one copy (no live counters to preserve yet)
            if (!x.copy_in(dir, dir_h.data(), sizeof(TenantSlot) * (size_t)dir_slots)) return xfail();
        } else {
            for (uint32_t d : new_dir_slots)
                if (!x.copy_in(dir + d, &dir_h[d], sizeof(TenantSlot))) return xfail();
        }
        new_dir_slots.clear();
        dir_dirty = false;
        return true;
    }
    // regions listed by prepare_check (hbc.n_grow entries in ob.grow_list): move each into a larger region
    bool grow_flagged_regions(const OpBatch& ob) {
        if (hbc.n_grow == 0) return true;
        std::vector<uint32_t> gl(2 * (size_t)hbc.n_grow);
        if (!x.copy_out(gl.data(), ob.grow_list, sizeof(uint32_t) * gl.size())) return xfail();
        for (uint32_t g = 0; g < hbc.n_grow; g++) {
            const uint32_t d = gl[2 * g];
            TenantSlot t;
            if (!x.copy_out(&t, dir + d, sizeof(TenantSlot))) return xfail();
            const uint64_t need = std::max<uint64_t>(gl[2 * g + 1], (uint64_t)t.n_nodes + 1);
            const uint32_t nb = std::max<uint32_t>(buckets_for(need + need / 2), t.buckets < 0x3FFFFFFFu ? t.buckets * 2 : t.buckets);
            uint64_t base;
            if (!alloc_region(2ull * nb, base)) return false;
            if (!x.rehash(mut(), t.base, 2 * t.buckets, (uint32_t)base, nb, d)) return xfail();
            const uint32_t upd[2] = {(uint32_t)base, nb};
            if (!x.copy_in(&dir[d].base, upd, sizeof(upd))) return xfail(); // base and buckets are adjacent
            dir_h[d].base = (uint32_t)base;
            dir_h[d].buckets = nb;
            free_regions.push_back({t.base, 2ull * t.buckets});
            trie_garbage += 2ull * t.buckets;
        }
        if (!read_counters()) return xfail();
        if (hbc.err & ERR_REGION_FULL) return broke("region growth failed");
        hbc.n_grow = 0;
        return true;
    }
    bool reset_empty() { // an index with no routes (apply on a fresh engine)
        uint32_t err = 0;
        static const uint32_t zero_off[1] = {0};
        if (!rebuild_sorted(nullptr, zero_off, 0, err)) return false;
        built = true;
        generation++;
        return true;
    }
    static void sort_unique(const uint8_t* keys, const uint32_t* key_off, uint32_t n, std::vector<uint8_t>& bytes, std::vector<uint32_t>& off) {
        std::vector<std::string_view> v(n);
        for (uint32_t i = 0; i < n; i++) v[i] = std::string_view((const char*)keys + key_off[i], key_off[i + 1] - key_off[i]);
        auto less = [](std::string_view a, std::string_view b) {
            const size_t m = std::min(a.size(), b.size());
            const int c = m ? memcmp(a.data(), b.data(), m) : 0;
            return c < 0 || (c == 0 && a.size() < b.size());
        };
        std::sort(v.begin(), v.end(), less);
        v.erase(std::unique(v.begin(), v.end()), v.end());
        bytes.clear();
        off.assign(1, 0);
        for (auto& k : v) {
            bytes.insert(bytes.end(), k.begin(), k.end());
            off.push_back((uint32_t)bytes.size());
        }
        bytes.resize(bytes.size() + 16, 0);
    }

    // the bulk pipeline on strictly ascending keys; err receives the builder's error flags
    bool rebuild_sorted(const uint8_t* keys, const uint32_t* key_off, uint32_t n, uint32_t& err) {
        PhaseTimer pt(x);
        drop();
        pt.lap("rebuild: drop old index");
        const uint64_t kb = n ? key_off[n] : 0;
        bc = (BuildCounters*)x.alloc(sizeof(BuildCounters));
        if (!bc) return fail("out of memory");
        BuildCounters init{};
        init.rp_used = 1; // word 0 is never handed out: 0 means "no list"
        if (!x.copy_in(bc, &init, sizeof(init))) return xfail();
        kpool_used = 16; // offset 0 is never a key: kref == 0 means "no such route"
        if (!ensure_keys(tiny ? kb : kb + kb / 2 + (1u << 20)) || !ensure_ids(tiny ? n : (uint64_t)n + n / 2 + (1u << 16))) return false;
        if (!x.zero(kpool, 16)) return xfail();
        if (!grow_dict(tiny ? 64 : 1u << 16, tiny ? 64 : 1u << 20)) return false;
        if (!grow_route_pos(tiny ? 8 : 1u << 16)) return false;
        if (!ensure_directory(tiny ? 1 : 64)) return false;
        next_id = 0;
        n_routes = 0;
        if (n == 0) {
            if (!x.sync()) return xfail();
            return true;
        }
        if (!ensure_scratch(n, false)) return false;
        OpBatch ob = batch(n);
        ob.key_base = kpool_used;
        ob.id_base = 0;
        ob.put_rank = nullptr;
        ob.op = nullptr;
        ob.bulk = 1;
        if (!x.copy_in_async(kpool + kpool_used, keys, kb) || !x.copy_in_async(s_key_off, key_off, sizeof(uint32_t) * ((size_t)n + 1)))
            return xfail();
        pt.lap("rebuild: alloc + upload keys");
        // ---- sizes: tenants (runs of the sorted scan) and their exact node counts ----
        if (!zero_batch_counters()) return false;
        DistIndexMut ix = mut();
        if (!x.bulk_prepare(ix, ob) || !x.scan_flags(ob.flag, ob.order, n) || !read_counters()) return xfail();
        err = hbc.err;
        if (hbc.err & ERR_BAD_KEY) return fail("malformed route key");
        if (hbc.err & ERR_UNSORTED) return fail("keys are not strictly ascending");
        uint32_t n_ten = 0;
        if (!x.copy_out(&n_ten, ob.order + (n - 1), sizeof(uint32_t))) return xfail();
        if (n_ten > s_bt_cap) {
            auto rel = [&](auto*& p) {
                if (p) x.release((void*)p);
                p = nullptr;
            };
            rel(s_bt_first); rel(s_bt_nodes); rel(s_bt_keys); rel(s_bt_dir);
            const size_t want = (size_t)n_ten + 64;
            s_bt_first = (uint32_t*)x.alloc(4 * want);
            s_bt_nodes = (uint32_t*)x.alloc(4 * want);
            s_bt_keys = (uint32_t*)x.alloc(4 * want);
            s_bt_dir = (uint32_t*)x.alloc(4 * want);
            if (!s_bt_first || !s_bt_nodes || !s_bt_keys || !s_bt_dir) return fail("out of memory (builder scratch)");
            s_bt_cap = want;
            ob = batch(n);
            ob.key_base = kpool_used;
            ob.bulk = 1;
        }
        // run starts, then per-run key / node counts from a second scan (of nn[]; its output borrows the unknown-tenant list)
        if (!x.bulk_tenants(ix, ob, ob.order) || !x.scan_flags(ob.nn, ob.unknown_list, n) || !x.bulk_counts(ob, n_ten, ob.unknown_list))
            return xfail();
        std::vector<uint32_t> bt_first(n_ten), bt_nodes(n_ten), bt_keys(n_ten), bt_dir(n_ten);
        if (!x.copy_out(bt_first.data(), s_bt_first, 4 * (size_t)n_ten) || !x.copy_out(bt_nodes.data(), s_bt_nodes, 4 * (size_t)n_ten) ||
            !x.copy_out(bt_keys.data(), s_bt_keys, 4 * (size_t)n_ten))
            return xfail();
        pt.lap("rebuild: bulk prepare + tenant runs");
        // ---- tenants + regions ----
        if (!ensure_directory(n_ten)) return false;
        uint64_t total_slots = 0;
        for (uint32_t t = 0; t < n_ten; t++) total_slots += 2ull * buckets_for(bt_nodes[t]);
        {
            if (total_slots >= 0xFFFFFFF0ull) return fail("trie too large (2^32 slots)");
            const uint64_t cap = tiny ? total_slots + 2 : std::min<uint64_t>(total_slots + total_slots / 4 + (1u << 16), 0xFFFFFFF0ull); // head-room: new tenants, growth
            trie = (TrieSlot*)x.alloc(cap * sizeof(TrieSlot));
            if (!trie) return fail("out of device memory (trie)");
            trie_cap = cap;
            trie_used = 0;
            if (!x.fill_slots(trie, cap)) return xfail();
        }
        for (uint32_t t = 0; t < n_ten; t++) {
            const uint32_t ko = key_off[bt_first[t]];
            const size_t tlen = ((size_t)keys[ko + 1] << 8) | keys[ko + 2];
            const std::string name((const char*)keys + ko + 3, tlen);
            if (tenant_slot.count(name)) return fail("keys are not strictly ascending"); // cannot happen after the order check
            if (!create_tenant(name, bt_nodes[t], bt_keys[t])) return false;
            bt_dir[t] = tenant_slot[name];
        }
        if (!flush_directory()) return false;
        if (!x.copy_in(s_bt_dir, bt_dir.data(), 4 * (size_t)n_ten)) return xfail();
        pt.lap("rebuild: directory + regions");
        // ---- locate (re-run with a larger dictionary if it filled up), sort, groups ----
        for (int attempt = 0;; attempt++) {
            if (attempt == 10) return fail("rebuild: dictionary growth did not converge");
            if (!zero_batch_counters()) return false;
            ix = mut();
            if (!x.locate(ix, ob) || !read_counters()) return xfail();
            if (hbc.err & (ERR_STUCK | ERR_REGION_FULL)) return fail("rebuild: trie construction failed");
            if (!(hbc.err & ERR_DICT_FULL)) break;
            if (!grow_dict((uint64_t)dict_slots * 4, (uint64_t)dpool_cap * 4)) return false;
        }
        pt.lap("rebuild: locate");
        if (!x.zero(ob.group_done, n) || !x.sort_targets(ob)) return xfail();
        pt.lap("rebuild: sort by filter node");
        for (int attempt = 0;; attempt++) {
            if (attempt == 8) return fail("rebuild: id-list pool growth did not converge");
            if (!zero_batch_counters()) return false;
            ix = mut();
            if (!x.group(ix, ob) || !read_counters()) return xfail();
            if (hbc.err) return fail("rebuild: group step failed");
            if (hbc.n_deferred == 0) break;
            if (!grow_route_pos(hbc.rp_used + 2 * hbc.rp_need + 1024)) return false;
        }
        pt.lap("rebuild: groups");
        n_routes = n;
        kpool_used += (kb + 15) & ~15ull;
        next_id = n;
        // right-size the dictionary: load factor 1/4 keeps a lookup at one line, a small table stays cache resident
        {
            uint64_t want = 4096;
            while (want < (uint64_t)hbc.n_tokens * 4) want <<= 1;
            if (want != dict_slots && !grow_dict(want, dpool_cap)) return false;
        }
        if (!x.sync()) return xfail();
        release_scratch();
        pt.lap("rebuild: dictionary resize, cleanup");
        return true;
    }
};

} // namespace bmq
