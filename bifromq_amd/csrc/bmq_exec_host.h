// bmq_exec_host.h -- HostExec: runs the builder functions of bmq_build_core.h on host threads over host memory.
// Used by host-only engines (bmq_config.device = -1: build / inspect an index without a GPU -- matching still requires
// the device) and by the sanitizer fuzzers (tools/host_fuzz.cpp runs the identical builder code under ASan/UBSan and,
// with several threads, TSan).  See bmq_dist_index.h for the Exec concept.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "bmq_build_core.h"
#include "bmq_fanout_core.h"
#include "bmq_retain_core.h"

namespace bmq {

struct HostExec {
    std::string err;
    unsigned threads = 0; // 0 = hardware concurrency (capped)

    void* alloc(size_t bytes) { return malloc(bytes + 32); }
    void release(void* p) { free(p); }
    bool copy_in(void* d, const void* s, size_t n) {
        if (n) memcpy(d, s, n);
        return true;
    }
    bool copy_in_async(void* d, const void* s, size_t n) { return copy_in(d, s, n); }
    bool upload_async(void* d, const void* s, size_t n) { return copy_in(d, s, n); }
    bool uploads_done() { return true; }
    bool read_back_async(void*, const void*, size_t) { return true; } // (not gated: the stage-by-stage form reads after every stage)
    bool read_back_wait(void*, size_t) { return true; }
    bool copy_out(void* d, const void* s, size_t n) { return copy_in(d, s, n); }
    bool copy(void* d, const void* s, size_t n) { return copy_in(d, s, n); }
    bool zero(void* p, size_t n) {
        if (n) memset(p, 0, n);
        return true;
    }
    bool sync() { return true; }

    template <class F> void par(size_t n, F&& f) {
        unsigned hw = threads ? threads : std::min(std::thread::hardware_concurrency(), 16u);
        if (hw == 0) hw = 1;
        const size_t chunk = 256;
        const unsigned nth = (unsigned)std::min<size_t>(hw, (n + chunk - 1) / chunk);
        if (nth <= 1) {
            for (size_t i = 0; i < n; i++) f(i);
            return;
        }
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                const size_t b = next.fetch_add(chunk);
                if (b >= n) break;
                const size_t e = std::min(n, b + chunk);
                for (size_t i = b; i < e; i++) f(i);
            }
        };
        std::vector<std::thread> th;
        for (unsigned w = 1; w < nth; w++) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    }

    bool fill_slots(TrieSlot* p, uint64_t n) {
        par(n, [&](size_t i) { p[i] = FREE_SLOT; });
        return true;
    }
    bool prepare(const DistIndexMut& ix, const OpBatch& ob) {
        par(ob.n, [&](size_t i) { prepare_one(ix, ob, (uint32_t)i); });
        return true;
    }
    static constexpr bool gated = false; // (host threads: every stage is followed by its read of the counters, there is nothing to save)
    bool gate(BuildCounters*, uint32_t) { return true; }
    bool prepare_check(const DistIndexMut& ix, const OpBatch& ob, uint32_t n_dir) {
        par(n_dir, [&](size_t d) { prepare_check_one(ix, ob, (uint32_t)d); });
        return true;
    }
    bool bulk_prepare(const DistIndexMut& ix, const OpBatch& ob) {
        par(ob.n, [&](size_t i) { bulk_prepare_one(ix, ob, (uint32_t)i); });
        return true;
    }
    bool scan_flags(const uint32_t* in, uint32_t* out, uint32_t n) { // inclusive
        uint32_t s = 0;
        for (uint32_t i = 0; i < n; i++) out[i] = (s += in[i]);
        return true;
    }
    bool bulk_tenants(const DistIndexMut& ix, const OpBatch& ob, const uint32_t* scan) {
        par(ob.n, [&](size_t i) { bulk_tenants_one(ix, ob, (uint32_t)i, scan); });
        return true;
    }
    bool bulk_counts(const OpBatch& ob, uint32_t n_ten, const uint32_t* nn_incl) {
        par(n_ten, [&](size_t t) { bulk_counts_one(ob, (uint32_t)t, n_ten, nn_incl); });
        return true;
    }
    bool locate(const DistIndexMut& ix, const OpBatch& ob) {
        if (!ob.op) {
            par(ob.n, [&](size_t i) { locate_one(ix, ob, (uint32_t)i, 2u); });
            return true;
        }
        par(ob.n, [&](size_t i) { locate_one(ix, ob, (uint32_t)i, 3u); }); // every op; a delete that finds no filter is marked
        par(ob.n, [&](size_t i) { locate_one(ix, ob, (uint32_t)i, 4u); }); // the marked deletes, behind every put of the batch
        return true;
    }
    bool sort_targets(const OpBatch& ob) {
        std::iota(ob.order, ob.order + ob.n, 0u);
        std::stable_sort(ob.order, ob.order + ob.n, [&](uint32_t a, uint32_t b) { return ob.target[a] < ob.target[b]; });
        for (uint32_t i = 0; i < ob.n; i++) ob.sorted_target[i] = ob.target[ob.order[i]];
        return true;
    }
    bool group(const DistIndexMut& ix, const OpBatch& ob) {
        par(ob.n, [&](size_t p) { group_one(ix, ob, (uint32_t)p); });
        return true;
    }
    bool rehash(const DistIndexMut& ix, uint32_t old_base, uint32_t old_slots, uint32_t new_base, uint32_t new_buckets, uint32_t d) {
        for (uint32_t pass = 0; pass < 2; pass++) par(old_slots, [&](size_t s) { rehash_one(ix, old_base, new_base, new_buckets, (uint32_t)s, pass, d); });
        return true;
    }
    bool dict_rehash(const DictSlot* old, uint32_t old_slots, const DistIndexMut& ix) {
        par(old_slots, [&](size_t i) { dict_rehash_one(old, (uint32_t)i, ix); });
        return true;
    }
    bool find(const DistIndexMut& ix, const uint8_t* q, uint32_t tenant_len, uint32_t filter_len, uint32_t* out, uint32_t cap) {
        find_copy(ix, q, tenant_len, filter_len, out, cap);
        return true;
    }
    // ---- fan-out grouping (bmq_fanout.h) ----
    static constexpr bool has_fanout_fast = false; // the counting-sort path is gfx950 kernels only; host engines take the generic passes
    bool fill_bytes(void* p, int byte, size_t n) {
        if (n) memset(p, byte, n);
        return true;
    }
    bool fo_fill(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b) {
        par(b.total, [&](size_t i) { fo_fill_one(ix, st, b, (uint32_t)i); });
        return true;
    }
    bool fo_verify(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b) {
        par(b.total, [&](size_t i) { fo_verify_one(ix, st, b, (uint32_t)i); });
        return true;
    }
    bool fo_keys(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b) {
        par(b.total, [&](size_t i) { fo_key_one(ix, st, b, (uint32_t)i); });
        return true;
    }
    bool sort_pairs32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, int /*end_bit*/) {
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) { return keys_in[a] < keys_in[c]; });
        for (uint32_t i = 0; i < n; i++) {
            keys_out[i] = keys_in[order[i]];
            vals_out[i] = vals_in[order[i]];
        }
        return true;
    }
    bool fo_emit(const FanoutBatch& b) {
        par(b.total, [&](size_t j) { fo_emit_one(b, (uint32_t)j); });
        return true;
    }
    bool fo_groups(const FanoutState& st, const FanoutBatch& b) {
        par(b.total, [&](size_t j) { fo_group_one(st, b, (uint32_t)j); });
        return true;
    }
    bool gather_refs(const DistIndexMut& ix, const uint32_t* ids, uint32_t n, uint32_t id_end, unsigned long long* out) {
        par(n, [&](size_t i) { gather_ref_one(ix, ids, (uint32_t)i, id_end, out); });
        return true;
    }
    bool gather_bytes(const DistIndexMut& ix, const unsigned long long* refs, const uint64_t* offs, uint32_t n, uint8_t* out) {
        par(n, [&](size_t i) { gather_bytes_one(ix, refs, offs, (uint32_t)i, out); });
        return true;
    }
    // ---- retain direction (bmq_retain_core.h) ----
    bool r_locate(const RetainMut& m, const RetainOps& ob) {
        par(ob.n, [&](size_t i) { rlocate_one(m, ob, (uint32_t)i, 0u); });
        par(ob.n, [&](size_t i) { rlocate_one(m, ob, (uint32_t)i, 1u); });
        return true;
    }
    bool r_commit(const RetainMut& m, const RetainOps& ob) {
        par(ob.n, [&](size_t i) { rcommit_one(m, ob, (uint32_t)i); });
        return true;
    }
    bool r_rank(const RetainMut& m, uint32_t n_words) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < n_words; w++) {
            m.dead_rank[w] = run;
            run += (uint32_t)__builtin_popcountll(m.dead_bits[w]);
        }
        m.dead_rank[n_words] = run;
        return true;
    }
    bool r_rehash(const RetainMut& m, uint32_t n_nodes) {
        par(n_nodes, [&](size_t i) { ov_rehash_one(m, (uint32_t)i); });
        return true;
    }
    bool r_topic_lens(const RetainMut& m, const uint32_t* ids, uint32_t n, uint32_t* lens) {
        par(n, [&](size_t i) { ov_topic_len_one(m, ids, (uint32_t)i, lens); });
        return true;
    }
    bool r_topic_write(const RetainMut& m, const uint32_t* ids, uint32_t n, const unsigned long long* offs, uint8_t* out) {
        par(n, [&](size_t i) { ov_topic_write_one(m, ids, (uint32_t)i, offs, out); });
        return true;
    }
    bool r_gc_select(const RetainMut& m, const GcQuery& q, uint8_t* /*flags*/, uint32_t* out_ids, uint32_t* out_count) {
        uint32_t n = 0;
        for (uint32_t id = 0; id < q.n_ids; id++)
            if (gc_flag_one(m, q, id)) out_ids[n++] = id;
        *out_count = n;
        return true;
    }
    bool r_find_tenant(const RetainMut& m, const uint8_t* name, uint32_t len, uint32_t* out) {
        LevelScan lv;
        unsigned long long pos = 0;
        lv.start = 0;
        scan_level_bytes<0u>(name, pos, len, lv.h, lv.inl, lv.len);
        out[0] = ov_find(m.onodes, m.oedges, m.oedge_mask, m.opool, 0u, lv.h.h1, lv.h.h2, lv.len, name, 0);
        return true;
    }
};

} // namespace bmq
