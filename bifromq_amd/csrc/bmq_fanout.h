// bmq_fanout.h -- control of the fan-out grouping (bmq_fanout_core.h) over an Exec (DevExec: gfx950 kernels + hipCUB radix sort /
// scan on the engine stream; HostExec: host threads, for host-only engines and the CPU tests).  Besides what DistIndex needs
// (bmq_dist_index.h) the Exec provides:
//   bool fill_bytes(p, byte, n)
//   bool fo_fill(ix, st, b), fo_verify(ix, st, b), fo_keys(ix, st, b), fo_emit(b), fo_groups(st, b)      one lane per pair
//   bool sort_pairs32(keys_in, keys_out, vals_in, vals_out, n, end_bit)                              stable, ascending
//   bool scan_flags(in, out, n)                                                                      inclusive sum
#pragma once
#include "bmq_dist_index.h"
#include "bmq_fanout_core.h"

namespace bmq {

struct FanoutResult {
    uint32_t n_groups = 0;  // groups written (or needed, when it exceeds group_cap)
    uint32_t special = 0;   // bit 0: the shared-subscription group is present, bit 1: the group of dead route ids is present
    bool group_overflow = false;
};

template <class Exec> class Fanout {
public:
    Fanout(Exec& exec, DistIndex<Exec>& index) : x(exec), ix(index) {}
    ~Fanout() { drop(); }
    Fanout(const Fanout&) = delete;
    Fanout& operator=(const Fanout&) = delete;

    std::string error;
    uint32_t initial_table = 1024; // tests shrink it to force growth

    // All pointers are exec memory.  row_ptr[n_topics + 1] with row_ptr[n_topics] == total; out_topic / out_route [total];
    // group_off [group_cap + 1], group_rep [group_cap].
    bool group(const uint32_t* row_ptr, const uint32_t* ids, uint32_t n_topics, uint32_t total, uint32_t* out_topic, uint32_t* out_route,
               uint32_t* group_off, uint32_t* group_rep, uint32_t group_cap, FanoutResult& res) {
        res = FanoutResult{};
        if (!ix.built) return fail("no index");
        if (total == 0) {
            const uint32_t z = 0;
            return x.copy_in(group_off, &z, sizeof(z)) ? true : xfail();
        }
        if (total >= 0x7FFFFFF0u) return fail("more than 2^31 (topic, route) pairs in one batch");
        if (!ensure_state() || !ensure_scratch(total)) return false;
        if (grow_next) {
            grow_next = false;
            if (!reset_table(st.gt_cap * 4)) return false;
        }
        if constexpr (Exec::has_fanout_fast) {
            const int rc = group_fast(row_ptr, ids, n_topics, total, out_topic, out_route, group_off, group_rep, group_cap, res);
            if (rc >= 0) return rc == 1;
            // (-1: more deliverer keys than the fast path's LDS counters hold: the generic passes below)
        }
        for (int attempt = 0; attempt < 12; attempt++) {
            FanoutBatch b{};
            b.row_ptr = row_ptr;
            b.ids = ids;
            b.n_topics = n_topics;
            b.total = total;
            b.id_end = ix.next_id;
            b.key = s_key;
            b.key_sorted = s_key_sorted;
            b.pos = s_pos;
            b.pos_sorted = s_pos_sorted;
            b.head = s_key;      // the unsorted keys are dead after the sort
            b.head_scan = s_pos; // ... and so are the unsorted positions
            b.out_topic = out_topic;
            b.out_route = out_route;
            b.group_off = group_off;
            b.group_rep = group_rep;
            b.group_cap = group_cap;
            const DistIndexMut m = ix.mut();
            uint32_t end_bit = 1;
            while ((1u << end_bit) <= st.gt_cap + 1) end_bit++; // sort keys are <= gt_cap + 1
            const uint32_t keep[4] = {0u, used_slots, 0u, 0u};
            if (!x.copy_in(st.flags, keep, sizeof(keep))) return xfail();
            if (!x.fo_fill(m, st, b) || !x.fo_verify(m, st, b) || !x.fo_keys(m, st, b) ||
                !x.sort_pairs32(b.key, b.key_sorted, b.pos, b.pos_sorted, total, (int)end_bit) || !x.fo_emit(b) ||
                !x.scan_flags(b.head, b.head_scan, total) || !x.fo_groups(st, b))
                return xfail();
            uint32_t fl[4] = {0, 0, 0, 0};
            if (!x.copy_out(fl, st.flags, sizeof(fl))) return xfail();
            used_slots = fl[1];
            if (fl[0] & FO_ERR_COLLISION) { // two deliverer keys with one 64-bit hash: new seed, every route is mapped afresh
                seed++;
                if (!reset_table(st.gt_cap)) return false;
                continue;
            }
            if (fl[0] & FO_ERR_FULL) {
                if (st.gt_cap >= (1u << 29)) return fail("deliverer table too large");
                if (!reset_table(st.gt_cap * 4)) return false;
                continue;
            }
            grow_next = (uint64_t)used_slots * 2 > st.gt_cap && st.gt_cap < (1u << 29); // keep the table at most half full
            res.n_groups = fl[2];
            res.group_overflow = fl[2] > group_cap;
            if (!res.group_overflow && !read_special(b, res)) return false;
            return true;
        }
        return fail("fan-out grouping did not settle");
    }

    void drop() {
        rel(f_key16);
        rel(f_hist);
        rel(f_dense);
        rel(f_ctr);
        f_key_cap = f_hist_cap = f_dense_cap = 0;
        ix.fo_dgroup = nullptr;
        ix.fo_cap = 0;
        rel(st.dgroup);
        rel(st.gt_hash);
        rel(st.gt_rep);
        rel(st.flags);
        rel(s_key);
        rel(s_key_sorted);
        rel(s_pos);
        rel(s_pos_sorted);
        s_cap = 0;
        st = FanoutState{};
    }

private:
    Exec& x;
    DistIndex<Exec>& ix;
    FanoutState st{};
    uint64_t generation = ~0ull;
    uint32_t seed = 1, used_slots = 0;
    bool grow_next = false;
    uint32_t *s_key = nullptr, *s_key_sorted = nullptr, *s_pos = nullptr, *s_pos_sorted = nullptr;
    size_t s_cap = 0;

    // ---- fast path (device): bmq_fanout_kernels.h ----
    uint16_t *f_key16 = nullptr, *f_dense = nullptr;
    uint32_t *f_hist = nullptr, *f_ctr = nullptr; // f_ctr: [0] pairs without a group slot, [1] used slots
    size_t f_key_cap = 0, f_hist_cap = 0, f_dense_cap = 0;
    uint32_t n_used = 0;
    bool dense_stale = true;

    static uint32_t fast_tile() { // BMQ_FO_TILE: experiments only
        static const uint32_t t = [] {
            const char* v = bmq_env("BMQ_FO_TILE");
            const long n = v ? atol(v) : 0;
            return n >= 64 && n <= 2048 ? (uint32_t)(n / 64 * 64) : FO_TILE;
        }();
        return t;
    }
    // 1: done, 0: failed (error set), -1: not applicable
    int group_fast(const uint32_t* row_ptr, const uint32_t* ids, uint32_t n_topics, uint32_t total, uint32_t* out_topic, uint32_t* out_route,
                   uint32_t* group_off, uint32_t* group_rep, uint32_t group_cap, FanoutResult& res) {
        if constexpr (!Exec::has_fanout_fast) return -1;
        else {
            if (st.gt_cap > 0xFFF0u) return -1;
            if (!f_ctr && !fresh(f_ctr, 4)) return 0;
            for (int attempt = 0; attempt < 12; attempt++) {
                if (f_dense_cap < st.gt_cap) {
                    if (!x.sync()) return xfail() ? 1 : 0;
                    if (!fresh(f_dense, (size_t)st.gt_cap + 8)) return 0;
                    f_dense_cap = st.gt_cap;
                    dense_stale = true;
                }
                if (dense_stale) {
                    uint32_t nu = 0;
                    if (!x.fo_dense(st, f_dense, f_ctr + 1) || !x.copy_out(&nu, f_ctr + 1, sizeof(nu))) return xfail() ? 1 : 0;
                    n_used = nu;
                    dense_stale = false;
                }
                if (n_used + 2 > FO_MAX_BINS) return -1;
                FanoutFast f{};
                f.row_ptr = row_ptr;
                f.ids = ids;
                f.n_topics = n_topics;
                f.total = total;
                f.id_end = ix.next_id;
                f.tile = fast_tile();
                f.n_tiles = (total + f.tile - 1) / f.tile;
                f.n_bins = n_used + 2;
                f.key_bits = 1;
                while ((1u << f.key_bits) < f.n_bins) f.key_bits++;
                const size_t hist_n = (size_t)f.n_bins * f.n_tiles;
                if (hist_n >= 0x7FFFFFF0ull) return -1;
                if (total > f_key_cap || hist_n > f_hist_cap) {
                    if (!x.sync()) return xfail() ? 1 : 0;
                    if (total > f_key_cap) {
                        if (!fresh(f_key16, (size_t)total + total / 4 + 64)) return 0;
                        f_key_cap = (size_t)total + total / 4 + 64;
                    }
                    if (hist_n > f_hist_cap) {
                        if (!fresh(f_hist, hist_n + hist_n / 4 + 64)) return 0;
                        f_hist_cap = hist_n + hist_n / 4 + 64;
                    }
                }
                f.dense = f_dense;
                f.key16 = f_key16;
                f.hist = f_hist;
                f.out_topic = out_topic;
                f.out_route = out_route;
                f.group_off = group_off;
                f.group_rep = group_rep;
                f.group_cap = group_cap;
                f.need_fill = f_ctr;
                const DistIndexMut m = ix.mut();
                const uint32_t keep[4] = {0u, used_slots, 0u, 0u};
                if (!x.copy_in_async(st.flags, keep, sizeof(keep)) || !x.zero(f_ctr, sizeof(uint32_t)) || !x.fo_fast(m, st, f)) return xfail() ? 1 : 0;
                uint32_t fl[4] = {0, 0, 0, 0}, need = 0;
                if (!x.copy_out(fl, st.flags, sizeof(fl)) || !x.copy_out(&need, f_ctr, sizeof(need))) return xfail() ? 1 : 0;
                if (need == 0) {
                    res.n_groups = fl[2];
                    res.group_overflow = fl[2] > group_cap;
                    res.special = fl[3];
                    return 1;
                }
                // route ids without a group slot (the first batch of a generation, routes added since): map them -- the passes of
                // bmq_fanout_core.h over all pairs, cached ids fall through at once -- and run the split again
                FanoutBatch b{};
                b.row_ptr = row_ptr;
                b.ids = ids;
                b.n_topics = n_topics;
                b.total = total;
                b.id_end = ix.next_id;
                if (!x.copy_in_async(st.flags, keep, sizeof(keep)) || !x.fo_fill(m, st, b) || !x.fo_verify(m, st, b) || !x.copy_out(fl, st.flags, sizeof(fl)))
                    return xfail() ? 1 : 0;
                used_slots = fl[1];
                dense_stale = true;
                if (fl[0] & FO_ERR_COLLISION) { // two deliverer keys with one 64-bit hash: new seed, every route is mapped afresh
                    seed++;
                    if (!reset_table(st.gt_cap)) return 0;
                    continue;
                }
                if ((fl[0] & FO_ERR_FULL) || (uint64_t)used_slots * 2 > st.gt_cap) { // keep the table at most half full
                    if (st.gt_cap >= (1u << 29)) return fail("deliverer table too large") ? 1 : 0;
                    if (!reset_table(st.gt_cap * 4)) return 0;
                    if (st.gt_cap > 0xFFF0u) return -1;
                    continue;
                }
            }
            return fail("fan-out grouping did not settle") ? 1 : 0;
        }
    }

    template <class T> void rel(T*& p) {
        if (p) x.release(p);
        p = nullptr;
    }
    bool fail(const std::string& m) {
        error = m;
        return false;
    }
    bool xfail() { return fail(x.err.empty() ? "exec failure" : x.err); }
    template <class T> bool fresh(T*& p, size_t n) {
        rel(p);
        p = (T*)x.alloc(n * sizeof(T) + 16);
        return p ? true : fail("out of memory");
    }
    bool reset_table(uint32_t cap) {
        if (!x.sync()) return xfail();
        if (cap != st.gt_cap || !st.gt_hash) {
            if (!fresh(st.gt_hash, cap) || !fresh(st.gt_rep, cap)) return false;
            st.gt_cap = cap;
        }
        st.seed = seed;
        used_slots = 0;
        dense_stale = true;
        if (!x.zero(st.gt_hash, sizeof(unsigned long long) * (size_t)cap) || !x.fill_bytes(st.gt_rep, 0xFF, sizeof(uint32_t) * (size_t)cap) ||
            !x.fill_bytes(st.dgroup, 0xFF, sizeof(uint32_t) * (size_t)st.id_cap))
            return xfail();
        return true;
    }
    // route ids are renumbered by every rebuild / compact (generation) and the id space grows with the index
    bool ensure_state() {
        if (!st.flags && !fresh(st.flags, 4)) return false;
        const bool regen = generation != ix.generation;
        if (st.dgroup && !regen && st.id_cap >= ix.id_cap) return true;
        if (!x.sync()) return xfail();
        const uint32_t cap = ix.id_cap ? ix.id_cap : 1;
        if (!st.dgroup || st.id_cap < cap) {
            ix.fo_dgroup = nullptr;
            if (!fresh(st.dgroup, cap)) return false;
            st.id_cap = cap;
        }
        // from here on the index builder marks the ids it deletes in dgroup[] itself (DistIndexMut::fo_dgroup): ids it deleted before
        // are found by pass 1 (their key reference is 0)
        ix.fo_dgroup = st.dgroup;
        ix.fo_cap = st.id_cap;
        generation = ix.generation;
        // a grown id space keeps the table (hash -> slot stays valid); the per-id cache simply refills
        if (regen || !st.gt_hash) return reset_table(st.gt_hash && !regen ? st.gt_cap : initial_table);
        return x.fill_bytes(st.dgroup, 0xFF, sizeof(uint32_t) * (size_t)st.id_cap) ? true : xfail();
    }
    bool ensure_scratch(uint32_t total) {
        if (total <= s_cap) return true;
        if (!x.sync()) return xfail();
        const size_t want = (size_t)total + total / 4 + 64;
        if (!fresh(s_key, want) || !fresh(s_key_sorted, want) || !fresh(s_pos, want) || !fresh(s_pos_sorted, want)) {
            s_cap = 0;
            return false;
        }
        s_cap = want;
        return true;
    }
    // the special groups sort last: look at the sort keys of the last one / two group heads
    bool read_special(const FanoutBatch& b, FanoutResult& res) {
        uint32_t last_key = 0;
        if (!x.copy_out(&last_key, b.key_sorted + (b.total - 1), sizeof(uint32_t))) return xfail();
        if (last_key == st.gt_cap) res.special = 1;
        else if (last_key == st.gt_cap + 1) {
            res.special = 2;
            if (res.n_groups >= 2) {
                uint32_t off = 0, k = 0;
                if (!x.copy_out(&off, b.group_off + (res.n_groups - 2), sizeof(uint32_t)) || !x.copy_out(&k, b.key_sorted + off, sizeof(uint32_t)))
                    return xfail();
                if (k == st.gt_cap) res.special |= 1;
            }
        }
        return true;
    }
};

} // namespace bmq
