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
            if (!fresh(st.dgroup, cap)) return false;
            st.id_cap = cap;
        }
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
