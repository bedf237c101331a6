// bmq_fanout_core.h -- fan-out grouping: the step behind the dist match.  Per-item functions, BMQ_HD like the
// index builder (bmq_build_core.h): gfx950 kernels in bmq_exec_dev.h, host threads in bmq_exec_host.h, control in bmq_fanout.h.
//
// What it replaces.  DistWorkerCoProc.batchDist hands every topic's matched routes to DeliverExecutorGroup.submit
// (DW/DeliverExecutorGroup.java:112-241), which sends each NormalMatching through DeliverExecutor.send (DW/DeliverExecutor.java:85-90)
// into the deliverer's batcher, keyed by DelivererKey(subBrokerId, delivererKey) (bifromq-deliverer/src/main/java/org/apache/bifromq/
// deliverer/DelivererKey.java:22, DeliveryCall.java:30-40); BatchDeliveryCall.add (BatchDeliveryCall.java:71-76) then files the call under
// tenant -> message pack (= topic) -> set of MatchInfo.  So the (topic, route) pairs of a batch end up grouped by DelivererKey; inside
// a group by topic.  Here that is a segmented sort of the match CSR: every route id maps to a dense GROUP number of its
// (subBrokerId, delivererKey) -- the two outer parts of the receiverUrl "<subBrokerId> NUL <receiverId> NUL <delivererKey>"
// (SCHEMA/KVSchemaUtil.java:56-58, parsed by SCHEMA/cache/ReceiverCache.java:32-37) at the end of its route key -- and the pairs are
// radix-sorted by that number, stable, so that a group's pairs stay in (topic, route id) order.  Routes of shared subscriptions
// (flag 2 / 3: the receiver is chosen per message at send time, DeliverExecutorGroup.java:243-279) form one trailing group.
//
// Group numbers: an open-addressing table of 64-bit hashes of (subBrokerId bytes, delivererKey bytes); the slot a hash lands in is the
// group number, remembered per route id (dgroup[]), so a route's key is parsed once.  A slot also remembers the first route that
// claimed it; every other route mapped to the slot is checked against that route's bytes (fo_verify_one, a kernel later: no
// in-kernel waiting), so two deliverer keys are never merged: a 64-bit collision is reported and the table re-seeded.
#pragma once
#include "bmq_build_core.h"

namespace bmq {

// Words several lanes of one pass may write (always the same value, or values pass 2 treats alike): relaxed atomics on both sides.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> BMQ_HD void fo_store(T* p, T v) { shared_store(p, v); }
#else
template <class T> BMQ_HD void fo_store(T* p, T v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
#endif

constexpr uint32_t FO_UNSET = 0xFFFFFFFFu; // dgroup[id]: not computed yet (FO_DEAD_ID, bmq_build_core.h: the route has been deleted)
constexpr uint32_t FO_NEW = 0x80000000u;   // dgroup[id] flag: mapped in this pass, bytes not verified against the slot's first route yet
enum : uint32_t { FO_ERR_FULL = 1u, FO_ERR_COLLISION = 2u, FO_ERR_CSR = 4u };

struct FanoutState { // persistent between batches (exec memory)
    uint32_t* dgroup;              // [id_cap] group slot of route id, FO_UNSET, or the special slots below
    unsigned long long* gt_hash;   // [gt_cap] 0 = free
    uint32_t* gt_rep;              // [gt_cap] first route id mapped to the slot
    uint32_t gt_cap;               // power of two; special slots: gt_cap = shared subscriptions, gt_cap + 1 = dead route ids
    uint32_t id_cap;
    uint32_t seed;
    uint32_t* flags;               // [4]: err bits, slots claimed (approximate upper bound: one per successful claim), n_groups, spare
};

struct FanoutBatch {
    const uint32_t* row_ptr; // [n_topics + 1]
    const uint32_t* ids;     // [total]
    uint32_t n_topics, total;
    uint32_t id_end;         // ids handed out so far
    uint32_t *key, *key_sorted, *pos, *pos_sorted; // [total] sort key = group slot, value = element position
    uint32_t* head;          // [total] 1 = first pair of its group (in sorted order); then its inclusive scan in head_scan
    uint32_t* head_scan;
    uint32_t *out_topic, *out_route; // [total] the pairs, ordered by (group, topic, route id)
    uint32_t *group_off, *group_rep; // [group_cap (+1)]
    uint32_t group_cap;
};

// ---- the fast path (gfx950 kernels: bmq_fanout_kernels.h; the host executor has none and takes the generic passes) ----
constexpr uint32_t FO_TILE = 1024;     // pairs per wave (k_fo_scatter holds a tile in LDS: 10 bytes per pair)
constexpr uint32_t FO_SC_WAVES = 2;    // waves per workgroup of k_fo_scatter (LDS: 2 x (10 KB + 8 bytes per key) <= 64 KB)
constexpr uint32_t FO_WAVES = 4;       // waves per workgroup (independent: each owns its LDS slice)
constexpr uint32_t FO_MAX_BINS = 1026; // dense groups + the two special ones must fit the per-wave LDS counters

struct FanoutFast {
    const uint32_t* row_ptr;
    const uint32_t* ids;
    uint32_t n_topics, total, id_end;
    uint32_t tile;            // pairs per wave (FO_TILE; a multiple of 64)
    uint32_t n_tiles, n_bins; // n_bins = used group slots + 2; key n_bins - 2 = shared subscriptions, n_bins - 1 = dead ids
    uint32_t key_bits;        // bits needed for keys < n_bins
    const uint16_t* dense;    // [gt_cap] group-table slot -> dense group number
    uint16_t* key16;          // [total]
    uint32_t* hist;           // [n_bins * n_tiles] counts, then (after the scan) start offsets
    uint32_t *out_topic, *out_route;
    uint32_t *group_off, *group_rep;
    uint32_t group_cap;
    uint32_t* need_fill;      // [1] pairs whose route id has no group slot yet
};

// (subBrokerId, delivererKey) of the route key at kp[off, off+len): byte spans of receiverUrl part 0 and part 2.  false = a route of a
// shared subscription (flag 2 / 3).
struct DelivererSpan {
    unsigned long long b0, e0, b2, e2;
};
BMQ_HD bool fo_deliverer_span(const uint8_t* kp, unsigned long long off, unsigned long long len, DelivererSpan& s) {
    const unsigned long long end = off + len;
    const unsigned long long rlen = be16_at(kp, end - 2);
    const unsigned long long recv = end - 2 - rlen;
    if (kp[recv - 1] != 1) return false;
    const unsigned long long rend = end - 2;
    unsigned long long p = recv;
    s.b0 = recv;
    while (p < rend && kp[p] != 0) p++;
    s.e0 = p;
    if (p < rend) p++;
    while (p < rend && kp[p] != 0) p++; // receiverId
    if (p < rend) p++;
    s.b2 = p;
    while (p < rend && kp[p] != 0) p++;
    s.e2 = p;
    return true;
}
BMQ_HD unsigned long long fo_hash(const uint8_t* kp, const DelivererSpan& s, uint32_t seed) {
    unsigned long long h = TENANT_HASH_INIT ^ ((unsigned long long)seed * 0x9E3779B97F4A7C15ull);
    for (unsigned long long i = s.b0; i < s.e0; i++) h = tenant_hash_step(h, kp[i]);
    h = tenant_hash_step(h, 0xFFu);
    for (unsigned long long i = s.b2; i < s.e2; i++) h = tenant_hash_step(h, kp[i]);
    return tenant_hash_final(h); // never 0
}
BMQ_HD bool fo_same_deliverer(const uint8_t* kp, const DelivererSpan& a, const DelivererSpan& b) {
    return a.e0 - a.b0 == b.e0 - b.b0 && a.e2 - a.b2 == b.e2 - b.b2 && bytes_equal(kp, a.b0, kp, b.b0, a.e0 - a.b0) &&
           bytes_equal(kp, a.b2, kp, b.b2, a.e2 - a.b2);
}

// pass 1, one lane per pair: give the pair's route id a group slot if it has none yet
BMQ_HD void fo_fill_one(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b, uint32_t i) {
    const uint32_t id = b.ids[i];
    if (id >= b.id_end || id >= st.id_cap) return; // counted as dead in pass 3
    if (atom_load(st.dgroup + id) != FO_UNSET) return;
    const unsigned long long r = ix.kref[id];
    if (r == 0) { // deleted before it was ever mapped: remembered as dead
        fo_store(st.dgroup + id, FO_DEAD_ID);
        return;
    }
    DelivererSpan s;
    if (!fo_deliverer_span(ix.kpool, r & KREF_OFF_MASK, r >> KREF_LEN_SHIFT, s)) {
        fo_store(st.dgroup + id, st.gt_cap); // shared subscription
        return;
    }
    const unsigned long long h = fo_hash(ix.kpool, s, st.seed);
    const uint32_t mask = st.gt_cap - 1;
    uint32_t slot = (uint32_t)(h >> 17) & mask;
    for (uint32_t probe = 0; probe < st.gt_cap; probe++, slot = (slot + 1) & mask) {
        // thousands of routes share a deliverer key: look before claiming, so that only the first comers pay for an atomic
        unsigned long long cur = atom_load(st.gt_hash + slot);
        if (cur == 0) cur = atom_cas(st.gt_hash + slot, 0ull, h);
        if (cur == 0) { // claimed: this route is the slot's reference
            fo_store(st.gt_rep + slot, id);
            atom_add(st.flags + 1, 1u);
            fo_store(st.dgroup + id, slot); // nothing to verify against
            return;
        }
        if (cur == h) {
            fo_store(st.dgroup + id, slot | FO_NEW);
            return;
        }
    }
    atom_or(st.flags + 0, (uint32_t)FO_ERR_FULL);
}
// pass 2 (a kernel later, so that gt_rep of every slot claimed in pass 1 is visible): compare the bytes with the slot's first route
BMQ_HD void fo_verify_one(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b, uint32_t i) {
    const uint32_t id = b.ids[i];
    if (id >= b.id_end || id >= st.id_cap) return;
    const uint32_t g = atom_load(st.dgroup + id);
    if (g == FO_UNSET || g == FO_DEAD_ID || !(g & FO_NEW)) return;
    const uint32_t slot = g & ~FO_NEW;
    const uint32_t rep = st.gt_rep[slot];
    const unsigned long long r = ix.kref[id], rr = rep < st.id_cap ? ix.kref[rep] : 0ull;
    DelivererSpan a, c;
    bool same = true; // a reference route deleted since cannot be compared any more: its hash stands for it
    if (r != 0 && rr != 0 && fo_deliverer_span(ix.kpool, r & KREF_OFF_MASK, r >> KREF_LEN_SHIFT, a) &&
        fo_deliverer_span(ix.kpool, rr & KREF_OFF_MASK, rr >> KREF_LEN_SHIFT, c))
        same = fo_same_deliverer(ix.kpool, a, c);
    if (!same) atom_or(st.flags + 0, (uint32_t)FO_ERR_COLLISION);
    fo_store(st.dgroup + id, slot);
}
// pass 3, one lane per pair: sort key = group slot (shared subscriptions: gt_cap, dead ids: gt_cap + 1), value = position
// (a route deleted since the match is dead whatever the cache remembers of it)
BMQ_HD void fo_key_one(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b, uint32_t i) {
    const uint32_t id = b.ids[i];
    uint32_t g = id < b.id_end && id < st.id_cap ? st.dgroup[id] : FO_UNSET; // (a deleted route: FO_DEAD_ID, from the builder or pass 1)
    if (g == FO_UNSET || g == FO_DEAD_ID) g = st.gt_cap + 1;
    b.key[i] = g & ~FO_NEW;
    b.pos[i] = i;
}
// row of pair i: the last t with row_ptr[t] <= i (rows may be empty)
BMQ_HD uint32_t fo_row_of(const uint32_t* row_ptr, uint32_t n_topics, uint32_t i) {
    uint32_t lo = 0, hi = n_topics; // invariant: row_ptr[lo] <= i < row_ptr[hi]
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (row_ptr[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}
// pass 4, one lane per SORTED position j: the pair itself and whether it starts a group
BMQ_HD void fo_emit_one(const FanoutBatch& b, uint32_t j) {
    const uint32_t p = b.pos_sorted[j];
    b.out_route[j] = b.ids[p];
    b.out_topic[j] = fo_row_of(b.row_ptr, b.n_topics, p);
    b.head[j] = (j == 0 || b.key_sorted[j] != b.key_sorted[j - 1]) ? 1u : 0u;
}
// pass 5 (after the inclusive scan of head[]), one lane per sorted position: group heads fill the group table
BMQ_HD void fo_group_one(const FanoutState& st, const FanoutBatch& b, uint32_t j) {
    if (j + 1 == b.total) {
        const uint32_t n = b.head_scan[j];
        st.flags[2] = n;
        if (n <= b.group_cap) b.group_off[n] = b.total;
    }
    if (!b.head[j]) return;
    const uint32_t g = b.head_scan[j] - 1;
    if (g >= b.group_cap) return; // reported through flags[2]
    b.group_off[g] = j;
    const uint32_t slot = b.key_sorted[j];
    // a normal group is named by one of ITS routes of this batch (alive as of the match; the slot's first route may be gone by now)
    b.group_rep[g] = slot < st.gt_cap ? b.out_route[j] : (slot == st.gt_cap ? 0xFFFFFFFEu : 0xFFFFFFFFu);
}

} // namespace bmq
