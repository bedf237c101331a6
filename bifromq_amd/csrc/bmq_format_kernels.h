// bmq_format_kernels.h -- gfx950 kernels of the RANGES result format (include/bmq.h: bmq_match_wait_ranges).
//
// What k_walk leaves behind for a topic is a short list of MATCHED RANGES: every matched filter node owns the route ids
// begin .. begin + count - 1 (its routes are neighbours in KV key order, SCHEMA/KVSchemaUtil.java:91-117), or -- for nodes touched by
// bmq_routes_apply since the last rebuild -- a list of ids in the index's side array (RANGE_INDIRECT).  k_expand turns the ranges into
// the id CSR (18 ids per topic on C3: 73 MB per million topics over PCIe); a host consumer that walks the routes anyway can expand
// (begin, count) itself, so this format ships the ranges: range_ptr[n + 1], 8 bytes per range, and the ids of the indirect lists copied
// into a side array of the RESULT (the consumer needs nothing of the index).
//   k_fmt_count : per topic, its number of ranges and of side ids            -> two exclusive prefix sums (hipcub) -> range_ptr, side_ptr
//   k_fmt_emit  : per topic, its ranges in ascending order of their first id (insertion sort while copying: lists are 1-8 long;
//                 lists longer than FMT_SORT_MAX are copied as they are), indirect lists copied to the side array
// A row whose consecutive ranges overlap after ordering (interleaved id sets: possible only after churn) is counted in sums[2]: the
// consumer orders the expanded ids of such rows itself (it sees the overlap while expanding).
#pragma once
#include <hip/hip_runtime.h>

#include "bmq_dist_kernels.h"

namespace bmq {

constexpr uint32_t FMT_SORT_MAX = 32;

struct FmtArgs {
    DistIndexView ix;
    const uint32_t *pair_off, *pair_cnt;
    const MatchRange* pairs;
    const Counters* ctr;
    uint32_t n_topics;
    uint32_t *cnt_r, *cnt_s;         // [n + 1]
    uint32_t *range_ptr, *side_ptr;  // [n + 1]: the scanned counts
    MatchRange* out_ranges;
    unsigned long long range_cap;
    uint32_t* out_side;
    unsigned long long side_cap;
    unsigned long long* sums;        // [0] ranges, [1] side ids, [2] rows whose ranges overlap, [3] bit 0: buffers too small, bit 1: batch incomplete
};

__global__ __launch_bounds__(256) void k_fmt_count(FmtArgs f) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t > f.n_topics) return;
    uint32_t nr = 0, ns = 0;
    if (t < f.n_topics && !(f.ctr->status & (ST_RERUN | ST_RANGE))) {
        nr = f.pair_cnt[t];
        const MatchRange* p = f.pairs + f.pair_off[t];
        for (uint32_t i = 0; i < nr; i++) {
            const uint32_t c = p[i].count;
            if (c & RANGE_INDIRECT) ns += c & ~RANGE_INDIRECT;
        }
    }
    f.cnt_r[t] = nr;
    f.cnt_s[t] = ns;
    if (t == 0) {
        f.sums[2] = 0;
        f.sums[3] = (f.ctr->status & (ST_RERUN | ST_RANGE)) ? 2ull : 0ull;
    }
}

__device__ __forceinline__ uint32_t fmt_first_id(const FmtArgs& f, const MatchRange& r) { return (r.count & RANGE_INDIRECT) ? f.out_side[r.begin] : r.begin; }
__device__ __forceinline__ uint32_t fmt_last_id(const FmtArgs& f, const MatchRange& r) {
    const uint32_t c = r.count & ~RANGE_INDIRECT;
    return (r.count & RANGE_INDIRECT) ? f.out_side[r.begin + c - 1] : r.begin + c - 1;
}

__global__ __launch_bounds__(256) void k_fmt_emit(FmtArgs f) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long n_r = f.range_ptr[f.n_topics], n_s = f.side_ptr[f.n_topics];
    const bool fits = n_r <= f.range_cap && n_s <= f.side_cap;
    if (t == 0) {
        f.sums[0] = n_r;
        f.sums[1] = n_s;
        if (!fits) f.sums[3] |= 1ull;
    }
    if (t >= f.n_topics || !fits) return;
    const uint32_t rp = f.range_ptr[t], nr = f.range_ptr[t + 1] - rp;
    if (nr == 0) return;
    uint32_t sp = f.side_ptr[t];
    const MatchRange* p = f.pairs + f.pair_off[t];
    MatchRange* out = f.out_ranges + rp;
    const bool order = nr <= FMT_SORT_MAX;
    for (uint32_t i = 0; i < nr; i++) {
        MatchRange r = p[i];
        const uint32_t len = r.count & ~RANGE_INDIRECT;
        if (r.count & RANGE_INDIRECT) { // the list moves into the result: begin = its place in out_side
            for (uint32_t k = 0; k < len; k++) f.out_side[sp + k] = f.ix.route_pos[r.begin + k];
            r.begin = sp;
            sp += len;
        }
        uint32_t j = i;
        if (order && len) {
            const uint32_t key = fmt_first_id(f, r);
            while (j > 0) {
                const MatchRange q = out[j - 1];
                if ((q.count & ~RANGE_INDIRECT) != 0 && fmt_first_id(f, q) <= key) break;
                out[j] = q;
                j--;
            }
        }
        out[j] = r;
    }
    // ascending ids = every range starts behind the end of the one before it (ids ascend inside a range by construction)
    bool bad = false;
    uint32_t last = 0;
    bool have = false;
    for (uint32_t i = 0; i < nr; i++) {
        const MatchRange q = out[i];
        if ((q.count & ~RANGE_INDIRECT) == 0) continue;
        if (have && fmt_first_id(f, q) <= last) bad = true;
        last = fmt_last_id(f, q);
        have = true;
    }
    if (bad) atomicAdd(f.sums + 2, 1ull);
}

} // namespace bmq
