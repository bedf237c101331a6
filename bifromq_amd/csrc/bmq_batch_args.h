// bmq_batch_args.h -- what the host and the kernels of one match batch share: the status bits, the per-batch counters and the argument
// block of the dist-direction kernels (k_dedup, k_walk, k_walk_slow, k_fill, k_expand, k_sort_rows; the retain direction fills the same
// block for k_expand / k_sort_rows).  Plain structs: included by bmq_dist_kernels.h (device) and by the wave emulator of tools/ (host).
#pragma once
#include <cstdint>

#include "bmq_layout.h"

namespace bmq {

// status bits raised by kernels, resolved by the host in bmq_match_finish()
enum : uint32_t {
    ST_NEED_PAIRS = 1u,     // matched-range buffer too small
    ST_NEED_SLOW = 2u,      // slow-topic list too small
    ST_NEED_SCRATCH = 4u,   // slow-path scratch too small
    ST_NOSPACE = 8u,        // caller's id buffer too small
    ST_RANGE = 16u,         // >= 2^32 ids
    ST_NEED_SORTLIST = 32u, // fix-up list too small
    ST_NEED_SPILL = 64u,    // range spill buffer too small
    ST_NEED_ADJ = 1024u,    // (128 and 512 are the retain direction's, bmq_retain_args.h) the dense batch's topic-byte buffer too small (bmq_dedup_adj_kernels.h;
                            // Counters.adj_bytes says what it takes).  Not part of ST_RERUN: the kernels behind it need not know -- the dense batch
                            // is empty then, every row comes out empty, and the host runs the batch again with the larger buffer
    ST_WANT_MIXED = 256u    // a wave held topics of many tenants (the batch is not grouped by tenant): the batch runs again through the MIXED instantiation
};
constexpr uint32_t ST_RERUN = ST_NEED_PAIRS | ST_NEED_SLOW | ST_NEED_SCRATCH | ST_NEED_SPILL;

struct Counters { // one per batch slot, zeroed behind every batch (k_reset)
    unsigned long long pair_alloc;
    unsigned long long scratch_alloc; // in uint32 units
    unsigned long long spill_alloc;   // in records
    unsigned long long n_visit;
    unsigned long long total_ids;
    unsigned long long n_ranges;
    unsigned long long topic_bytes;
    uint32_t slow_count;
    uint32_t sort_count;
    uint32_t status;
    uint32_t n_walked; // rows of the dense batch the walk ran on (bmq_dedup_adj_kernels.h); 0: the walk ran on the batch itself
    uint32_t adj_bytes; // ST_NEED_ADJ: topic bytes of the dense batch
    uint32_t heavy_count; // blocks k_walk put on BatchArgs.heavy_list (may exceed heavy_cap: the blocks beyond it are not split)
    uint32_t rw_next[64]; // k_retain_walk: the next quad of filters of each partition of the batch (RW_PARTS)
};
constexpr uint32_t RW_PARTS = 64;
// k_expand's tail (round 6): a batch ORDERED by (tenant, topic) -- the shape BatchDistRequest has -- puts the rows under a hot prefix, which
// match the same big filters, side by side: 1 % of its 64-row blocks hold 4-7 x the ranges and ids of the average block, their waves live
// 6 x as long (150 k clocks, the duration of the whole launch), and the ones at the end of the dispatch order ARE the launch's tail
// (+ 48 % on C3's batch ordered).  Whoever writes a block's sums (k_walk, k_fill, k_fill_adj) lists the block if it holds split_ranges
// ranges or split_ids ids or more; k_expand's grid starts with three helper waves per listed block (rows 16-31, 32-47, 48-63: the heavy work
// begins first), the block's own wave expands rows 0-15.  The thresholds are the host's: a multiple of the mean block of the batches before
// (bmq_engine.hip, launch_dist), so that the list -- heavy_cap = n_blocks / EXPAND_HEAVY_DIV entries -- holds the heavy blocks of the batch and
// not a random part of them (a list that overflows leaves heavy blocks whole wherever they lie, also at the end: measured, no gain then).
#ifndef BMQ_EXPAND_SPLIT_RANGES
#define BMQ_EXPAND_SPLIT_RANGES 1024 // thresholds of an engine's first large batch
#endif
#ifndef BMQ_EXPAND_SPLIT_IDS
#define BMQ_EXPAND_SPLIT_IDS 3072
#endif
#ifndef BMQ_EXPAND_SPLIT_MULT_X16
#define BMQ_EXPAND_SPLIT_MULT_X16 32 // afterwards: 32 / 16 = 2 x the mean block (measured 1.5 / 1.75 / 2 / 2.5: profiles/r06/extras/ab_expand_split.txt)
#endif
#ifndef BMQ_EXPAND_HEAVY_DIV
#define BMQ_EXPAND_HEAVY_DIV 4
#endif
constexpr uint32_t EXPAND_SPLIT_RANGES = BMQ_EXPAND_SPLIT_RANGES, EXPAND_SPLIT_IDS = BMQ_EXPAND_SPLIT_IDS, EXPAND_SPLIT_MULT_X16 = BMQ_EXPAND_SPLIT_MULT_X16;
constexpr uint32_t EXPAND_HEAVY_DIV = BMQ_EXPAND_HEAVY_DIV;
constexpr uint32_t EXPAND_PARTS = 4; // a listed block is expanded by this many waves, 64 / EXPAND_PARTS rows each
constexpr uint32_t SUPER_SHIFT = 8;   // id counts are summed per 2^SUPER_SHIFT waves (super_sums) on top of the per-wave counts
constexpr uint32_t SUPER_STRIDE = 16; // ... one sum per 128-byte line: 256 waves bump each, neighbours must not share a line

// Contiguous space in the matched-range buffer is handed out by N_SUB independent allocators, each owning one slice
// of the buffer and living in its own 128-byte line: a single counter bumped by every wave of a batch (15 625 waves
// for 1M topics) serialises in the L2 atomic unit and was measured to set the kernel's duration.
constexpr uint32_t N_SUB = 64;
struct alignas(128) SubAlloc {
    unsigned long long used;
    unsigned long long pad[15];
};
struct BatchArgs {
    DistIndexView ix;
    // inputs (device)
    const uint8_t* tenants;
    const uint32_t* tenant_off;
    uint32_t n_tenants;
    const uint32_t* topic_tenant;
    const uint8_t* topics;
    const uint32_t* topic_off;
    uint32_t n_topics;
    // per-batch scratch (device)
    uint32_t* pair_off;      // [n_topics]
    uint32_t* pair_cnt;      // [n_topics]
    uint32_t* route_cnt;     // [n_topics]
    MatchRange* pairs;
    unsigned long long pair_cap;
    SubAlloc* subs;          // [2 * N_SUB] allocators of `pairs` (first N_SUB) and of `spill` (second N_SUB)
    unsigned long long* super_sums; // [(n_blocks >> SUPER_SHIFT + 1) * SUPER_STRIDE] ids per 2^SUPER_SHIFT blocks (zeroed by k_reset)
    uint4* blk_stats;        // [n_blocks] per 64-topic block: {nodes visited, ranges, topic bytes, split} written by k_walk; the last
                             // k_expand wave of every super-block sums its 256 records into ctr (null: retain direction).  split != 0: the
                             // block is on heavy_list -- its own k_expand wave expands rows 0-15 only, three helper waves the rest
    uint32_t* heavy_list;    // [heavy_cap] blocks whose 64 rows hold split_ranges ranges / split_ids ids or more (null: no splitting)
    uint32_t heavy_cap;
    uint32_t split_ranges, split_ids;
    uint4* spill;            // LDS range buffer flushes: {begin, count, topic-local, 0}
    unsigned long long spill_cap;
    unsigned long long* wave_sums; // [n_blocks] ids per 64-topic block
    uint32_t n_blocks;
    uint32_t tpw_shift;      // a wave owns 2^tpw_shift topics (6 = all 64 lanes; small batches use 4 or 2: more, shorter waves)
    uint32_t* slow_list;
    uint32_t slow_cap;
    uint32_t* scratch;
    unsigned long long scratch_cap; // uint32 units
    uint32_t* sort_list;
    uint32_t sort_cap;
    Counters* ctr;
    // outputs (device)
    uint32_t* out_row_ptr;
    uint32_t* out_ids;
    unsigned long long out_capacity;
    unsigned long long* out_total;
    // LDS geometry
    uint32_t qcap; // pow2
    uint32_t pcap; // >= 128
    // in-batch de-duplication (bmq_dedup_kernels.h); rep == nullptr: off
    uint32_t* rep;                  // [n_topics] the row that stands for this row's (tenant, topic): itself, or an identical earlier claimant
                                    // (bmq_config.dedup_sorted: the walk kernels get the dense batch's identity, the kernels behind them any non-null pointer)
    uint32_t* visit_cnt;            // [n_topics] nodes discovered for the row's topic (written for representatives)
    unsigned long long* dd_table;   // open addressing, dd_mask + 1 entries: generation << 56 | hash tag << 32 | row
    uint32_t dd_mask;
    uint32_t dd_gen;                // 1..255
    uint32_t debug_flags; // BMQ_DEBUG env, read by builds with -DBMQ_EXPERIMENTS=1 only (tools/build_variant.sh): 1 = stop after tokenising, 2 = fill dbg_wave
    uint4* dbg_wave;      // [n_blocks] {phase 1, phase 2, phase 3 shader clocks, rounds | items << 8} of every k_walk wave, or null
};

// The profiling experiments of the kernels (BatchArgs.debug_flags: per-wave clocks, the residency census, alternative block orders, ...) are
// compiled in by -DBMQ_EXPERIMENTS=1 only: the library that ships is the kernel that is measured, without their branches.
#ifndef BMQ_EXPERIMENTS
#define BMQ_EXPERIMENTS 0
#endif
#define BMQ_DBG(a, bits) (BMQ_EXPERIMENTS != 0 && ((a).debug_flags & (bits)) != 0)

// De-duplication of a batch that arrives ORDERED by (tenant, topic) (bmq_config.dedup_sorted; bmq_dedup_adj_kernels.h): equal rows are
// neighbours, the first row of a run (its HEAD) stands for the run, and the heads are copied into a dense batch of their own that the walk
// kernels run on unchanged -- (c_topics, c_off, c_tenant) in the layout of (topics, topic_off, topic_tenant), the rows behind the last head
// marked "no such tenant" (never walked).  Per-block sums use the super-block scheme of the id counts above.
struct AdjArgs {
    const uint8_t* topics;
    const uint32_t* topic_off;
    const uint32_t* topic_tenant;
    uint32_t n_topics;
    uint32_t n_blocks;
    uint32_t tpw_shift;
    uint32_t* drow;                 // [n_topics] out: the dense row that answers for the row (its own, or its run head's)
    unsigned long long* blk_mask;   // [n_blocks] bit l: row l of the block is a head (differs from the row before it)
    unsigned long long* blk_cnt;    // [n_blocks] heads << 32 | bytes of the heads' topics, per block of 2^tpw_shift rows
    unsigned long long* super_cnt;  // [n_super * SUPER_STRIDE] entry 0 of a line: the sum of blk_cnt over 2^SUPER_SHIFT blocks (zeroed in front of every batch)
    uint8_t* c_topics;              // the dense batch (16-byte aligned, readable 16 bytes past the last byte like every packed input)
    unsigned long long c_cap;       // ... bytes it may hold; a batch whose heads need more raises ST_NEED_ADJ, walks nothing and runs again
    uint32_t* c_off;                // [n_topics + 1]
    uint32_t* c_tenant;             // [n_topics]
    uint32_t* c_rep;                // [n_topics] the identity: every dense row is walked
    Counters* ctr;                  // n_walked = number of heads
};

#if defined(__HIPCC__) || defined(BMQ_WAVE_EMU)
// (one lane of the wave that knows a block's totals) lists the block for k_expand's helpers if it is heavy; the return value goes into blk_stats[blk].w
__device__ __forceinline__ uint32_t heavy_mark(const BatchArgs& d, uint32_t blk, unsigned long long ranges, unsigned long long ids) {
    if (d.heavy_list == nullptr || (ranges < d.split_ranges && ids < d.split_ids)) return 0u;
    const uint32_t hs = atomicAdd(&d.ctr->heavy_count, 1u);
    if (hs >= d.heavy_cap) return 0u;
    d.heavy_list[hs] = blk;
    return 1u;
}
// n contiguous entries of `pairs` from sub-allocator `key` (false: the slice is full, the batch is re-run with a larger buffer)
__device__ __forceinline__ bool pair_alloc(SubAlloc* subs, unsigned long long pair_cap, uint32_t key, uint32_t n,
                                           unsigned long long& base) {
    const unsigned long long slice = pair_cap / N_SUB;
    const uint32_t s = key & (N_SUB - 1);
    const unsigned long long off = atomicAdd(&subs[s].used, (unsigned long long)n);
    base = (unsigned long long)s * slice + off;
    return off + n <= slice;
}
#endif

} // namespace bmq
