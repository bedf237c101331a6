// expand_emu.cpp -- host-side logic test of k_expand (bifromq_amd/csrc/bmq_expand_kernel.h) under the wave64 emulator of wave_emu.h.
// Test tooling: it checks the kernel's LOGIC (pass boundaries, rank sort, short / streamed ranges, indirect ranges, gathered lists,
// capacity overflow) against a plain restatement of its contract on thousands of random batches; what the GPU does with the same source
// is what tests/ (-m gpu) check against the oracle.
//   g++ -O1 -g -std=c++17 -I bifromq_amd/csrc tools/emu/expand_emu.cpp -o build/expand_emu && build/expand_emu [cases] [seed]
//   -DEMU_OLD: the round-2/3 kernel (tools/emu/k_expand_r3.h) through the same cases -- the cross-check of this file's restatement.
#define BMQ_WAVE_EMU 1
#include "wave_emu.h"

#include <random>
#include <set>
#include <vector>

#include "bmq_batch_args.h"

#ifdef EMU_OLD
#define BMQ_EXP_WAVES 1
namespace bmq {
inline uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += o;
    }
    total = __shfl(inc, 63);
    return inc - v;
}
inline unsigned long long wave_sum_u64(unsigned long long v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
} // namespace bmq
#include "k_expand_r3.h"
#else
#include "bmq_expand_kernel.h"
#endif

using namespace bmq;

struct Case {
    uint32_t n_topics, tpw_shift;
    std::vector<uint32_t> pair_off, pair_cnt, route_cnt;
    std::vector<MatchRange> pairs;
    std::vector<uint32_t> route_pos;
    // expected
    std::vector<uint32_t> row_ptr, ids;
    std::set<uint32_t> bad_rows;
};

static uint32_t first_id(const Case& c, const MatchRange& r) { return (r.count & RANGE_INDIRECT) ? c.route_pos[r.begin] : r.begin; }
static uint32_t last_id(const Case& c, const MatchRange& r) {
    const uint32_t n = r.count & ~RANGE_INDIRECT;
    return (r.count & RANGE_INDIRECT) ? c.route_pos[r.begin + n - 1] : r.begin + n - 1;
}

struct Knobs {
    uint32_t n_topics, tpw_shift;
    double p_empty_row, p_shuffle, p_indirect, p_long, p_huge_row, p_interleave, p_zero_len;
    uint32_t max_np;
    bool gather;
};

static Case make_case(std::mt19937_64& rng, const Knobs& k) {
    Case c;
    c.n_topics = k.n_topics, c.tpw_shift = k.tpw_shift;
    auto uni = [&](uint32_t lo, uint32_t hi) { return lo + (uint32_t)(rng() % (hi - lo + 1)); };
    auto coin = [&](double p) { return (double)(rng() >> 11) / (double)(1ull << 53) < p; };
    c.route_pos.assign(8, 0xDEADBEEFu); // slot 0..7 unused
    std::vector<std::vector<MatchRange>> rows(k.n_topics);
    for (uint32_t t = 0; t < k.n_topics; t++) {
        if (coin(k.p_empty_row)) continue;
        uint32_t np = coin(k.p_huge_row) ? uni(33, 700) : (coin(0.15) ? uni(6, k.max_np) : uni(1, 5));
        uint32_t id = uni(0, 1000);
        for (uint32_t i = 0; i < np; i++) {
            id += uni(0, 9);
            uint32_t len = coin(k.p_long) ? (coin(0.2) ? uni(300, 5000) : uni(60, 200)) : (coin(0.3) ? uni(1, 40) : uni(1, 6));
            if (coin(k.p_zero_len)) len = 0;
            MatchRange r;
            if (len && coin(k.p_indirect)) {
                r.begin = (uint32_t)c.route_pos.size();
                r.count = len | RANGE_INDIRECT;
                uint32_t v = id;
                for (uint32_t j = 0; j < len; j++) {
                    c.route_pos.push_back(v);
                    v += 1 + (coin(k.p_interleave) ? uni(0, 40) : 0); // a sparse list reaches into the ids of the ranges after it
                }
                id = coin(k.p_interleave) ? id + len : v;
            } else {
                r.begin = id, r.count = len;
                id += len ? len : 1; // (an empty range still gets a begin of its own: keys stay distinct)
            }
            rows[t].push_back(r);
        }
        if (coin(k.p_shuffle)) std::shuffle(rows[t].begin(), rows[t].end(), rng);
    }
    // layout of `pairs`: per wave one contiguous piece (row after row), or every row's list on its own
    c.pair_off.assign(k.n_topics, 0), c.pair_cnt.assign(k.n_topics, 0), c.route_cnt.assign(k.n_topics, 0);
    const uint32_t tpw = 1u << k.tpw_shift;
    if (!k.gather) {
        for (uint32_t t0 = 0; t0 < k.n_topics; t0 += tpw) {
            for (uint32_t g = uni(0, 5); g; g--) c.pairs.push_back(MatchRange{0x7FFFFFFFu, 0x7FFFFFFFu}); // other waves' space
            for (uint32_t t = t0; t < std::min(t0 + tpw, k.n_topics); t++) {
                c.pair_off[t] = rows[t].empty() ? uni(0, 100000) : (uint32_t)c.pairs.size(); // (rows without ranges: whatever the walk left)
                for (auto& r : rows[t]) c.pairs.push_back(r);
            }
        }
    } else {
        std::vector<uint32_t> order(k.n_topics);
        for (uint32_t t = 0; t < k.n_topics; t++) order[t] = t;
        std::shuffle(order.begin(), order.end(), rng);
        for (uint32_t t : order) {
            c.pair_off[t] = (uint32_t)c.pairs.size();
            for (auto& r : rows[t]) c.pairs.push_back(r);
            if (coin(0.3)) c.pairs.push_back(MatchRange{0x7FFFFFFFu, 0x7FFFFFFFu});
        }
    }
    for (uint32_t i = 0; i < 8; i++) c.pairs.push_back(MatchRange{0, 0}); // (slack behind the last list)
    // the contract, restated
    c.row_ptr.assign(k.n_topics + 1, 0);
    for (uint32_t t = 0; t < k.n_topics; t++) {
        std::vector<MatchRange> l = rows[t];
        c.pair_cnt[t] = (uint32_t)l.size();
        if (l.size() > 1 && l.size() <= SORT_PAIRS)
            std::stable_sort(l.begin(), l.end(), [&](const MatchRange& x, const MatchRange& y) { return first_id(c, x) < first_id(c, y); });
        bool bad = false;
        uint32_t nr = 0;
        for (size_t i = 0; i < l.size(); i++) {
            if (i && first_id(c, l[i]) <= last_id(c, l[i - 1])) bad = true;
            const uint32_t n = l[i].count & ~RANGE_INDIRECT;
            for (uint32_t j = 0; j < n; j++) c.ids.push_back((l[i].count & RANGE_INDIRECT) ? c.route_pos[l[i].begin + j] : l[i].begin + j);
            nr += n;
        }
        c.route_cnt[t] = nr;
        c.row_ptr[t + 1] = c.row_ptr[t] + nr;
        if (bad && nr > 1) c.bad_rows.insert(t);
    }
    return c;
}

static unsigned long long g_waves = 0;

// runs the kernel over the whole case; capacity < 0: exactly what is needed
static bool run_case(const Case& c, long long capacity, std::mt19937_64& rng, const char* what) {
    const uint32_t tpw = 1u << c.tpw_shift;
    const uint32_t n_blocks = (c.n_topics + tpw - 1) / tpw;
    const uint64_t total = c.row_ptr[c.n_topics];
    const uint64_t cap = capacity < 0 ? total : (uint64_t)capacity;
    std::vector<uint32_t> out_row_ptr(c.n_topics + 1, 0xCCCCCCCCu), out_ids(cap + 8, 0xCCCCCCCCu), sort_list(c.n_topics + 1, 0xCCCCCCCCu);
    std::vector<unsigned long long> wave_sums(n_blocks, 0), super_sums(((n_blocks >> SUPER_SHIFT) + 1) * SUPER_STRIDE, 0);
    std::vector<uint4> blk_stats(n_blocks);
    unsigned long long ev = 0, er = 0, eb = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        unsigned long long s = 0;
        for (uint32_t t = b * tpw; t < std::min((b + 1) * tpw, c.n_topics); t++) s += c.route_cnt[t];
        wave_sums[b] = s;
        super_sums[(size_t)(b >> SUPER_SHIFT) * SUPER_STRIDE] += s;
        blk_stats[b] = make_uint4((uint32_t)(rng() % 1000), (uint32_t)(rng() % 1000), (uint32_t)(rng() % 1000), 0);
        ev += blk_stats[b].x, er += blk_stats[b].y, eb += blk_stats[b].z;
    }
    Counters ctr;
    memset(&ctr, 0, sizeof(ctr));
    unsigned long long out_total = ~0ull;
    BatchArgs a;
    memset(&a, 0, sizeof(a));
    a.ix.route_pos = c.route_pos.data();
    a.n_topics = c.n_topics;
    a.pair_off = const_cast<uint32_t*>(c.pair_off.data());
    a.pair_cnt = const_cast<uint32_t*>(c.pair_cnt.data());
    a.route_cnt = const_cast<uint32_t*>(c.route_cnt.data());
    a.pairs = const_cast<MatchRange*>(c.pairs.data());
    a.super_sums = super_sums.data();
    a.wave_sums = wave_sums.data();
    a.blk_stats = blk_stats.data();
    a.n_blocks = n_blocks;
    a.tpw_shift = c.tpw_shift;
    a.sort_list = sort_list.data();
    a.sort_cap = c.n_topics + 1;
    a.ctr = &ctr;
    a.out_row_ptr = out_row_ptr.data();
    a.out_ids = out_ids.data();
    a.out_capacity = cap;
    a.out_total = &out_total;
    std::vector<uint32_t> order(n_blocks);
    for (uint32_t b = 0; b < n_blocks; b++) order[b] = b;
    std::shuffle(order.begin(), order.end(), rng);
    for (uint32_t b : order) {
        wemu::run_wave(b, [&] { k_expand(a); });
        g_waves++;
    }
    auto fail = [&](const char* msg, long long x = -1, long long y = -1, long long z = -1) {
        fprintf(stderr, "FAIL (%s): %s %lld %lld %lld  [n_topics %u tpw %u total %llu cap %llu]\n", what, msg, x, y, z, c.n_topics, tpw, (unsigned long long)total,
                (unsigned long long)cap);
        return false;
    };
    if (out_total != total || ctr.total_ids != total) return fail("total", (long long)out_total, (long long)total);
    for (uint32_t t = 0; t <= c.n_topics; t++)
        if (out_row_ptr[t] != c.row_ptr[t]) return fail("row_ptr", t, out_row_ptr[t], c.row_ptr[t]);
    if (ctr.n_visit != ev || ctr.n_ranges != er || ctr.topic_bytes != eb) return fail("statistics");
    if (total > cap) {
        if (!(ctr.status & ST_NOSPACE)) return fail("ST_NOSPACE missing");
    } else if (ctr.status) return fail("status", ctr.status);
    // waves that end inside the capacity have written their rows, the others nothing
    for (uint32_t b = 0; b < n_blocks; b++) {
        const uint32_t t0 = b * tpw, t1 = std::min((b + 1) * tpw, c.n_topics);
        const bool written = c.row_ptr[t1] <= cap;
        for (uint64_t i = c.row_ptr[t0]; i < c.row_ptr[t1] && i < cap; i++)
            if (written ? out_ids[i] != c.ids[i] : out_ids[i] != 0xCCCCCCCCu) {
                uint32_t t = t0;
                while (c.row_ptr[t + 1] <= i) t++;
                return fail(written ? "id" : "id written beyond the capacity's last whole wave", (long long)i, out_ids[i], c.ids[i]), fail("  ... in row / np", t,
                                                                                                                                             c.pair_cnt[t]);
            }
    }
    for (uint64_t i = cap; i < cap + 8; i++)
        if (out_ids[i] != 0xCCCCCCCCu) return fail("write behind the buffer", (long long)i);
    std::set<uint32_t> got;
    for (uint32_t i = 0; i < ctr.sort_count; i++) got.insert(sort_list[i]);
    if (got.size() != ctr.sort_count) return fail("sort_list holds a row twice");
    std::set<uint32_t> want;
    for (uint32_t t : c.bad_rows) {
        const uint32_t b = t / tpw, t1 = std::min((b + 1) * tpw, c.n_topics);
        if (c.row_ptr[t1] <= cap) want.insert(t); // (only waves that write check the order)
    }
    if (got != want) {
        for (uint32_t t : want)
            if (!got.count(t)) return fail("row missing in sort_list", t, c.pair_cnt[t]);
        for (uint32_t t : got)
            if (!want.count(t)) return fail("row in sort_list that is in order", t, c.pair_cnt[t]);
    }
    return true;
}

int main(int argc, char** argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 300;
    const unsigned long long seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed);
    int done = 0;
    for (int i = 0; i < cases; i++) {
        Knobs k;
        const int shape = i % 8;
        k.tpw_shift = (i % 5 == 4) ? 4 : (i % 11 == 10 ? 2 : 6);
        k.n_topics = 1 + (uint32_t)(rng() % (shape == 7 ? 20000 : 400));
        k.p_empty_row = shape == 1 ? 0.7 : 0.1;
        k.p_shuffle = shape == 2 ? 0.0 : 0.8;
        k.p_indirect = (shape == 3 || shape == 6) ? 0.3 : (shape == 4 ? 0.0 : 0.03);
        k.p_long = shape == 5 ? 0.4 : 0.03;
        k.p_huge_row = shape == 6 ? 0.2 : 0.01;
        k.p_interleave = shape == 3 ? 0.3 : 0.0;
        k.p_zero_len = shape == 6 ? 0.02 : 0.0;
        k.max_np = shape == 0 ? 12 : 32;
        k.gather = (i % 3) == 2;
        if (shape == 7) k.p_huge_row = 0.0005, k.p_long = 0.002;
        Case c = make_case(rng, k);
        char what[128];
        snprintf(what, sizeof(what), "case %d shape %d gather %d", i, shape, (int)k.gather);
        if (!run_case(c, -1, rng, what)) return 1;
        done++;
        if (i % 4 == 1 && c.row_ptr[c.n_topics] > 10) { // the caller's buffer is too small: rows in front of the overflow are still written
            const long long cap = (long long)(rng() % c.row_ptr[c.n_topics]);
            if (!run_case(c, cap, rng, what)) return 1;
            done++;
        }
        if (i % 4 == 3 && !run_case(c, 0, rng, what)) return 1; // COUNTS / RANGES formats: row pointers only
    }
    printf("ok: %d runs, %llu waves, %llu cross-lane rendezvous\n", done, g_waves, wemu::st().rendezvous);
    return 0;
}
