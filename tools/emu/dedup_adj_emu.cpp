// dedup_adj_emu.cpp -- host-side logic test of k_dd_adj_heads / k_dd_adj_scatter / k_fill_adj (bifromq_amd/csrc/bmq_dedup_adj_kernels.h)
// under the wave64 emulator of wave_emu.h.  Test tooling: the kernels' LOGIC -- run heads across block and super-block borders, the dense
// batch (order, offsets, bytes copied through the LDS image incl. the shared first / last 16-byte chunks, the byte-copy path of blocks
// that do not fit it), the rows behind the last head, a buffer that is too small, 2^tpw_shift rows per wave -- against a plain
// restatement on random batches, ordered and not.  What the GPU makes of the same source is what tests/ (-m gpu) check against the oracle.
//   g++ -O1 -g -std=c++17 -I bifromq_amd/csrc -I tools/emu tools/emu/dedup_adj_emu.cpp -o build/dedup_adj_emu && build/dedup_adj_emu [cases] [seed]
//   -DBMQ_ADJ_IMG=256: a small LDS image, so that the byte-copy path runs all the time
//   add -fsanitize=address,undefined (ASAN_OPTIONS=detect_stack_use_after_return=0: the lanes are ucontext fibers): LDS arrays are function statics here, so a read or
//   write past one is reported -- both harnesses run clean that way (round 5)
#define BMQ_WAVE_EMU 1
#include "wave_emu.h"

#include <random>
#include <string>
#include <vector>

#include "bmq_batch_args.h"
namespace bmq {
inline uint32_t global_word_at(const uint8_t* base, uint32_t p) { // 4 bytes at any alignment (the device reads two aligned words)
    uint32_t w;
    memcpy(&w, base + p, 4);
    return w;
}
inline uint32_t lds_word_at(const uint32_t* words, uint32_t rel) { // (the device: two aligned LDS words + v_alignbyte)
    uint32_t w;
    memcpy(&w, reinterpret_cast<const uint8_t*>(words) + rel, 4);
    return w;
}
} // namespace bmq
#include "bmq_expand_kernel.h" // the cross-lane vocabulary (wave_total_u64, wave_incl_scan, read_lane)
#include "bmq_dedup_adj_kernels.h"

using namespace bmq;

#define FAIL(...)                     \
    do {                              \
        fprintf(stderr, __VA_ARGS__); \
        return 1;                     \
    } while (0)

static int one_case(std::mt19937_64& rng, uint32_t n, uint32_t tpw_shift, bool ordered, uint32_t pool_size, uint32_t max_len, bool short_buffer, uint64_t& n_heads_total) {
    auto rnd = [&](size_t m) { return (size_t)(rng() % m); };
    std::vector<std::string> pool;
    for (uint32_t i = 0; i < pool_size; i++) {
        std::string s;
        const size_t len = rnd(8) == 0 ? rnd(max_len + 1) : rnd(std::min<uint32_t>(max_len, 40) + 1);
        for (size_t k = 0; k < len; k++) s.push_back("ab/c"[rnd(4)]);
        pool.push_back(s);
    }
    std::vector<std::pair<uint32_t, std::string>> rows;
    for (uint32_t i = 0; i < n; i++) {
        const double u = (double)(rng() >> 11) / (double)(1ull << 53);
        const size_t z = std::min<size_t>((size_t)(1.0 / (1e-9 + u)) - 1, pool_size - 1); // Zipf-like repeats
        rows.emplace_back((uint32_t)rnd(3), pool[z]);
    }
    if (ordered) std::sort(rows.begin(), rows.end());
    std::vector<uint8_t> topics;
    std::vector<uint32_t> off(n + 1, 0), tenant(n);
    for (uint32_t i = 0; i < n; i++) {
        tenant[i] = rows[i].first == 2 ? 77u : rows[i].first; // (77: a tenant index the batch's table does not have -- passes through)
        topics.insert(topics.end(), rows[i].second.begin(), rows[i].second.end());
        off[i + 1] = (uint32_t)topics.size();
    }
    topics.resize(topics.size() + 32, 0xEE);
    // expected
    std::vector<uint32_t> x_rep(n), x_heads;
    for (uint32_t i = 0; i < n; i++) {
        const bool head = i == 0 || rows[i] != rows[i - 1];
        x_rep[i] = head ? i : x_rep[i - 1];
        if (head) x_heads.push_back(i);
    }
    n_heads_total += x_heads.size();
    uint32_t x_bytes = 0;
    for (uint32_t h : x_heads) x_bytes += off[h + 1] - off[h];

    const uint32_t nb = (n + (1u << tpw_shift) - 1) >> tpw_shift, n_super = ((nb - 1) >> SUPER_SHIFT) + 1;
    std::vector<uint32_t> drow(n, 0xABABABABu), c_off(n + 1, 0xABABABABu), c_tenant(n, 0xABABABABu), c_rep(n, 0xABABABABu);
    std::vector<unsigned long long> blk_cnt(nb, ~0ull), blk_mask(nb, 0x5555ull), super_cnt((size_t)n_super * SUPER_STRIDE, 0ull);
    const uint64_t cap = short_buffer ? (uint64_t)x_bytes + 63 - rnd(std::min<uint32_t>(x_bytes + 1, 64)) : (uint64_t)x_bytes + 64 + rnd(100);
    std::vector<uint8_t> c_topics((size_t)x_bytes + 256, 0xCD);
    Counters ctr{};
    AdjArgs g{};
    g.topics = topics.data(), g.topic_off = off.data(), g.topic_tenant = tenant.data();
    g.n_topics = n, g.n_blocks = nb, g.tpw_shift = tpw_shift;
    g.drow = drow.data(), g.blk_cnt = blk_cnt.data(), g.blk_mask = blk_mask.data(), g.super_cnt = super_cnt.data();
    g.c_topics = c_topics.data(), g.c_cap = cap, g.c_off = c_off.data(), g.c_tenant = c_tenant.data(), g.c_rep = c_rep.data(), g.ctr = &ctr;
    for (uint32_t b = 0; b < nb; b++) wemu::run_wave(nb - 1 - b, [&] { k_dd_adj_heads(g); });
    // (the blocks in a scrambled order: nothing may depend on which wave runs first; every block once when 7 and nb are coprime -- else the rest
    // of them behind: the kernel is idempotent)
    for (uint32_t b = 0; b < nb; b++) wemu::run_wave((b * 7 + 3) % nb, [&] { k_dd_adj_scatter(g); });
    if (nb % 7 == 0)
        for (uint32_t b = 0; b < nb; b++) wemu::run_wave(b, [&] { k_dd_adj_scatter(g); });
    std::vector<uint32_t> dense(n, 0); // head row -> its dense row
    for (uint32_t d = 0; d < x_heads.size(); d++) dense[x_heads[d]] = d;
    for (uint32_t i = 0; i < n; i++)
        if (drow[i] != dense[x_rep[i]]) FAIL("drow[%u] = %u, expected %u (n %u tpw %u ordered %d)\n", i, drow[i], dense[x_rep[i]], n, 1u << tpw_shift, (int)ordered);
    if (ctr.n_walked != x_heads.size()) FAIL("n_walked %u, expected %zu\n", ctr.n_walked, x_heads.size());
    const bool over = (uint64_t)x_bytes + 64 > cap;
    if (over != ((ctr.status & ST_NEED_ADJ) != 0)) FAIL("ST_NEED_ADJ %u, expected %d\n", ctr.status, (int)over);
    if (over && ctr.adj_bytes != x_bytes) FAIL("adj_bytes %u, expected %u\n", ctr.adj_bytes, x_bytes);
    uint32_t run = 0;
    for (uint32_t d = 0; d < x_heads.size(); d++) {
        const uint32_t h = x_heads[d], len = off[h + 1] - off[h];
        if (c_rep[d] != d) FAIL("c_rep[%u] = %u\n", d, c_rep[d]);
        if (over) {
            if (c_tenant[d] != 0xFFFFFFFFu || c_off[d] != 0) FAIL("over: dense row %u not neutral\n", d);
            continue;
        }
        if (c_tenant[d] != tenant[h] || c_off[d] != run) FAIL("dense row %u: tenant %u off %u, expected %u %u\n", d, c_tenant[d], c_off[d], tenant[h], run);
        if (memcmp(c_topics.data() + run, topics.data() + off[h], len) != 0) FAIL("dense row %u: bytes differ (n %u tpw %u, off %u len %u)\n", d, n, 1u << tpw_shift, run, len);
        run += len;
    }
    for (uint32_t r = (uint32_t)x_heads.size(); r < n; r++)
        if (c_tenant[r] != 0xFFFFFFFFu || c_rep[r] != r || c_off[r + 1] != (over ? 0u : x_bytes)) FAIL("row %u behind the heads: tenant %u rep %u off %u\n", r, c_tenant[r], c_rep[r], c_off[r + 1]);
    if (c_off[x_heads.size()] != (over ? 0u : x_bytes)) FAIL("c_off[n_heads] = %u, expected %u\n", c_off[x_heads.size()], over ? 0u : x_bytes);
    for (size_t i = over ? 0 : x_bytes; i < c_topics.size(); i++)
        if (c_topics[i] != 0xCD) FAIL("byte %zu behind the dense batch was written (bytes %u, over %d)\n", i, x_bytes, (int)over);
    // k_fill_adj: every row takes what the walk left for its head's dense row
    std::vector<uint32_t> cpo(n), cpc(n), crc(n), cvis(n), po(n, 1), pc(n, 1), rc(n, 1);
    for (uint32_t d = 0; d < n; d++) cpo[d] = (uint32_t)rnd(1u << 20), cpc[d] = (uint32_t)rnd(9), crc[d] = (uint32_t)rnd(5000), cvis[d] = (uint32_t)rnd(40);
    std::vector<unsigned long long> wave_sums(nb, ~0ull), super_sums((size_t)n_super * SUPER_STRIDE, 0ull);
    std::vector<uint4> blk_stats(nb);
    BatchArgs a{};
    a.topic_off = off.data(), a.n_topics = n, a.n_blocks = nb, a.tpw_shift = tpw_shift;
    a.pair_off = po.data(), a.pair_cnt = pc.data(), a.route_cnt = rc.data();
    a.wave_sums = wave_sums.data(), a.super_sums = super_sums.data(), a.blk_stats = blk_stats.data();
    AdjFill f{drow.data(), cpo.data(), cpc.data(), crc.data(), cvis.data()};
    for (uint32_t b = 0; b < nb; b++) wemu::run_wave(b, [&] { k_fill_adj(a, f); });
    std::vector<unsigned long long> x_super(n_super, 0);
    for (uint32_t b = 0; b < nb; b++) {
        unsigned long long s = 0, v = 0, p = 0, by = 0;
        for (uint32_t i = b << tpw_shift; i < std::min(n, (b + 1) << tpw_shift); i++) {
            const uint32_t d = dense[x_rep[i]];
            if (po[i] != cpo[d] || pc[i] != cpc[d] || rc[i] != crc[d]) FAIL("k_fill_adj: row %u\n", i);
            s += crc[d], v += cvis[d], p += cpc[d], by += off[i + 1] - off[i];
        }
        if (wave_sums[b] != s || blk_stats[b].x != v || blk_stats[b].y != p || blk_stats[b].z != by) FAIL("k_fill_adj: block %u sums\n", b);
        x_super[b >> SUPER_SHIFT] += s;
    }
    for (uint32_t i = 0; i < n_super; i++)
        if (super_sums[(size_t)i * SUPER_STRIDE] != x_super[i]) FAIL("k_fill_adj: super-block %u\n", i);
    return 0;
}

int main(int argc, char** argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 60;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
    std::mt19937_64 rng(seed);
    uint64_t heads = 0, rows = 0;
    for (int c = 0; c < cases; c++) {
        const uint32_t shifts[3] = {6, 4, 2};
        const uint32_t tpw_shift = shifts[c % 3];
        uint32_t n = 1 + (uint32_t)(rng() % (c % 10 == 0 ? 40000 : 3000)); // (40 000 rows of 64: 625 blocks = three super-blocks)
        if (tpw_shift != 6) n = 1 + n % 900;
        const uint32_t pool = 1 + (uint32_t)(rng() % (c % 4 == 0 ? 8 : 400));
        const uint32_t max_len = c % 5 == 0 ? 700 : 60;
        if (one_case(rng, n, tpw_shift, c % 6 != 5, pool, max_len, c % 7 == 6, heads)) {
            fprintf(stderr, "case %d failed (seed %llu)\n", c, (unsigned long long)seed);
            return 1;
        }
        rows += n;
    }
    printf("dedup_adj emu ok: %d cases, %llu rows, %llu heads, LDS image %u bytes\n", cases, (unsigned long long)rows, (unsigned long long)heads, ADJ_IMG);
    return 0;
}
