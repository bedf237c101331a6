// walk_emu.cpp -- host-side logic test of the dist direction's match kernels -- k_walk<TC, QC, PC, MIXED> (bmq_walk_kernel.h), k_walk_slow
// (bmq_dist_kernels.h) and k_expand behind them -- under the wave64 emulator of wave_emu.h, on indexes the product's own builder makes
// (bmq_dist_index.h through HostExec: the code the builder kernels run).  Test tooling: the kernels' LOGIC -- tokeniser and ragged token
// table, chunked waves, the work stack and the range buffer with their spill chains (smallest LDS lists), tenants of a wave walked one after
// the other, the MIXED instantiation for batches that are not grouped by tenant, '$' topics, empty levels, unknown tenants, topics deeper than
// FAST_LEVELS (k_walk_slow), indexes after mutations (id lists, indirect ranges), and the whole pipeline of an engine with bmq_config.dedup_sorted
// (k_dd_adj_heads -> k_dd_adj_scatter -> the walk kernels on the dense batch -> k_fill_adj -> k_expand, wired as launch_dist wires them) on
// ordered batches full of repeats -- against a brute force over the model's route keys (the rule of SURVEY.md 8a-0).  What the GPU makes of the same source is what tests/ (-m gpu) check against the oracle.
//   g++ -O1 -g -std=c++17 -I bifromq_amd/csrc -I tools/emu tools/emu/walk_emu.cpp bifromq_amd/csrc/bmq_codec.cpp -o build/walk_emu -pthread && build/walk_emu [rounds] [seed]
//   add -fsanitize=address,undefined (ASAN_OPTIONS=detect_stack_use_after_return=0: the lanes are ucontext fibers): LDS arrays are function statics here, so a read or
//   write past one is reported -- both harnesses run clean that way (round 5)
#define BMQ_WAVE_EMU 1
#include "wave_emu.h"

#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "bmq_codec.h"
#include "bmq_dist_index.h"
#include "bmq_exec_host.h"

// the device builtins the kernels' sources spell out
#define __align__(n)
inline uint32_t wemu_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (s & 3u))); }
#define __builtin_amdgcn_alignbyte(hi, lo, s) wemu_alignbyte((hi), (lo), (s))
#define __builtin_amdgcn_readlane(v, l) ((int)bmq::read_lane((uint32_t)(v), (uint32_t)(l)))
#define __builtin_amdgcn_s_getreg(x) 0u
#define __ffs(x) __builtin_ffs(x)
#define __popc(x) __builtin_popcount(x)
namespace bmq {
inline uint32_t lds_word_at(const uint32_t* words, uint32_t rel) { // (bmq_dedup_adj_kernels.h comes along with bmq_dist_kernels.h)
    uint32_t w;
    memcpy(&w, reinterpret_cast<const uint8_t*>(words) + rel, 4);
    return w;
}
} // namespace bmq
#include "bmq_dist_kernels.h"

using namespace bmq;

static std::vector<std::string> split(std::string_view s, char sep) {
    std::vector<std::string> out;
    size_t b = 0;
    for (size_t i = 0; i <= s.size(); i++)
        if (i == s.size() || s[i] == sep) {
            out.emplace_back(s.substr(b, i - b));
            b = i + 1;
        }
    return out;
}
// the rule itself, on level lists
static bool filter_matches(const std::vector<std::string>& f, const std::vector<std::string>& t) {
    for (size_t i = 0; i < f.size(); i++) {
        const bool wild0 = i == 0 && !t.empty() && !t[0].empty() && t[0][0] == '$';
        if (f[i] == "#" && i + 1 == f.size()) return !wild0;
        if (i >= t.size()) return false;
        if (f[i] == "+") {
            if (wild0) return false;
            continue;
        }
        if (f[i] != t[i]) return false;
    }
    return f.size() == t.size();
}

#define FAIL(...)                     \
    do {                              \
        fprintf(stderr, __VA_ARGS__); \
        return 1;                     \
    } while (0)

template <class T> static T* buf(std::vector<uint8_t>& store, size_t n) {
    store.assign(n * sizeof(T) + 64, 0);
    return reinterpret_cast<T*>(store.data());
}

struct Coverage {
    uint64_t rows = 0, ids = 0, batches = 0, mixed = 0, slow_rows = 0, spills = 0, chunked = 0, sorted_rows = 0, after_apply = 0, two_tenant_waves = 0;
    uint64_t adj_batches = 0, adj_rows = 0, adj_walked = 0, adj_slow = 0;
    uint64_t split_blocks = 0, split_overflow = 0, split_adj = 0; // k_expand: heavy blocks expanded by four waves, blocks the list had no room for
};

// one batch through walk (+ slow) + expand; rows -> sorted id lists
template <int TC, int QC, int PC>
static int run_batch(const DistIndexView& ix, const std::vector<std::string>& tnames, const std::vector<uint32_t>& tt, const std::vector<std::string>& topics, uint32_t tpw_shift,
                     std::vector<std::vector<uint32_t>>& rows, Coverage& cov, bool adj = false) {
    const uint32_t n = (uint32_t)topics.size();
    std::vector<uint8_t> tb, pb;
    std::vector<uint32_t> toff{0}, poff{0};
    for (auto& s : tnames) tb.insert(tb.end(), s.begin(), s.end()), toff.push_back((uint32_t)tb.size());
    for (auto& s : topics) pb.insert(pb.end(), s.begin(), s.end()), poff.push_back((uint32_t)pb.size());
    tb.resize(tb.size() + 32, 0), pb.resize(pb.size() + 32, 0);
    // the topic bytes 16-byte aligned, as the ABI asks
    std::vector<uint8_t> pstore(pb.size() + 32);
    uint8_t* pal = pstore.data() + ((16 - ((uintptr_t)pstore.data() & 15)) & 15);
    memcpy(pal, pb.data(), pb.size());
    const uint32_t nb = (n + (1u << tpw_shift) - 1) >> tpw_shift, n_super = ((nb - 1) >> SUPER_SHIFT) + 1;
    std::vector<uint8_t> s_heavy;
    std::vector<uint8_t> s_po, s_pc, s_rc, s_pairs, s_subs, s_super, s_stats, s_spill, s_ws, s_slow, s_scr, s_sort, s_ctr, s_row, s_ids, s_tot;
    std::vector<uint8_t> s_drow, s_mask, s_cnt, s_asup, s_ctop, s_coff, s_cten, s_crep, s_cpo, s_cpc, s_crc, s_vis; // bmq_config.dedup_sorted: the dense batch and its results
    BatchArgs a{};
    a.ix = ix;
    a.tenants = tb.data(), a.tenant_off = toff.data(), a.n_tenants = (uint32_t)tnames.size();
    a.topic_tenant = tt.data(), a.topics = pal, a.topic_off = poff.data(), a.n_topics = n;
    a.pair_cap = 1u << 19, a.spill_cap = 1u << 21, a.slow_cap = n + 8, a.scratch_cap = 1u << 20, a.sort_cap = n + 8;
    a.n_blocks = nb, a.tpw_shift = tpw_shift;
    a.qcap = QC, a.pcap = PC;
    const uint64_t out_cap = 1u << 20;
    bool mixed = false;
    for (int attempt = 0; attempt < 3; attempt++) {
        a.pair_off = buf<uint32_t>(s_po, n), a.pair_cnt = buf<uint32_t>(s_pc, n), a.route_cnt = buf<uint32_t>(s_rc, n);
        a.pairs = buf<MatchRange>(s_pairs, a.pair_cap), a.subs = buf<SubAlloc>(s_subs, 2 * N_SUB + 1);
        a.subs = reinterpret_cast<SubAlloc*>(((uintptr_t)a.subs + 127) & ~(uintptr_t)127);
        a.super_sums = buf<unsigned long long>(s_super, (size_t)n_super * SUPER_STRIDE), a.blk_stats = buf<uint4>(s_stats, nb);
        a.spill = buf<uint4>(s_spill, a.spill_cap), a.wave_sums = buf<unsigned long long>(s_ws, nb);
        a.slow_list = buf<uint32_t>(s_slow, a.slow_cap), a.scratch = buf<uint32_t>(s_scr, a.scratch_cap), a.sort_list = buf<uint32_t>(s_sort, a.sort_cap);
        a.ctr = buf<Counters>(s_ctr, 1);
        a.heavy_list = nullptr, a.heavy_cap = 0;
        if (tpw_shift == 6 && (n & 1u)) a.heavy_cap = 1 + (n >> 1) % 3, a.heavy_list = buf<uint32_t>(s_heavy, a.heavy_cap), a.split_ranges = 24, a.split_ids = 40; // (a list of 1-3 entries: it overflows)
        a.out_row_ptr = buf<uint32_t>(s_row, n + 1), a.out_ids = buf<uint32_t>(s_ids, out_cap), a.out_capacity = out_cap;
        a.out_total = buf<unsigned long long>(s_tot, 1);
        wemu::grid_size() = nb;
        BatchArgs w = a; // what the walk kernels run on
        AdjFill gf{};
        if (adj) { // the wiring of launch_dist (bmq_engine.hip) for an engine with bmq_config.dedup_sorted
            AdjArgs g{};
            g.topics = a.topics, g.topic_off = a.topic_off, g.topic_tenant = a.topic_tenant, g.n_topics = n, g.n_blocks = nb, g.tpw_shift = tpw_shift;
            g.drow = buf<uint32_t>(s_drow, n), g.blk_mask = buf<unsigned long long>(s_mask, nb), g.blk_cnt = buf<unsigned long long>(s_cnt, nb);
            g.super_cnt = buf<unsigned long long>(s_asup, (size_t)n_super * SUPER_STRIDE);
            g.c_cap = pb.size() + 64;
            s_ctop.assign(g.c_cap + 128, 0xCD);
            g.c_topics = s_ctop.data() + ((16 - ((uintptr_t)s_ctop.data() & 15)) & 15);
            g.c_off = buf<uint32_t>(s_coff, n + 1), g.c_tenant = buf<uint32_t>(s_cten, n), g.c_rep = buf<uint32_t>(s_crep, n), g.ctr = a.ctr;
            for (uint32_t b = 0; b < nb; b++) wemu::run_wave(b, [&] { k_dd_adj_heads(g); });
            for (uint32_t b = 0; b < nb; b++) wemu::run_wave(nb - 1 - b, [&] { k_dd_adj_scatter(g); });
            if (a.ctr->status & ST_NEED_ADJ) FAIL("the dense batch did not fit a buffer as large as the batch\n");
            a.rep = g.drow, a.visit_cnt = buf<uint32_t>(s_vis, n);
            w = a;
            w.topics = g.c_topics, w.topic_off = g.c_off, w.topic_tenant = g.c_tenant, w.rep = g.c_rep;
            w.pair_off = buf<uint32_t>(s_cpo, n), w.pair_cnt = buf<uint32_t>(s_cpc, n), w.route_cnt = buf<uint32_t>(s_crc, n);
            gf = AdjFill{g.drow, w.pair_off, w.pair_cnt, w.route_cnt, w.visit_cnt};
        }
        for (uint32_t b = 0; b < nb; b++) {
            if (mixed) wemu::run_wave(b, [&] { k_walk<TC, QC, PC, true>(w); });
            else wemu::run_wave(b, [&] { k_walk<TC, QC, PC, false>(w); });
        }
        if ((a.ctr->status & ST_WANT_MIXED) && !mixed) { // the batch is not grouped by tenant: once more through the instantiation for that (bmq_engine.hip)
            mixed = true;
            cov.mixed++;
            continue;
        }
        if (a.ctr->status & ST_RERUN) FAIL("walk asked for larger buffers: status %u (the harness' are meant to be large enough)\n", a.ctr->status);
        if (a.ctr->slow_count) {
            cov.slow_rows += a.ctr->slow_count;
            if (adj) cov.adj_slow += a.ctr->slow_count;
            wemu::grid_size() = 2;
            for (uint32_t b = 0; b < 2; b++) wemu::run_wave(b, [&] { k_walk_slow(w); });
            if (a.ctr->status & ST_RERUN) FAIL("slow walk asked for larger buffers: status %u\n", a.ctr->status);
        }
        if (adj) {
            wemu::grid_size() = nb;
            for (uint32_t b = 0; b < nb; b++) wemu::run_wave(b, [&] { k_fill_adj(a, gf); });
            cov.adj_batches++, cov.adj_rows += n, cov.adj_walked += a.ctr->n_walked;
        }
        break;
    }
    {
        unsigned long long spilled = 0; // records handed out by the spill area's sub-allocators: a full range buffer was flushed, a full stack parked
        for (uint32_t i = 0; i < N_SUB; i++) spilled += a.subs[N_SUB + i].used;
        if (spilled) cov.spills++;
    }
    { // k_expand's grid as launch_dist lays it out: the helper waves of the heavy blocks in front
        const uint32_t grid = nb + (EXPAND_PARTS - 1) * a.heavy_cap;
        wemu::grid_size() = grid;
        for (uint32_t b = 0; b < grid; b++) wemu::run_wave(b, [&] { k_expand(a); });
        const uint32_t listed = std::min(a.ctr->heavy_count, a.heavy_cap);
        cov.split_blocks += listed, cov.split_overflow += a.ctr->heavy_count - listed;
        if (adj) cov.split_adj += listed;
        for (uint32_t i = 0; i < listed; i++)
            if (a.heavy_list[i] >= nb || a.blk_stats[a.heavy_list[i]].w != 1u) FAIL("heavy list entry %u: block %u is not flagged\n", i, a.heavy_list[i]);
    }
    if (a.ctr->status & (ST_NOSPACE | ST_RANGE | ST_RERUN)) FAIL("expand: status %u\n", a.ctr->status);
    if (*a.out_total != a.out_row_ptr[n]) FAIL("total %llu, row_ptr[n] %u\n", *a.out_total, a.out_row_ptr[n]);
    rows.assign(n, {});
    std::set<uint32_t> to_sort(a.sort_list, a.sort_list + std::min(a.ctr->sort_count, a.sort_cap));
    cov.sorted_rows += to_sort.size();
    for (uint32_t t = 0; t < n; t++) {
        rows[t].assign(a.out_ids + a.out_row_ptr[t], a.out_ids + a.out_row_ptr[t + 1]);
        if (to_sort.count(t)) std::sort(rows[t].begin(), rows[t].end()); // (k_sort_rows' business)
        else if (!std::is_sorted(rows[t].begin(), rows[t].end())) FAIL("row %u is not ascending and not listed for k_sort_rows\n", t);
        cov.ids += rows[t].size();
    }
    cov.rows += n, cov.batches++;
    return 0;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 12;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
    std::mt19937_64 rng(seed);
    auto rnd = [&](size_t n) { return (size_t)(rng() % n); };
    const std::vector<std::string> tenants = {"t", "tenantB", "x", "a-much-longer-tenant-identifier", ""};
    const std::vector<std::string> alpha = {"a", "b", "c", "", "$sys", "+", "a-level-longer-than-sixteen-bytes", "\xE4\xBD\xA0\xE5\xA5\xBD", "0", "exactly-16-bytes", "#"};
    auto rand_filter = [&](size_t max_depth) {
        std::string f;
        const size_t depth = 1 + rnd(max_depth);
        for (size_t i = 0; i < depth; i++) {
            if (i) f += '/';
            if (i + 1 == depth && rnd(5) == 0) f += "#";
            else {
                std::string l = alpha[rnd(alpha.size())];
                if (l == "#") l = "#x";
                f += l;
            }
        }
        return f;
    };
    auto rand_topic = [&](size_t max_depth) {
        std::string t;
        const size_t depth = 1 + rnd(max_depth);
        for (size_t i = 0; i < depth; i++) {
            if (i) t += '/';
            std::string l = alpha[rnd(alpha.size())];
            if (l == "+" || l == "#") l = "zz";
            t += l;
        }
        return t;
    };
    auto rand_key = [&](size_t max_depth) {
        const std::string& tn = tenants[rnd(tenants.size())];
        const uint8_t flag = rnd(10) == 0 ? 2 : 1;
        return encode_route_key(tn, rand_filter(max_depth), flag,
                                flag == 1 ? std::to_string(rnd(3)) + std::string("\0", 1) + "inbox" + std::to_string(rnd(40)) + std::string("\0d", 2) + std::to_string(rnd(12))
                                          : "g" + std::to_string(rnd(3)));
    };
    Coverage cov;
    for (int round = 0; round < rounds; round++) {
        HostExec hx;
        hx.threads = 2;
        DistIndex<HostExec> h(hx);
        h.tiny = true;
        std::map<std::string, uint32_t> model;
        const size_t max_depth = round % 4 == 3 ? 22 : 5; // (every fourth round: filters and topics deeper than FAST_LEVELS)
        {
            std::set<std::string> ks;
            const size_t nk = 1 + rnd(round % 3 == 0 ? 4000 : 600);
            for (size_t i = 0; i < nk; i++) ks.insert(rand_key(max_depth));
            if (round % 4 == 1) // a family that branches at every level (tenant "x"): every mix of literal and '+' over five levels, and '#' behind every prefix of those --
                                // a wave of such topics holds hundreds of pending items (both stacks of the work list are parked) and emits more than 64 ranges in one sink
                for (uint32_t m = 0; m < 32; m++)
                    for (uint32_t d = 1; d <= 5; d++) {
                        std::string f;
                        for (uint32_t l = 0; l < d; l++) f += (l ? "/" : "") + (((m >> l) & 1u) ? std::string("+") : "w" + std::to_string(l));
                        if (d < 5 && (m >> d) != 0) continue;
                        ks.insert(encode_route_key("x", f, 1, std::string("0\0wide\0d", 8) + std::to_string(m))); // (below depth 5: a node with routes of its own AND '#' routes)
                        if (d < 5) ks.insert(encode_route_key("x", f + "/#", 1, std::string("0\0wide#\0d", 9) + std::to_string(m)));
                    }
            for (size_t i = 0; i < 40; i++) ks.insert(encode_route_key("t", "a/b", 1, "0" + std::string("\0", 1) + "fan" + std::to_string(i) + std::string("\0d", 2))); // one filter, many receivers
            std::vector<uint8_t> bytes;
            std::vector<uint32_t> off{0};
            uint32_t r = 0;
            for (auto& k : ks) {
                model[k] = r++;
                bytes.insert(bytes.end(), k.begin(), k.end());
                off.push_back((uint32_t)bytes.size());
            }
            bytes.resize(bytes.size() + 16, 0);
            if (!h.rebuild(bytes.data(), off.data(), (uint32_t)ks.size())) FAIL("rebuild: %s\n", h.error.c_str());
        }
        uint32_t next_id = (uint32_t)model.size();
        for (int phase = 0; phase < 2; phase++) {
            if (phase == 1) { // the same index after a batch of mutations: ids out of key order, id lists, dead ids
                std::vector<std::string> keys;
                std::vector<uint8_t> ops;
                const size_t nm = 1 + rnd(400);
                for (size_t i = 0; i < nm; i++) {
                    if (!model.empty() && rnd(2)) {
                        auto it = model.begin();
                        std::advance(it, rnd(std::min<size_t>(model.size(), 300)));
                        keys.push_back(it->first), ops.push_back(1);
                    } else keys.push_back(rand_key(max_depth)), ops.push_back(0);
                }
                uint32_t put_no = 0;
                for (size_t i = 0; i < keys.size(); i++) {
                    if (ops[i]) model.erase(keys[i]);
                    else {
                        if (!model.count(keys[i])) model[keys[i]] = next_id + put_no;
                        put_no++;
                    }
                }
                next_id += put_no;
                std::vector<uint8_t> bytes;
                std::vector<uint32_t> off{0};
                for (auto& k : keys) bytes.insert(bytes.end(), k.begin(), k.end()), off.push_back((uint32_t)bytes.size());
                bytes.resize(bytes.size() + 16, 0);
                if (!h.apply(bytes.data(), off.data(), ops.data(), (uint32_t)keys.size())) FAIL("apply: %s\n", h.error.c_str());
                cov.after_apply++;
            }
            // the model per tenant: (filter levels, id)
            std::map<std::string, std::vector<std::pair<std::vector<std::string>, uint32_t>>> by_tenant;
            for (auto& e : model) {
                RouteKeyParts kp;
                if (!decode_route_key(e.first, kp)) FAIL("model key does not decode\n");
                by_tenant[std::string(kp.tenant)].emplace_back(split(kp.esc_filter, '\0'), e.second);
            }
            for (int bt = 0; bt < 4; bt++) {
                std::vector<std::string> tnames = tenants;
                tnames.push_back("ghost"); // a tenant the index does not know
                const uint32_t shifts[3] = {6, 4, 2};
                const uint32_t tpw_shift = shifts[(round + bt) % 3];
                const uint32_t n0 = 1 + (uint32_t)rnd(tpw_shift == 6 ? 330 : 60);
                std::vector<std::pair<uint32_t, std::string>> rowsrc;
                for (uint32_t i = 0; i < n0; i++) {
                    if (round % 4 == 1 && rnd(3) != 0) { // the branching family's topics: the full path, a prefix of it, one level off
                        std::string t;
                        const uint32_t d = rnd(3) == 0 ? 1 + (uint32_t)rnd(5) : 5, off = rnd(5) == 0 ? (uint32_t)rnd(5) : 99;
                        for (uint32_t l = 0; l < d; l++) t += (l ? "/" : "") + (l == off ? std::string("zz") : "w" + std::to_string(l));
                        rowsrc.emplace_back(2u /* "x" */, t);
                    } else rowsrc.emplace_back((uint32_t)rnd(tnames.size()), rnd(12) == 0 ? std::string() : rand_topic(max_depth));
                }
                const bool grouped = bt != 2; // the third batch of a phase arrives in any order: waves hold many tenants each (MIXED)
                const bool ordered = bt == 3;  // the fourth: ordered by (tenant, topic), every row up to four times -- through the neighbour-compare kernels (dedup_sorted)
                if (ordered) {
                    const size_t m = rowsrc.size();
                    for (size_t i = 0; i < m; i++)
                        for (size_t c = rnd(4); c > 0; c--) rowsrc.push_back(rowsrc[i]);
                    std::sort(rowsrc.begin(), rowsrc.end());
                } else if (grouped) std::stable_sort(rowsrc.begin(), rowsrc.end(), [](auto& x, auto& y) { return x.first < y.first; });
                std::vector<uint32_t> tt;
                std::vector<std::string> topics;
                for (auto& r : rowsrc) tt.push_back(r.first), topics.push_back(r.second);
                std::vector<std::vector<uint32_t>> got;
                const bool small_lists = (round + bt) % 2 == 1; // the smallest LDS lists: stack and range buffer spill all the time
                const DistIndexView ix = h.view();
                const uint32_t n = (uint32_t)topics.size(); // (the ordered batch grew)
                const int rc = small_lists ? run_batch<BMQ_WALK_GEOM_SMALLEST>(ix, tnames, tt, topics, tpw_shift, got, cov, ordered) : run_batch<BMQ_WALK_GEOM_DEFAULT>(ix, tnames, tt, topics, tpw_shift, got, cov, ordered);
                if (rc) FAIL("round %d phase %d batch %d (n %u, tpw %u, %s, %s lists) failed (seed %llu)\n", round, phase, bt, n, 1u << tpw_shift, ordered ? "ordered + dedup_sorted" : grouped ? "grouped" : "any order",
                             small_lists ? "smallest" : "default", (unsigned long long)seed);
                for (uint32_t i = 0; i < n; i++) {
                    std::vector<uint32_t> want;
                    auto it = by_tenant.find(tnames[tt[i]]);
                    if (it != by_tenant.end()) {
                        const auto tl = split(topics[i], '/');
                        for (auto& fe : it->second)
                            if (filter_matches(fe.first, tl)) want.push_back(fe.second);
                    }
                    std::sort(want.begin(), want.end());
                    if (got[i] != want)
                        FAIL("round %d phase %d batch %d row %u: tenant '%s' topic '%s': kernels give %zu ids, the rule %zu (n %u, tpw %u, %s, %s lists; seed %llu)\n", round, phase, bt, i,
                             tnames[tt[i]].c_str(), topics[i].c_str(), got[i].size(), want.size(), n, 1u << tpw_shift, ordered ? "ordered + dedup_sorted" : grouped ? "grouped" : "any order", small_lists ? "smallest" : "default",
                             (unsigned long long)seed);
                }
            }
        }
    }
    printf("walk emu ok: %d rounds, %llu batches (%llu through the MIXED instantiation, %llu on an index after mutations), %llu rows, %llu ids, %llu rows through k_walk_slow, "
           "%llu batches with spill chains, %llu rows left to k_sort_rows; %llu ordered batches through the neighbour-compare kernels (%llu rows, %llu walked, %llu of those by k_walk_slow)\n",
           rounds, (unsigned long long)cov.batches, (unsigned long long)cov.mixed, (unsigned long long)cov.after_apply * 4, (unsigned long long)cov.rows, (unsigned long long)cov.ids,
           (unsigned long long)cov.slow_rows, (unsigned long long)cov.spills, (unsigned long long)cov.sorted_rows, (unsigned long long)cov.adj_batches, (unsigned long long)cov.adj_rows,
           (unsigned long long)cov.adj_walked, (unsigned long long)cov.adj_slow);
    printf("k_expand splitting: %llu heavy blocks expanded by four waves (%llu of them behind k_fill_adj), %llu more the list had no room for\n", (unsigned long long)cov.split_blocks,
           (unsigned long long)cov.split_adj, (unsigned long long)cov.split_overflow);
    if (rounds >= 8 && (!cov.split_blocks || !cov.split_overflow || !cov.split_adj)) FAIL("the cases missed k_expand's split blocks: %llu listed, %llu overflowed, %llu in ordered batches\n",
                                                                                         (unsigned long long)cov.split_blocks, (unsigned long long)cov.split_overflow, (unsigned long long)cov.split_adj);
    printf("k_walk's work stack and range buffer: the range buffer flushed %llu times, the stack parked %llu times, %llu chunks taken back\n", walk_cov.flushes, walk_cov.parks, walk_cov.restores);
    if (rounds >= 8 && (walk_cov.flushes < 20 || walk_cov.parks < 20 || walk_cov.restores < 20)) FAIL("the cases hardly touched the cold paths of k_walk's lists\n");
    if (rounds >= 8 && (!cov.mixed || !cov.slow_rows || !cov.spills || !cov.adj_slow || cov.adj_walked >= cov.adj_rows)) FAIL("the cases missed a path: mixed %llu slow %llu spills %llu\n", (unsigned long long)cov.mixed, (unsigned long long)cov.slow_rows, (unsigned long long)cov.spills);
    return 0;
}
