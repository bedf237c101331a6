// rwalk_emu.cpp -- host-side logic test of k_retain_walk (bifromq_amd/csrc/bmq_rwalk_kernel.h) under the wave64 emulator of wave_emu.h.
// Test tooling: the kernel's LOGIC -- tokeniser, the hand-out of lanes to units, range / list frontiers, the '$' hole of a first-level
// wildcard, '+' over lists, the emission order, room reservation in `pairs`, lists that outgrow LDS and the arena, filters too deep for
// it -- against a brute force over the topic strings (the rule of SURVEY.md 8a-0: RS/index/RetainTopicIndex.java:36-124), on indexes the
// product's own host builder makes (bmq_retain.cpp).  What the GPU makes of the same source is what tests/ (-m gpu) check against the oracle.
//   g++ -O1 -g -std=c++17 -I bifromq_amd/csrc -I tools/emu tools/emu/rwalk_emu.cpp bifromq_amd/csrc/bmq_retain.cpp -o build/rwalk_emu && build/rwalk_emu [rounds] [seed]
#define BMQ_WAVE_EMU 1
#include "wave_emu.h"

#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "bmq_batch_args.h"
#include "bmq_expand_kernel.h" // wave_total_u64 and the cross-lane vocabulary
#include "bmq_retain.h"
#include "bmq_retain_core.h"
#include "bmq_retain_args.h"
#include "bmq_rwalk_kernel.h"

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
// a wildcard in the first level never matches a '$' topic; '#' also matches the level it hangs off
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

template <int G> static int run(int rounds, uint64_t seed) {
    std::mt19937_64 rng(seed);
    auto rnd = [&](size_t n) { return (size_t)(rng() % n); };
    const std::vector<std::string> tenants = {"t", "tenantB", "a-much-longer-tenant-identifier"};
    uint64_t n_filters_checked = 0, n_ids = 0, n_reruns = 0, n_deep = 0;
    for (int round = 0; round < rounds; round++) {
        // ---- an index: `wide` makes nodes with hundreds of children (lists beyond the LDS part, ranges across rounds) ----
        const bool wide = round % 3 == 1;
        std::vector<std::string> alpha = {"a", "b", "c", "", "$sys", "$x", "a-level-longer-than-sixteen-bytes", "\xE4\xBD\xA0\xE5\xA5\xBD", "0", "!"};
        if (wide)
            for (int i = 0; i < 300; i++) alpha.push_back("w" + std::to_string(i));
        auto rand_topic = [&]() {
            std::string t;
            const size_t depth = 1 + rnd(wide ? 4 : 6);
            for (size_t i = 0; i < depth; i++) t += (i ? "/" : "") + alpha[rnd(rnd(3) ? std::min<size_t>(alpha.size(), 10) : alpha.size())];
            return t;
        };
        std::map<std::pair<std::string, std::string>, uint32_t> ids; // (tenant, topic) -> id
        std::vector<RetainIndexHost::Item> items;
        const size_t n_topics = round == 0 ? 0 : 1 + rnd(wide ? 6000 : 1500);
        const size_t n_ten = 1 + rnd(tenants.size());
        for (size_t i = 0; i < n_topics; i++) {
            RetainIndexHost::Item it;
            it.tenant = tenants[rnd(n_ten)];
            it.topic = rand_topic();
            if (wide && rnd(4) == 0) it.topic = "a/" + alpha[10 + rnd(300)] + "/" + alpha[rnd(4)] + (rnd(2) ? "/b" : "");
            items.push_back(std::move(it));
        }
        RetainIndexHost h;
        if (!h.rebuild(std::move(items))) FAIL("round %d: rebuild failed: %s\n", round, h.error.c_str());
        std::map<std::string, std::vector<std::pair<std::vector<std::string>, uint32_t>>> by_tenant;
        for (uint32_t id = 0; id < h.n_topics; id++) {
            std::string_view tn, tp;
            if (!h.topic(id, tn, tp)) FAIL("round %d: topic(%u)\n", round, id);
            by_tenant[std::string(tn)].push_back({split(tp, '/'), id});
        }
        RetainIndexView v{};
        v.nodes = h.nodes.data();
        v.edges = h.edges.data();
        v.posts = h.posts.data();
        v.gps = h.gps.data();
        v.tenants = h.tenants.data();
        v.tenant_mask = (uint32_t)h.tenants.size() - 1;
        v.dict = h.dict.data();
        v.dict_group_mask = (uint32_t)h.dict.size() / DICT_GROUP - 1;
        v.pool = h.pool.data();
        // ---- a batch of filters ----
        const uint32_t n = round == 0 ? 5 : 1 + (uint32_t)rnd(90);
        std::vector<std::string> flt(n);
        std::vector<uint32_t> ft(n);
        std::string tbytes, fbytes;
        std::vector<uint32_t> toff{0}, foff{0};
        std::vector<std::string> tn_list;
        for (size_t t = 0; t < tenants.size(); t++) { // one more than the index may hold: unknown tenants
            tn_list.push_back(tenants[t]);
            tbytes += tenants[t];
            toff.push_back((uint32_t)tbytes.size());
        }
        tn_list.push_back("nobody");
        tbytes += "nobody";
        toff.push_back((uint32_t)tbytes.size());
        const bool one_tenant = rnd(2);
        for (uint32_t i = 0; i < n; i++) {
            std::string f;
            const size_t kind = rnd(12);
            if (kind == 0) f = "#";
            else if (kind == 1) f = rnd(2) ? "+" : "+/#";
            else if (kind == 2) f = rnd(2) ? "+/+" : "+/+/#";
            else if (kind == 3 && wide) f = std::string("a/+/") + alpha[rnd(4)] + (rnd(2) ? "/b" : (rnd(2) ? "/#" : ""));
            else if (kind == 4 && wide) f = rnd(2) ? "a/+/+" : "+/+/+/b";
            else if (kind == 5) { // deeper than the kernel walks: listed, left empty
                const size_t depth = 17 + rnd(4);
                for (size_t k = 0; k < depth; k++) f += (k ? "/" : "") + std::string("a");
            } else if (kind == 6) { // 9 .. 16 levels: a second tokeniser pass
                const size_t depth = 9 + rnd(8);
                for (size_t k = 0; k < depth; k++) f += (k ? "/" : "") + (rnd(3) ? std::string("a") : std::string("+"));
            } else {
                const size_t depth = 1 + rnd(6);
                for (size_t k = 0; k < depth; k++) {
                    if (k) f += '/';
                    if (k + 1 == depth && rnd(3) == 0) f += "#";
                    else if (rnd(3) == 0) f += "+";
                    else f += rnd(12) ? alpha[rnd(std::min<size_t>(alpha.size(), 10))] : std::string("not-in-the-dictionary");
                }
            }
            flt[i] = f;
            ft[i] = one_tenant ? 0u : (uint32_t)rnd(tn_list.size());
            fbytes += f;
            foff.push_back((uint32_t)fbytes.size());
        }
        while (fbytes.size() % 16) fbytes.push_back('\0');
        fbytes.append(16, '\0');
        tbytes.append(32, '\0');
        // ---- the kernel, tiny capacities first (every growth path), then the sizes the host would grow to ----
        const uint32_t grid = 1 + (uint32_t)rnd(3);
        wemu::grid_size() = grid;
        uint32_t rw_cap = rnd(2) ? 4 : 64;
        unsigned long long pair_cap = N_SUB * (rnd(2) ? 8ull : 256ull);
        for (int attempt = 0;; attempt++) {
            if (attempt > 24) FAIL("round %d: growth does not converge (rw_cap %u, pair_cap %llu, %u filters, grid %u)\n", round, rw_cap, pair_cap, n, grid);
            std::vector<uint32_t> pair_off(n, 0xDEAD), pair_cnt(n, 0xDEAD), route_cnt(n, 0xDEAD), deep_list(n, 0xDEAD);
            std::vector<MatchRange> pairs(pair_cap);
            std::vector<SubAlloc> subs(2 * N_SUB);
            for (auto& s : subs) s.used = 0;
            std::vector<unsigned long long> super((n / 64 / 256 + 2) * SUPER_STRIDE, 0), wsum(n / 64 + 1, 0);
            std::vector<uint32_t> arena((size_t)grid * G * 2 * rw_cap + 4, 0xABABABABu);
            Counters ctr{};
            RetainArgs r{};
            r.ix = v;
            r.tenants = (const uint8_t*)tbytes.data();
            r.tenant_off = toff.data();
            r.n_tenants = (uint32_t)tn_list.size();
            r.filter_tenant = ft.data();
            r.filters = (const uint8_t*)fbytes.data();
            r.filter_off = foff.data();
            r.n_filters = n;
            r.deep_list = deep_list.data();
            r.rw_arena = arena.data();
            r.rw_cap = rw_cap;
            BatchArgs a{};
            a.n_topics = n;
            a.tpw_shift = 6;
            a.n_blocks = (n + 63) / 64;
            a.pair_off = pair_off.data();
            a.pair_cnt = pair_cnt.data();
            a.route_cnt = route_cnt.data();
            a.pairs = pairs.data();
            a.pair_cap = pair_cap;
            a.subs = subs.data();
            a.super_sums = super.data();
            a.wave_sums = wsum.data();
            a.ctr = &ctr;
            for (uint32_t blk = 0; blk < grid; blk++)
                wemu::run_wave(blk, [&]() {
                    static RwLds<G> L;
                    retain_walk_rounds<G, false>(r, a, L);
                });
            if (ctr.status & ~(ST_NEED_PAIRS | ST_RETAIN_LIST)) FAIL("round %d: status %#x\n", round, ctr.status);
            if (ctr.status) {
                if (getenv("RWALK_EMU_VERBOSE")) fprintf(stderr, "round %d attempt %d: status %#x rw_cap %u pair_cap %llu\n", round, attempt, ctr.status, rw_cap, pair_cap);
                if (ctr.status & ST_NEED_PAIRS) pair_cap *= 2;
                if (ctr.status & ST_RETAIN_LIST) rw_cap *= 4;
                n_reruns++;
                continue;
            }
            // ---- against the brute force ----
            std::set<uint32_t> deep(deep_list.begin(), deep_list.begin() + ctr.slow_count);
            unsigned long long total = 0;
            for (uint32_t i = 0; i < n; i++) {
                const auto fl = split(flt[i], '/');
                std::vector<uint32_t> exp;
                const bool is_deep = fl.size() > RW_LV;
                if (is_deep != (deep.count(i) != 0)) FAIL("round %d filter %u '%s': deep listing\n", round, i, flt[i].c_str());
                if (!is_deep) {
                    auto it = by_tenant.find(tn_list[ft[i]]);
                    if (it != by_tenant.end())
                        for (auto& tp : it->second)
                            if (filter_matches(fl, tp.first)) exp.push_back(tp.second);
                } else n_deep++;
                std::vector<uint32_t> got;
                if (pair_cnt[i] > pair_cap || (pair_cnt[i] && pair_off[i] + (unsigned long long)pair_cnt[i] > pair_cap)) FAIL("round %d filter %u: range list out of bounds\n", round, i);
                for (uint32_t k = 0; k < pair_cnt[i]; k++) {
                    const MatchRange m = pairs[pair_off[i] + k];
                    if (m.count == 0 || m.count > h.n_topics) FAIL("round %d filter %u '%s': range %u has count %u\n", round, i, flt[i].c_str(), k, m.count);
                    for (uint32_t j = 0; j < m.count; j++) got.push_back(m.begin + j);
                }
                if (got != exp) { // (as emitted: ascending order is part of the contract)
                    fprintf(stderr, "round %d (G %d, grid %u, rw_cap %u) filter %u '%s' of tenant '%s': %zu ids, expected %zu\n", round, G, grid, rw_cap, i, flt[i].c_str(),
                            tn_list[ft[i]].c_str(), got.size(), exp.size());
                    for (size_t k = 0; k < std::min<size_t>(got.size(), 12); k++) fprintf(stderr, " %u", got[k]);
                    fprintf(stderr, " | expected");
                    for (size_t k = 0; k < std::min<size_t>(exp.size(), 12); k++) fprintf(stderr, " %u", exp[k]);
                    fprintf(stderr, "\n");
                    return 1;
                }
                if (route_cnt[i] != exp.size()) FAIL("round %d filter %u: route_cnt %u != %zu\n", round, i, route_cnt[i], exp.size());
                total += exp.size();
                n_ids += exp.size();
                n_filters_checked++;
            }
            (void)total;
            if (n && ctr.topic_bytes == 0 && fbytes.size() > 32 && deep.size() != n) FAIL("round %d: no filter bytes counted\n", round);
            break;
        }
    }
    static const char* const paths[8] = {"postings look-ups", "slices copied", "slices emitted", "merged subtrees", "bulk chunks", "arena reads", "overflow probes", "'+' over a list"};
    for (int i = 0; i < 8; i++) {
        if (bmq::g_rw_cover[i] == 0 && rounds >= 20) FAIL("G %d: no case reached the path '%s'\n", G, paths[i]);
        printf("  %s: %llu\n", paths[i], bmq::g_rw_cover[i]);
        bmq::g_rw_cover[i] = 0;
    }
    printf("ok: G %d: %llu filters (%llu deeper than the kernel walks), %llu ids, %llu re-runs after growth\n", G, (unsigned long long)n_filters_checked,
           (unsigned long long)n_deep, (unsigned long long)n_ids, (unsigned long long)n_reruns);
    return 0;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 30;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    if (run<8>(rounds, seed)) return 1;
    if (run<2>(rounds / 2 + 1, seed + 1)) return 1;
    if (run<4>(rounds / 2 + 1, seed + 2)) return 1;
    return 0;
}
