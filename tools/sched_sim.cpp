// tools/sched_sim.cpp -- planning tool (CPU only): in which ORDER should a k_walk wave take the items of its work list?  Builds the route index of a few
// tenants of bench.py's C3 population with the product's builder on the host executor, forms the waves k_walk forms (64 consecutive publishes of a
// tenant-grouped batch), replays the drain of bmq_walk_kernel.h round by round (boot resolves the root, P0 and PP0; a round takes up to 64 items; a found
// node's '+' child beside it is resolved in the same round; literal probes are pushed in front of '+' probes) and counts ROUNDS per wave under different
// policies of choosing the <= 64 items of a round:
//   lifo      the newest items (the kernel up to round 6: depth first)
//   fifo      the oldest items (breadth first; the list grows)
//   fifo<cap> the oldest items while the list holds <= cap items, the newest otherwise
//   rem       the items with the most topic levels still below them (critical path first; an upper bound of what an order can do)
//     g++ -O2 -std=c++17 -pthread -I bifromq_amd/csrc tools/sched_sim.cpp bifromq_amd/csrc/bmq_gen.cpp bifromq_amd/csrc/bmq_codec.cpp -o /tmp/sched_sim && /tmp/sched_sim [tenants=32] [topics=200000]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <string>
#include <vector>

#include "bmq_dist_index.h"
#include "bmq_exec_host.h"

extern "C" {
void* bmqgen_create(uint64_t seed, uint32_t tenant_base, uint32_t n_tenants, uint32_t routes_per_tenant, int mode);
uint32_t bmqgen_n_keys(void* h);
const uint8_t* bmqgen_key_bytes(void* h);
const uint32_t* bmqgen_key_off(void* h);
const uint8_t* bmqgen_tenant_bytes(void* h);
const uint32_t* bmqgen_tenant_off(void* h);
uint32_t bmqgen_topics(void* h, uint64_t seed, uint32_t n_topics, uint32_t tenant_lo, uint32_t tenant_hi, uint32_t hit_permille, int grouped);
const uint8_t* bmqgen_topic_bytes(void* h);
const uint32_t* bmqgen_topic_off(void* h);
const uint32_t* bmqgen_topic_tenant(void* h);
}
using namespace bmq;

struct Item {
    uint32_t node, level, topic; // probe for the child of `node` that consumes topic level `level`
    bool is_plus;
    uint64_t pslot; // slot of `node` in the tenant's region (~0: the root)
    uint32_t rem;   // topic levels behind `level`
};
constexpr uint64_t AT_ROOT = ~0ull;

int main(int argc, char** argv) {
    const uint32_t n_ten = argc > 1 ? (uint32_t)atoi(argv[1]) : 32, n_topics = argc > 2 ? (uint32_t)atoi(argv[2]) : 200000;
    void* g = bmqgen_create(0xB1F20003ull, 0, n_ten, 10000, 1 /* MODE_MIXED */);
    HostExec x;
    x.threads = 8;
    DistIndex<HostExec> h(x);
    if (!h.rebuild(bmqgen_key_bytes(g), bmqgen_key_off(g), bmqgen_n_keys(g))) {
        fprintf(stderr, "rebuild: %s\n", h.error.c_str());
        return 1;
    }
    const uint32_t n = bmqgen_topics(g, 11, n_topics, 0, n_ten, 900, 1);
    const uint8_t* tb = bmqgen_topic_bytes(g);
    const uint32_t* to = bmqgen_topic_off(g);
    const uint32_t* tt = bmqgen_topic_tenant(g);
    const uint8_t* nb = bmqgen_tenant_bytes(g);
    const uint32_t* no = bmqgen_tenant_off(g);
    const DistIndexMut ix = h.mut();

    // tokens of every topic
    std::vector<std::vector<uint32_t>> toks(n);
    std::vector<uint32_t> dslot(n);
    for (uint32_t i = 0; i < n; i++) {
        dslot[i] = tenant_find(ix.tenants, ix.tenant_mask, ix.tenant_names, nb, no[tt[i]], no[tt[i] + 1]);
        unsigned long long pos = to[i];
        const unsigned long long end = to[i + 1];
        for (;;) {
            LevelHash lh;
            uint32_t inl[4], len;
            const unsigned long long start = pos;
            scan_level_bytes<0x2F2F2F2Fu>(tb, pos, end, lh, inl, len);
            toks[i].push_back(dict_intern(ix, lh, len, inl, tb, start, false));
            if (pos >= end) break;
            pos++;
        }
    }
    struct Policy {
        const char* name;
        int kind; // 0 lifo, 1 fifo(cap), 2 rem
        uint32_t cap;
        uint32_t t2 = 99, valve = 1u << 30; // kind 10: classes rem >= cap | rem >= t2 ... ; above `valve` items the LOWEST class is taken first
        bool rev_boot = false;
    };
    const Policy pols[] = {{"lifo", 0, 0}, {"fifo", 1, 1u << 30}, {"fifo<=176", 1, 176}, {"fifo<=128", 1, 128}, {"fifo<=96", 1, 96}, {"rem", 2, 0}, {"lifo-by-rem-in-round", 3, 0}, {"lifo boot210", 4, 0}, {"lifo boot210 sink21", 5, 0}, {"lifo boot210 sink21 plus-below-lit", 6, 0}, {"lifo boot012 plus-below-lit", 7, 0}, {"two stacks rem>=2", 8, 2}, {"two stacks rem>=3", 8, 3}, {"two stacks rem>=4", 8, 4}, {"boot210 + two stacks rem>=3", 9, 3},
        {"2 stacks >=3 valve 112", 10, 3, 0, 112}, {"2 stacks >=3 valve 96", 10, 3, 0, 96}, {"2 stacks >=3 valve 128", 10, 3, 0, 128},
        {"2 stacks >=3 valve 112 boot210", 10, 3, 0, 112, true}, {"2 stacks >=3 valve 96 boot210", 10, 3, 0, 96, true},
        {"3 stacks >=5,>=3", 10, 5, 3}, {"3 stacks >=4,>=2", 10, 4, 2}, {"3 stacks >=5,>=3 valve 112", 10, 5, 3, 112}, {"3 stacks >=4,>=2 valve 112", 10, 4, 2, 112},
        {"2 stacks >=2 valve 112", 10, 2, 0, 112}, {"2 stacks >=4 valve 112", 10, 4, 0, 112},
        {"2 stacks >=3 QC176 park", 11, 3, 176}, {"2 stacks >=3 QC200 park", 11, 3, 200}, {"2 stacks >=3 QC224 park", 11, 3, 224}, {"2 stacks >=3 QC240 park", 11, 3, 240},
        {"2 stacks >=3 QC176 park boot210", 11, 3, 176, 0, true}, {"2 stacks >=3 QC224 park boot210", 11, 3, 224, 0, true}, {"lifo QC176 park", 11, 99, 176}, {"2 stacks >=2 QC176 park", 11, 2, 176}, {"2 stacks >=4 QC176 park", 11, 4, 176},
        {"HI 128 + shared 200", 11, 3, 200, 128}, {"HI 96 + shared 232", 11, 3, 232, 96}, {"HI 112 + shared 216", 11, 3, 216, 112}, {"HI 128 + shared 200 R2", 11, 2, 200, 128}, {"HI 128 + shared 200 R4", 11, 4, 200, 128}};
    for (const Policy& pol : pols) {
        uint64_t waves = 0, rounds = 0, items = 0, maxlist = 0, over176 = 0, ovf_waves = 0, parks = 0, hi_parks = 0, flushes = 0, maxhi = 0, maxlo = 0, maxp = 0, ranges = 0;
        std::vector<uint64_t> fill(65, 0);
        std::vector<uint32_t> rounds_of;
        for (uint32_t w0 = 0; w0 < n; w0 += 64) {
            const uint32_t w1 = std::min(n, w0 + 64);
            std::vector<Item> L; // the work list, oldest first
            auto push_children = [&](std::vector<Item>& lit, std::vector<Item>& plus, uint32_t topic, uint32_t node, uint64_t slot, uint32_t dl, uint32_t bloom, bool sys0, bool plus_resolved) {
                const auto& tk = toks[topic];
                if (dl >= tk.size()) return;
                const uint32_t t = tk[dl];
                if (t != TOK_UNKNOWN && ((bloom >> bloom_bit(t)) & 1u)) lit.push_back({node, dl, topic, false, slot, (uint32_t)tk.size() - 1 - dl});
                if ((bloom & BLOOM_PLUS) && !sys0 && !plus_resolved) plus.push_back({node, dl, topic, true, slot, (uint32_t)tk.size() - 1 - dl});
            };
            // boot: up to three sinks (root, P0, PP0)
            {
                std::vector<Item> lit[3], plus[3];
                for (uint32_t i = w0; i < w1; i++) {
                    if (dslot[i] == NONE) continue;
                    const TenantSlot& rg = h.dir[dslot[i]];
                    const bool sys = to[i + 1] > to[i] && tb[to[i]] == '$';
                    const bool has_p0 = (rg.root_lit_bloom & BLOOM_PLUS) && rg.root_plus != NONE;
                    push_children(lit[0], plus[0], i, 0, AT_ROOT, 0, rg.root_lit_bloom, sys, has_p0);
                    if (!has_p0 || sys || toks[i].size() < 1) continue;
                    const TrieSlot& p0 = h.trie[rg.base + rg.root_plus];
                    const TrieSlot& o = h.trie[rg.base + (rg.root_plus ^ 1u)];
                    const bool has_pp0 = o.parent == p0.node && o.token == TOK_PLUS;
                    push_children(lit[1], plus[1], i, p0.node, rg.root_plus, 1, p0.lit_bloom, false, has_pp0);
                    if (!has_pp0 || toks[i].size() < 2 || !(p0.lit_bloom & BLOOM_PLUS)) continue;
                    push_children(lit[2], plus[2], i, o.node, rg.root_plus ^ 1u, 2, o.lit_bloom, false, false);
                }
                const bool rev = pol.kind == 4 || pol.kind == 5 || pol.kind == 6 || pol.kind == 9 || pol.rev_boot, pbl = pol.kind == 6 || pol.kind == 7;
                for (int q = 0; q < 3; q++) {
                    const int p = rev ? 2 - q : q;
                    if (pbl) L.insert(L.end(), plus[p].begin(), plus[p].end());
                    L.insert(L.end(), lit[p].begin(), lit[p].end());
                    if (!pbl) L.insert(L.end(), plus[p].begin(), plus[p].end());
                }
            }
            uint32_t r = 0;
            bool ovf = L.size() > 176;
            if (pol.kind == 11) {
                const uint32_t QC = pol.t2;
                std::vector<Item> HI, LO;
                std::vector<std::vector<Item>> parked;
                const uint32_t HC = pol.valve; // != 2^30: HI has its own HC entries, LO shares QC entries with the matched ranges (<= 152 of them)
                const bool split = HC != (1u << 30);
                uint32_t pcount = 0;
                auto push_batch = [&](std::vector<Item>& b) { // one sink's pushes
                    if (b.empty()) return;
                    if (!split) {
                        if (HI.size() + LO.size() + b.size() > QC) {
                            if (!LO.empty()) parked.push_back(LO), LO.clear(), parks++;
                            if (HI.size() + b.size() > QC) parked.push_back(HI), HI.clear(), parks++;
                        }
                    } else {
                        size_t nh = 0, nl = 0;
                        for (auto& it : b) (it.rem >= pol.cap ? nh : nl)++;
                        if (HI.size() + nh > HC) parked.push_back(HI), HI.clear(), parks++, hi_parks++;
                        if (LO.size() + nl + pcount > QC) {
                            if (!LO.empty()) parked.push_back(LO), LO.clear(), parks++;
                            else pcount = 0, flushes++;
                        }
                    }
                    for (auto& it : b) (it.rem >= pol.cap ? HI : LO).push_back(it);
                    maxhi = std::max<uint64_t>(maxhi, HI.size());
                    maxlo = std::max<uint64_t>(maxlo, LO.size());
                };
                auto emit = [&](uint32_t n) {
                    if (!n) return;
                    ranges += n;
                    if (split && (pcount + n > 152 || LO.size() + pcount + n > QC)) pcount = 0, flushes++;
                    pcount += n;
                    maxp = std::max<uint64_t>(maxp, pcount);
                };
                { // boot: L holds the boot's pushes in order; split them back into the three sinks' batches is not needed: sizes <= 128 each
                    std::vector<Item> b;
                    size_t k = 0;
                    while (k < L.size()) { // feed in batches of <= 128 (a sink pushes at most 128)
                        b.assign(L.begin() + k, L.begin() + std::min(L.size(), k + 128));
                        k += b.size();
                        push_batch(b);
                    }
                    L.clear();
                }
                while (!HI.empty() || !LO.empty() || !parked.empty()) {
                    if (HI.empty() && LO.empty()) LO = parked.back(), parked.pop_back();
                    maxlist = std::max<uint64_t>(maxlist, HI.size() + LO.size());
                    std::vector<Item> cur;
                    while (cur.size() < 64 && !HI.empty()) cur.push_back(HI.back()), HI.pop_back();
                    while (cur.size() < 64 && !LO.empty()) cur.push_back(LO.back()), LO.pop_back();
                    r++;
                    items += cur.size();
                    fill[cur.size()]++;
                    std::vector<Item> lit[2], plus[2];
                    uint32_t em[2] = {0, 0};
                    for (const Item& it : cur) {
                        const TenantSlot& rg = h.dir[dslot[it.topic]];
                        const uint32_t tok = it.is_plus ? TOK_PLUS : toks[it.topic][it.level];
                        uint32_t bk = edge_bucket(it.node, tok, rg.buckets);
                        for (uint32_t probes = 0; probes < rg.buckets; probes++) {
                            const TrieSlot* hit = nullptr;
                            uint64_t slot = 0;
                            for (uint32_t j = 0; j < 2 && !hit; j++) {
                                const TrieSlot& e = h.trie[rg.base + 2 * bk + j];
                                if (e.parent == it.node && e.token == tok) hit = &e, slot = 2ull * bk + j;
                            }
                            if (hit) {
                                const TrieSlot& o = h.trie[rg.base + (slot ^ 1ull)];
                                const bool inner = it.level + 1 < toks[it.topic].size();
                                const bool beside = inner && (hit->lit_bloom & BLOOM_PLUS) && o.parent == hit->node && o.token == TOK_PLUS;
                                em[0] += (!inner && hit->own_count) + (hit->hash_count != 0);
                                push_children(lit[0], plus[0], it.topic, hit->node, slot, it.level + 1, hit->lit_bloom, false, beside);
                                if (beside) {
                                    em[1] += (it.level + 2 == toks[it.topic].size() && o.own_count) + (o.hash_count != 0);
                                    push_children(lit[1], plus[1], it.topic, o.node, slot ^ 1ull, it.level + 2, o.lit_bloom, false, false);
                                }
                                break;
                            }
                            if (h.trie[rg.base + 2 * bk].parent == NONE || h.trie[rg.base + 2 * bk + 1].parent == NONE) break;
                            bk = bk + 1 == rg.buckets ? 0 : bk + 1;
                        }
                    }
                    for (int p = 0; p < 2; p++) {
                        emit(em[p]);
                        std::vector<Item> b(lit[p]);
                        b.insert(b.end(), plus[p].begin(), plus[p].end());
                        push_batch(b);
                    }
                }
                waves++;
                rounds += r;
                rounds_of.push_back(r);
                continue;
            }
            while (!L.empty()) {
                maxlist = std::max<uint64_t>(maxlist, L.size());
                over176 += L.size() > 176;
                ovf = ovf || L.size() > 176;
                const uint32_t take = (uint32_t)std::min<size_t>(64, L.size());
                std::vector<Item> cur;
                if (pol.kind == 10) {
                    auto cls = [&](const Item& it) { return it.rem >= pol.cap ? 0 : (it.rem >= pol.t2 ? 1 : 2); };
                    const bool low_first = L.size() > pol.valve;
                    std::vector<Item> rest;
                    for (int pass = 0; pass < 3; pass++) {
                        const int want = low_first ? 2 - pass : pass;
                        for (size_t k = L.size(); k-- > 0;) {
                            if (L[k].node == 0xFFFFFFFEu) continue;
                            if (cls(L[k]) == want && cur.size() < take) cur.push_back(L[k]), L[k].node = 0xFFFFFFFEu;
                        }
                    }
                    for (auto& it : L) if (it.node != 0xFFFFFFFEu) rest.push_back(it);
                    L.swap(rest);
                } else if (pol.kind == 8 || pol.kind == 9) { // two stacks: HI (rem >= cap) taken first, newest first; then LO newest first
                    std::vector<Item> rest;
                    for (int pass = 0; pass < 2; pass++)
                        for (size_t k = L.size(); k-- > 0;) {
                            if (L[k].node == 0xFFFFFFFEu) continue;
                            const bool hi = L[k].rem >= pol.cap;
                            if ((pass == 0) == hi && cur.size() < take) cur.push_back(L[k]), L[k].node = 0xFFFFFFFEu;
                        }
                    for (auto& it : L) if (it.node != 0xFFFFFFFEu) rest.push_back(it);
                    L.swap(rest);
                } else if (pol.kind == 0 || pol.kind >= 3 || (pol.kind == 1 && L.size() > pol.cap)) {
                    cur.assign(L.end() - take, L.end());
                    L.erase(L.end() - take, L.end());
                } else if (pol.kind == 1) {
                    cur.assign(L.begin(), L.begin() + take);
                    L.erase(L.begin(), L.begin() + take);
                } else {
                    std::stable_sort(L.begin(), L.end(), [](const Item& a, const Item& b) { return a.rem < b.rem; }); // most levels left at the end
                    cur.assign(L.end() - take, L.end());
                    L.erase(L.end() - take, L.end());
                }
                r++;
                items += take;
                fill[take]++;
                std::vector<Item> lit[2], plus[2];
                for (const Item& it : cur) {
                    const TenantSlot& rg = h.dir[dslot[it.topic]];
                    const uint32_t tok = it.is_plus ? TOK_PLUS : toks[it.topic][it.level];
                    uint32_t bk = edge_bucket(it.node, tok, rg.buckets);
                    for (uint32_t probes = 0; probes < rg.buckets; probes++) {
                        const TrieSlot* hit = nullptr;
                        uint64_t slot = 0;
                        for (uint32_t j = 0; j < 2 && !hit; j++) {
                            const TrieSlot& e = h.trie[rg.base + 2 * bk + j];
                            if (e.parent == it.node && e.token == tok) hit = &e, slot = 2ull * bk + j;
                        }
                        if (hit) {
                            const TrieSlot& o = h.trie[rg.base + (slot ^ 1ull)];
                            const bool inner = it.level + 1 < toks[it.topic].size();
                            const bool beside = inner && (hit->lit_bloom & BLOOM_PLUS) && o.parent == hit->node && o.token == TOK_PLUS;
                            push_children(lit[0], plus[0], it.topic, hit->node, slot, it.level + 1, hit->lit_bloom, false, beside);
                            if (beside) push_children(lit[1], plus[1], it.topic, o.node, slot ^ 1ull, it.level + 2, o.lit_bloom, false, false);
                            break;
                        }
                        if (h.trie[rg.base + 2 * bk].parent == NONE || h.trie[rg.base + 2 * bk + 1].parent == NONE) break;
                        bk = bk + 1 == rg.buckets ? 0 : bk + 1;
                    }
                }
                if (pol.kind == 3) { // within a round's pushes: the items with the fewest levels left first (they end up deepest in the stack... no: on top = taken first)
                    std::vector<Item> all;
                    for (int p = 0; p < 2; p++) {
                        all.insert(all.end(), lit[p].begin(), lit[p].end());
                        all.insert(all.end(), plus[p].begin(), plus[p].end());
                    }
                    std::stable_sort(all.begin(), all.end(), [](const Item& a, const Item& b) { return a.rem < b.rem; });
                    L.insert(L.end(), all.begin(), all.end());
                } else {
                    const bool rev = pol.kind == 5 || pol.kind == 6, pbl = pol.kind == 6 || pol.kind == 7;
                    for (int q = 0; q < 2; q++) {
                        const int p = rev ? 1 - q : q;
                        if (pbl) L.insert(L.end(), plus[p].begin(), plus[p].end());
                        L.insert(L.end(), lit[p].begin(), lit[p].end());
                        if (!pbl) L.insert(L.end(), plus[p].begin(), plus[p].end());
                    }
                }
            }
            waves++;
            ovf_waves += ovf;
            rounds += r;
            rounds_of.push_back(r);
        }
        std::sort(rounds_of.begin(), rounds_of.end());
        uint64_t lt16 = 0, lt32 = 0, lt64 = 0, tot = 0;
        for (uint32_t f = 1; f <= 64; f++) tot += fill[f], lt16 += f < 16 ? fill[f] : 0, lt32 += f < 32 ? fill[f] : 0, lt64 += f < 64 ? fill[f] : 0;
        printf("%-22s waves %llu  rounds/wave %.2f (p50 %u p99 %u max %u)  items/wave %.1f  rounds with <16 / <32 / <64 items: %.1f%% / %.1f%% / %.1f%%  longest list %llu, rounds begun with > 176 items: %llu, waves that ever held > 176: %.1f%%, parks/wave %.2f (HI %.3f) flushes/wave %.3f max HI %llu LO %llu ranges %llu, ranges/wave %.1f\n",
               pol.name, (unsigned long long)waves, (double)rounds / waves, rounds_of[rounds_of.size() / 2], rounds_of[rounds_of.size() * 99 / 100], rounds_of.back(),
               (double)items / waves, 100.0 * lt16 / tot, 100.0 * lt32 / tot, 100.0 * lt64 / tot, (unsigned long long)maxlist, (unsigned long long)over176, 100.0 * ovf_waves / waves, (double)parks / waves, (double)hi_parks / waves, (double)flushes / waves, (unsigned long long)maxhi, (unsigned long long)maxlo, (unsigned long long)maxp, (double)ranges / waves);
    }
    return 0;
}
