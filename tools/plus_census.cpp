// tools/plus_census.cpp -- layout v3 planning / checking tool (CPU only): builds the route index of a few tenants of bench.py's C3 population with
// the product's own builder on the host executor (the functions the GPU kernels run), walks a batch of the workload's publishes over the image
// by the layout's reading rule and reports how many of the nodes a publish discovers need a line fetch of their own:
//   visits = nodes discovered (N_visit of SURVEY 8d), plus = those reached over a '+' edge, beside = '+' children found in their parent's line,
//   root_plus = the root's '+' child (taken from the directory entry's reference: no vector fetch in a tenant-grouped batch).
//     g++ -O2 -std=c++17 -pthread -I bifromq_amd/csrc tools/plus_census.cpp bifromq_amd/csrc/bmq_gen.cpp bifromq_amd/csrc/bmq_codec.cpp -o /tmp/plus_census && /tmp/plus_census [tenants=32] [topics=200000]
#include <cstdio>
#include <cstdlib>
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
    // the image: '+' children by placement
    uint64_t n_nodes = 0, n_plus = 0, n_plus_beside = 0, n_root_plus = 0;
    for (uint32_t d = 0; d < h.dir_slots; d++) {
        const TenantSlot& t = h.dir[d];
        if (!(t.hash_lo | t.hash_hi)) continue;
        for (uint32_t s = 0; s < 2 * t.buckets; s++) {
            const TrieSlot& e = h.trie[t.base + s];
            if (e.parent == NONE) continue;
            n_nodes++;
            if (e.token != TOK_PLUS) continue;
            if (e.parent == 0) {
                n_root_plus++;
                continue;
            }
            n_plus++;
            const TrieSlot& o = h.trie[t.base + (s ^ 1u)];
            n_plus_beside += o.parent != NONE && o.node == e.parent;
        }
    }
    printf("image: %llu nodes in %u tenants, %llu '+' children below non-root nodes of which %llu (%.1f %%) lie beside their parent, %llu root '+' children\n",
           (unsigned long long)n_nodes, n_ten, (unsigned long long)n_plus, (unsigned long long)n_plus_beside, 100.0 * n_plus_beside / (n_plus ? n_plus : 1),
           (unsigned long long)n_root_plus);
    // the publishes
    const uint32_t n = bmqgen_topics(g, 11, n_topics, 0, n_ten, 900, 1);
    const uint8_t* tb = bmqgen_topic_bytes(g);
    const uint32_t* to = bmqgen_topic_off(g);
    const uint32_t* tt = bmqgen_topic_tenant(g);
    const uint8_t* nb = bmqgen_tenant_bytes(g);
    const uint32_t* no = bmqgen_tenant_off(g);
    const DistIndexMut ix = h.mut();
    uint64_t visits = 0, plus = 0, beside = 0, root_plus = 0, fetches = 0, below_beside = 0, nb_taken = 0, nb_free = 0;
    struct Item {
        uint32_t node, level;
        bool is_plus;
        uint64_t pslot;
    };
    constexpr uint64_t AT_ROOT = ~0ull;
    std::vector<Item> st;
    std::vector<uint32_t> toks;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t d = tenant_find(ix.tenants, ix.tenant_mask, ix.tenant_names, nb, no[tt[i]], no[tt[i] + 1]);
        if (d == NONE) continue;
        const TenantSlot& rg = h.dir[d];
        toks.clear();
        unsigned long long pos = to[i];
        const unsigned long long end = to[i + 1];
        for (;;) {
            LevelHash lh;
            uint32_t inl[4], len;
            const unsigned long long start = pos;
            scan_level_bytes<0x2F2F2F2Fu>(tb, pos, end, lh, inl, len);
            toks.push_back(dict_intern(ix, lh, len, inl, tb, start, false));
            if (pos >= end) break;
            pos++;
        }
        const bool sys = end > to[i] && tb[to[i]] == '$';
        st.clear();
        auto visit = [&](uint32_t node, uint64_t slot, uint32_t dl, uint32_t bloom) {
            if (dl >= toks.size()) return;
            const uint32_t t = toks[dl];
            if (t != TOK_UNKNOWN && ((bloom >> bloom_bit(t)) & 1u)) st.push_back({node, dl, false, slot});
            if ((bloom & BLOOM_PLUS) && !(dl == 0 && sys)) st.push_back({node, dl, true, slot});
        };
        visit(0, AT_ROOT, 0, rg.root_lit_bloom);
        while (!st.empty()) {
            const Item it = st.back();
            st.pop_back();
            const uint32_t tok = it.is_plus ? TOK_PLUS : toks[it.level];
            if (it.is_plus && it.pslot == AT_ROOT && rg.root_plus != NONE) {
                const TrieSlot& p0 = h.trie[rg.base + rg.root_plus];
                visits++, plus++, root_plus++;
                visit(p0.node, rg.root_plus, it.level + 1, p0.lit_bloom);
                continue;
            }
            if (it.is_plus && it.pslot != AT_ROOT) {
                const TrieSlot& o = h.trie[rg.base + (it.pslot ^ 1ull)];
                if (o.parent == it.node && o.token == TOK_PLUS) {
                    visits++, plus++, beside++;
                    visit(o.node, it.pslot ^ 1ull, it.level + 1, o.lit_bloom);
                    continue;
                }
            }
            if (it.is_plus && it.pslot != AT_ROOT) { // a '+' child that costs a fetch of its own: why?
                {
                    const TrieSlot& o = h.trie[rg.base + (it.pslot ^ 1ull)];
                    const TrieSlot& me = h.trie[rg.base + it.pslot];
                    if (o.parent == NONE) nb_free++;
                    else if (me.token == TOK_PLUS && o.node == me.parent) below_beside++; // the neighbour is this node's own parent
                    else nb_taken++;
                }
            }
            uint32_t bk = edge_bucket(it.node, tok, rg.buckets);
            for (uint32_t probes = 0; probes < rg.buckets; probes++) {
                fetches++;
                const TrieSlot* hit = nullptr;
                uint64_t slot = 0;
                for (uint32_t j = 0; j < 2 && !hit; j++) {
                    const TrieSlot& e = h.trie[rg.base + 2 * bk + j];
                    if (e.parent == it.node && e.token == tok) hit = &e, slot = 2ull * bk + j;
                }
                if (hit) {
                    visits++, plus += it.is_plus;
                    visit(hit->node, slot, it.level + 1, hit->lit_bloom);
                    break;
                }
                if (h.trie[rg.base + 2 * bk].parent == NONE || h.trie[rg.base + 2 * bk + 1].parent == NONE) break;
                bk = bk + 1 == rg.buckets ? 0 : bk + 1;
            }
        }
    }
    printf("publishes: %u; per publish: %.2f nodes discovered, %.2f of them over a '+' edge (%.2f the root's, %.2f beside their parent), %.2f line fetches "
           "(layout v2: one per probe = %.2f + misses)\n",
           n, (double)visits / n, (double)plus / n, (double)root_plus / n, (double)beside / n, (double)fetches / n, (double)visits / n);
    printf("'+' probes that needed a fetch, per publish: %.2f below a '+' child that lies beside ITS parent, %.2f neighbour slot taken by "
           "another edge, %.2f neighbour slot free (the child was left at its hashed home by a region growth, or does not exist: Bloom false positive)\n",
           (double)below_beside / n, (double)nb_taken / n, (double)nb_free / n);
    return 0;
}
