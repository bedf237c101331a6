// host_fuzz.cpp -- test tool, not product: exercises the HOST side of the dist index (bmq_index.cpp: rebuild, incremental
// apply, region growth, directory, dictionary, indirect ranges) under AddressSanitizer/UBSan, without a GPU.
//   * model: std::set of route keys (byte order = KV order); ids must be the ranks in it;
//   * after every rebuild/apply the HBM image (TenantSlot directory, TrieSlot regions, DictSlot table, route_pos) is walked
//     on the CPU exactly as k_walk does it (bucket probes, Bloom mask, root payload from the directory) for random topics
//     and compared with a brute-force application of the matching rule of SURVEY.md 8a-0 to every key of the model.
// Build + run: make -C bifromq_amd/csrc fuzz   (tests/test_host.py runs a short round)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../bifromq_amd/csrc/bmq_index.h"

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

// the walk of k_walk, on the host image
static std::vector<uint32_t> image_match(const DistIndexHost& h, std::string_view tenant, std::string_view topic) {
    std::vector<uint32_t> ids;
    const uint32_t ttok = dict_find(h.dict, h.pool, tenant);
    if (ttok == TOK_UNKNOWN) return ids;
    const uint32_t mask = (uint32_t)h.tenants.size() - 1;
    uint32_t d = tenant_hash(ttok) & mask;
    while (h.tenants[d].token != ttok) {
        if (h.tenants[d].token == 0) return ids;
        d = (d + 1) & mask;
    }
    const TenantSlot rg = h.tenants[d];
    const auto levels = split(topic, '/');
    std::vector<uint32_t> toks;
    for (auto& l : levels) toks.push_back(dict_find(h.dict, h.pool, l));
    const bool sys = !levels[0].empty() && levels[0][0] == '$';
    auto emit = [&](uint32_t b, uint32_t cf) {
        const uint32_t c = cf & ~RANGE_INDIRECT;
        for (uint32_t i = 0; i < c; i++) ids.push_back((cf & RANGE_INDIRECT) ? h.route_pos[rg.rp_base + b + i] : rg.rank_base + b + i);
    };
    struct Item {
        uint32_t slot, dl;
    };
    std::vector<Item> st;
    auto visit = [&](uint32_t slot, uint32_t dl, uint32_t own_b, uint32_t own_c, uint32_t hash_b, uint32_t hash_c, uint32_t plus, uint32_t bloom) {
        const bool root_sys = dl == 0 && sys;
        if (dl == toks.size() && own_c) emit(own_b, own_c);
        if (hash_c && !root_sys) emit(hash_b, hash_c);
        if (dl < toks.size()) {
            const uint32_t t = toks[dl];
            if (t != TOK_UNKNOWN && ((bloom >> bloom_bit(t)) & 1u)) { // literal child: bucket probes, first-free order
                uint32_t bk = edge_bucket(slot, t, rg.buckets);
                for (;;) {
                    const TrieSlot& a = h.trie[rg.base + 2 * bk];
                    const TrieSlot& b = h.trie[rg.base + 2 * bk + 1];
                    if (a.parent == slot && a.token == t) { st.push_back({2 * bk, dl + 1}); break; }
                    if (b.parent == slot && b.token == t) { st.push_back({2 * bk + 1, dl + 1}); break; }
                    if (a.parent == NONE || b.parent == NONE) break;
                    bk = bk + 1 == rg.buckets ? 0 : bk + 1;
                }
            }
            if (plus != NONE && !root_sys) st.push_back({plus, dl + 1});
        }
    };
    visit(rg.root, 0, 0, 0, rg.root_hash_begin, rg.root_hash_count, rg.root_plus_child, rg.root_lit_bloom); // round 0: payload from the directory
    while (!st.empty()) {
        const Item it = st.back();
        st.pop_back();
        const TrieSlot& s = h.trie[rg.base + it.slot];
        visit(it.slot, it.dl, s.own_begin, s.own_count, s.hash_begin, s.hash_count, s.plus_child, s.lit_bloom);
    }
    std::sort(ids.begin(), ids.end());
    return ids;
}

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 30;
    const size_t max_keys = argc > 3 ? (size_t)atoll(argv[3]) : 3000; // keys of a rebuild round
    std::mt19937_64 rng(seed);
    const std::vector<std::string> tenants = {"t", "tenantB", "x", "a-much-longer-tenant-identifier"};
    const std::vector<std::string> alpha = {"a", "b", "c", "", "$sys", "+", "a-level-longer-than-sixteen-bytes", "\xE4\xBD\xA0\xE5\xA5\xBD", "0"};
    auto rnd = [&](size_t n) { return (size_t)(rng() % n); };
    auto rand_filter = [&]() {
        std::string f;
        const size_t depth = 1 + rnd(5);
        for (size_t i = 0; i < depth; i++) {
            if (i) f += '/';
            if (i + 1 == depth && rnd(5) == 0) f += "#";
            else f += alpha[rnd(alpha.size())];
        }
        return f;
    };
    auto rand_topic = [&]() {
        std::string t;
        const size_t depth = 1 + rnd(5);
        for (size_t i = 0; i < depth; i++) {
            if (i) t += '/';
            std::string l = alpha[rnd(alpha.size())];
            if (l == "+") l = "zz"; // topics carry no wildcards; "zz" is never a filter level
            t += l;
        }
        return t;
    };
    auto rand_key = [&]() {
        const std::string& tn = tenants[rnd(tenants.size())];
        const uint8_t flag = rnd(10) == 0 ? 2 : 1;
        const std::string recv = flag == 1 ? "0\0inbox" + std::to_string(rnd(40)) + std::string("\0d", 2) : "g" + std::to_string(rnd(3));
        return encode_route_key(tn, rand_filter(), flag, flag == 1 ? std::string("0\0", 2) + "inbox" + std::to_string(rnd(40)) + std::string("\0d", 2) : recv);
    };
    std::set<std::string> model;
    DistIndexHost h;
    uint64_t checks = 0;
    for (int round = 0; round < rounds; round++) {
        std::vector<std::string> keys;
        std::vector<uint8_t> ops;
        const bool full = round == 0 || rnd(8) == 0;
        if (full) {
            model.clear();
            const size_t n = rnd(3) == 0 ? 0 : 1 + rnd(max_keys);
            for (size_t i = 0; i < n; i++) model.insert(rand_key());
            keys.assign(model.begin(), model.end());
            std::shuffle(keys.begin(), keys.end(), rng);
        } else {
            const size_t n = 1 + rnd(rnd(4) == 0 ? 2000 : 60);
            for (size_t i = 0; i < n; i++) {
                if (!model.empty() && rnd(2)) { // delete an existing key (or, rarely, a key that is not there)
                    auto it = model.begin();
                    std::advance(it, rnd(std::min<size_t>(model.size(), 500)));
                    keys.push_back(rnd(20) ? *it : rand_key());
                    ops.push_back(1);
                } else {
                    keys.push_back(rand_key());
                    ops.push_back(0);
                }
            }
            for (size_t i = 0; i < keys.size(); i++) { // in order
                if (ops[i]) model.erase(keys[i]);
                else model.insert(keys[i]);
            }
        }
        std::vector<uint8_t> bytes;
        std::vector<uint32_t> off{0};
        for (auto& k : keys) {
            bytes.insert(bytes.end(), k.begin(), k.end());
            off.push_back((uint32_t)bytes.size());
        }
        const bool ok = full ? h.rebuild(bytes.data(), off.data(), (uint32_t)keys.size()) : h.apply(bytes.data(), off.data(), ops.data(), (uint32_t)keys.size());
        if (!ok) {
            fprintf(stderr, "round %d: %s failed: %s\n", round, full ? "rebuild" : "apply", h.error.c_str());
            return 1;
        }
        // ids are ranks
        if (h.n_routes != model.size()) {
            fprintf(stderr, "round %d: n_routes %llu != %zu\n", round, (unsigned long long)h.n_routes, model.size());
            return 1;
        }
        std::vector<std::string> ordered(model.begin(), model.end());
        for (size_t i = 0; i < ordered.size(); i += 1 + ordered.size() / 300)
            if (h.route_key((uint32_t)i) != ordered[i]) {
                fprintf(stderr, "round %d: route_key(%zu) differs\n", round, i);
                return 1;
            }
        // decoded form of the model, for the brute force
        struct Dec {
            std::string tenant;
            std::vector<std::string> levels;
        };
        std::vector<Dec> dec(ordered.size());
        for (size_t i = 0; i < ordered.size(); i++) {
            RouteKeyParts kp;
            if (!decode_route_key(ordered[i], kp)) return 2;
            dec[i].tenant = std::string(kp.tenant);
            dec[i].levels = split(kp.esc_filter, '\0');
        }
        for (int q = 0; q < 150; q++) {
            const std::string& tn = q % 25 == 24 ? std::string("nobody") : tenants[rnd(tenants.size())];
            const std::string topic = rand_topic();
            const auto tl = split(topic, '/');
            std::vector<uint32_t> want;
            for (size_t i = 0; i < dec.size(); i++)
                if (dec[i].tenant == tn && filter_matches(dec[i].levels, tl)) want.push_back((uint32_t)i);
            const auto got = image_match(h, tn, topic);
            checks++;
            if (got != want) {
                fprintf(stderr, "round %d (%s): tenant '%s' topic '%s': image gives %zu ids, the rule %zu\n", round, full ? "rebuild" : "apply", tn.c_str(),
                        topic.c_str(), got.size(), want.size());
                for (uint32_t id : want)
                    if (!std::binary_search(got.begin(), got.end(), id)) {
                        std::string f;
                        for (auto& l : dec[id].levels) f += (f.empty() ? "" : "/") + (l.empty() ? std::string("<empty>") : l);
                        fprintf(stderr, "  missing id %u filter %s\n", id, f.c_str());
                        std::string mq;
                        for (size_t li = 0; li < dec[id].levels.size(); li++) mq += (li ? "/" : "") + dec[id].levels[li];
                        {
                            auto it = h.by_name.find(tn);
                            if (it != h.by_name.end()) {
                                const TenantState& t = *it->second;
                                uint32_t used = 0;
                                for (uint32_t k = 0; k < 2 * t.buckets; k++) used += h.trie[t.base + k].parent != NONE;
                                fprintf(stderr, "  tenant state: keys %zu nodes %u buckets %u cap_slots %u used slots %u\n", t.keys.size(), t.n_nodes, t.buckets,
                                        t.cap_slots, used);
                            }
                        }
                        for (auto& e : h.by_name)
                            fprintf(stderr, "    tenant '%s': base %u cap_slots %u buckets %u nodes %u keys %zu\n", e.first.c_str(), e.second->base, e.second->cap_slots,
                                    e.second->buckets, e.second->n_nodes, e.second->keys.size());
                        fprintf(stderr, "    next_free %u trie.size %zu\n", h.next_free, h.trie.size());
                        {
                            const TenantState& t = *h.by_name.find(tn)->second;
                            for (uint32_t k = 0; k < t.cap_slots; k++) {
                                const TrieSlot& q = h.trie[t.base + k];
                                fprintf(stderr, "      slot %u: parent %x token %u own %u+%u hash %u+%u plus %x bloom %08x\n", k, q.parent, q.token, q.own_begin, q.own_count,
                                        q.hash_begin, q.hash_count, q.plus_child, q.lit_bloom);
                            }
                        }
                        const auto ff = h.find_filter(tn, mq);
                        fprintf(stderr, "  find_filter('%s') -> %zu ids%s\n", mq.c_str(), ff.size(), ff.empty() ? "" : (ff[0] == id ? " (first == id)" : " (other)"));
                        const uint32_t ttok = dict_find(h.dict, h.pool, tn);
                        uint32_t dd = tenant_hash(ttok) & ((uint32_t)h.tenants.size() - 1);
                        while (h.tenants[dd].token != ttok) dd = (dd + 1) & ((uint32_t)h.tenants.size() - 1);
                        const TenantSlot rg = h.tenants[dd];
                        const TrieSlot& root = h.trie[rg.base + rg.root];
                        fprintf(stderr, "  directory: root %u bloom %08x plus %u hash_count %u | root slot: bloom %08x plus %u hash_count %u\n", rg.root,
                                rg.root_lit_bloom, rg.root_plus_child, rg.root_hash_count, root.lit_bloom, root.plus_child, root.hash_count);
                    }
                for (uint32_t id : got)
                    if (!std::binary_search(want.begin(), want.end(), id)) fprintf(stderr, "  extra id %u\n", id);
                return 1;
            }
        }
    }
    printf("host_fuzz ok: seed %llu, %d rounds, %llu topic checks, final %zu routes\n", (unsigned long long)seed, rounds, (unsigned long long)checks, model.size());
    return 0;
}
