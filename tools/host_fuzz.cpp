// host_fuzz.cpp -- test tool, not product: runs the index BUILDER (bmq_build_core.h -- the code the gfx950 builder kernels
// execute, bmq_dist_index.h -- its host-side control) on host threads through HostExec, under AddressSanitizer/UBSan or
// ThreadSanitizer, without a GPU.
//   * model: std::map route key -> id (byte order = KV order); after a rebuild ids must be the ranks, a put gets the next id,
//     a key keeps its id until it is deleted;
//   * after every rebuild/apply the image (TenantSlot directory, TrieSlot regions, DictSlot table, route_pos, key store) is
//     walked on the CPU exactly as k_walk does it (bucket probes, Bloom mask, root payload from the directory) for random
//     topics and compared with a brute-force application of the matching rule of SURVEY.md 8a-0 to every key of the model;
//   * tiny initial capacities (BMQ_FUZZ_SMALL, default on) force region growth, dictionary growth, id-list pool growth,
//     directory growth and key-store growth all the time.
// Build + run: make -C bifromq_amd/csrc fuzz   (tests/test_host.py runs short rounds)
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../bifromq_amd/csrc/bmq_codec.h"
#include "../bifromq_amd/csrc/bmq_dist_index.h"
#include "../bifromq_amd/csrc/bmq_exec_host.h"
#include "../bifromq_amd/csrc/bmq_fanout.h"

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

static uint32_t dict_find_host(const DistIndex<HostExec>& h, std::string_view level) {
    LevelHash lh = level_hash_init();
    uint32_t inl[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < level.size(); i += 4) {
        uint32_t w = 0;
        for (size_t k = 0; k < 4 && i + k < level.size(); k++) w |= (uint32_t)(uint8_t)level[i + k] << (8 * k);
        level_hash_word(lh, w);
        if (i < 16) inl[i >> 2] = w;
    }
    const uint32_t gmask = h.dict_slots / DICT_GROUP - 1, tag = level_hash_tag(lh);
    uint32_t g = level_hash_slot(lh, (uint32_t)level.size()) & gmask;
    for (uint32_t probes = 0; probes <= gmask; probes++) {
        bool group_full = true;
        for (uint32_t j = 0; j < DICT_GROUP; j++) {
            const DictSlot& d = h.dict[DICT_GROUP * g + j];
            if (!d.tag) {
                group_full = false;
                continue;
            }
            if (d.tag == tag && d.len == level.size() && d.inl[0] == inl[0] && d.inl[1] == inl[1] && d.inl[2] == inl[2] && d.inl[3] == inl[3] &&
                (level.size() <= 16 || memcmp(h.dpool + d.pool_off, level.data(), level.size()) == 0))
                return d.token;
        }
        if (!group_full) return TOK_UNKNOWN;
        g = (g + 1) & gmask;
    }
    return TOK_UNKNOWN;
}

// the walk of k_walk, on the host image
static std::vector<uint32_t> image_match(const DistIndex<HostExec>& h, std::string_view tenant, std::string_view topic, uint64_t* visits, uint64_t* beside = nullptr /* [2]: '+' children found beside their parent, '+' children found below a non-root node */) {
    std::vector<uint32_t> ids;
    const DistIndexMut ix = h.mut();
    const uint32_t d = tenant_find(ix.tenants, ix.tenant_mask, ix.tenant_names, (const uint8_t*)tenant.data(), 0, tenant.size());
    if (d == NONE) return ids;
    const TenantSlot rg = h.dir[d];
    const auto levels = split(topic, '/');
    std::vector<uint32_t> toks;
    for (auto& l : levels) toks.push_back(dict_find_host(h, l));
    const bool sys = !levels[0].empty() && levels[0][0] == '$';
    auto emit = [&](uint32_t b, uint32_t cf) {
        const uint32_t c = cf & ~RANGE_INDIRECT;
        for (uint32_t i = 0; i < c; i++) ids.push_back((cf & RANGE_INDIRECT) ? h.route_pos[b + i] : b + i);
    };
    // Layout v3 (bmq_layout.h): the '+' child of a node lies in the OTHER slot of the node's line if that slot was free when the child came
    // into being, else at its hashed home; the root's '+' child P0 lies at its hashed home and the directory entry names the slot.  This walker reads the image by that rule and nothing else.
    constexpr uint64_t AT_ROOT = ~0ull;
    struct Item {
        uint32_t node, tok, level;
        uint64_t pslot; // where `node`'s own slot is (relative to the region), or AT_ROOT
    };
    std::vector<Item> st;
    auto visit = [&](uint32_t node, uint64_t slot, uint32_t dl, uint32_t own_b, uint32_t own_c, uint32_t hash_b, uint32_t hash_c, uint32_t bloom) {
        const bool root_sys = dl == 0 && sys;
        if (dl == toks.size() && own_c) emit(own_b, own_c);
        if (hash_c && !root_sys) emit(hash_b, hash_c);
        if (dl < toks.size()) {
            const uint32_t t = toks[dl];
            if (t != TOK_UNKNOWN && ((bloom >> bloom_bit(t)) & 1u)) st.push_back({node, t, dl, slot});
            if ((bloom & BLOOM_PLUS) && !root_sys) st.push_back({node, TOK_PLUS, dl, slot});
        }
    };
    visit(0, AT_ROOT, 0, 0, 0, rg.root_hash_begin, rg.root_hash_count, rg.root_lit_bloom); // round 0: payload from the directory
    while (!st.empty()) {
        const Item it = st.back();
        st.pop_back();
        auto arrive = [&](const TrieSlot* hit, uint64_t slot) {
            if (visits) (*visits)++;
            visit(hit->node, slot, it.level + 1, hit->own_begin, hit->own_count, hit->hash_begin, hit->hash_count, hit->lit_bloom);
        };
        if (it.tok == TOK_PLUS) {
            if (it.pslot == AT_ROOT) {
                if (rg.root_plus != NONE) {
                    const TrieSlot& p0 = h.trie[rg.base + rg.root_plus];
                    if (p0.parent != 0 || p0.token != TOK_PLUS) {
                        fprintf(stderr, "root_plus of a tenant does not name the root's '+' child\n");
                        abort();
                    }
                    arrive(&p0, rg.root_plus);
                    continue;
                }
            } else {
                const uint64_t ns = it.pslot ^ 1ull;
                const TrieSlot& o = h.trie[rg.base + ns];
                if (o.parent == it.node && o.token == TOK_PLUS) {
                    if (beside) beside[0]++, beside[1]++;
                    arrive(&o, ns);
                    continue;
                }
                // (a free or otherwise taken neighbour says nothing: a region growth leaves a '+' child that was not beside its parent at its hashed home)
            }
        }
        uint32_t bk = edge_bucket(it.node, it.tok, rg.buckets);
        for (uint32_t probes = 0; probes < rg.buckets; probes++) { // bucket probes, first-free order
            const TrieSlot& a = h.trie[rg.base + 2 * bk];
            const TrieSlot& b = h.trie[rg.base + 2 * bk + 1];
            if (a.parent == it.node && a.token == it.tok) {
                if (beside && it.tok == TOK_PLUS && it.pslot != AT_ROOT) beside[1]++;
                arrive(&a, 2ull * bk);
                break;
            }
            if (b.parent == it.node && b.token == it.tok) {
                if (beside && it.tok == TOK_PLUS && it.pslot != AT_ROOT) beside[1]++;
                arrive(&b, 2ull * bk + 1);
                break;
            }
            if (a.parent == NONE || b.parent == NONE) break;
            bk = bk + 1 == rg.buckets ? 0 : bk + 1;
        }
    }
    std::sort(ids.begin(), ids.end());
    return ids;
}

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 30;
    const size_t max_keys = argc > 3 ? (size_t)atoll(argv[3]) : 3000; // keys of a rebuild round
    const unsigned threads = argc > 4 ? (unsigned)atoi(argv[4]) : 4;
    std::mt19937_64 rng(seed);
    std::vector<std::string> tenants = {"t", "tenantB", "x", "a-much-longer-tenant-identifier", ""};
    const std::vector<std::string> alpha = {"a", "b", "c", "", "$sys", "+", "a-level-longer-than-sixteen-bytes", "\xE4\xBD\xA0\xE5\xA5\xBD", "0",
                                            "exactly-16-bytes", "seventeen-bytes-x", "#"};
    auto rnd = [&](size_t n) { return (size_t)(rng() % n); };
    auto rand_filter = [&]() {
        std::string f;
        const size_t depth = 1 + rnd(5);
        for (size_t i = 0; i < depth; i++) {
            if (i) f += '/';
            if (i + 1 == depth && rnd(5) == 0) f += "#";
            else {
                std::string l = alpha[rnd(alpha.size())];
                if (l == "#") l = "#x"; // '#' is a wildcard only as the last level; as a literal level it never appears in a valid filter
                f += l;
            }
        }
        return f;
    };
    auto rand_topic = [&]() {
        std::string t;
        const size_t depth = 1 + rnd(5);
        for (size_t i = 0; i < depth; i++) {
            if (i) t += '/';
            std::string l = alpha[rnd(alpha.size())];
            if (l == "+" || l == "#") l = "zz"; // topics carry no wildcards; "zz" is never a filter level
            t += l;
        }
        return t;
    };
    uint64_t extra_tenants = 0;
    auto rand_key = [&]() {
        if (rnd(400) == 0) tenants.push_back("late-tenant-" + std::to_string(extra_tenants++)); // tenants appear over time
        const std::string& tn = tenants[rnd(tenants.size())];
        const uint8_t flag = rnd(10) == 0 ? 2 : 1;
        // receiverUrl = subBrokerId NUL receiverId NUL delivererKey: a few brokers x a few deliverer keys (the fan-out grouping below)
        return encode_route_key(tn, rand_filter(), flag,
                                flag == 1 ? std::to_string(rnd(3)) + std::string("\0", 1) + "inbox" + std::to_string(rnd(40)) + std::string("\0d", 2) +
                                                std::to_string(rnd(12))
                                          : "g" + std::to_string(rnd(3)));
    };
    std::map<std::string, uint32_t> model; // key -> id
    uint32_t next_id = 0;
    HostExec hx;
    hx.threads = threads;
    DistIndex<HostExec> h(hx);
    h.tiny = getenv("BMQ_FUZZ_BIG") == nullptr;
    uint64_t checks = 0, n_apply = 0, n_rebuild = 0, fo_pairs = 0, n_generations = 0, plus_stat[2] = {0, 0};
    Fanout<HostExec> fo(hx, h); // fan-out grouping (bmq_fanout.h) over the same index, kept across rebuilds and applies
    fo.initial_table = 4;       // 36 deliverer keys: the group table grows twice
    for (int round = 0; round < rounds; round++) {
        std::vector<std::string> keys;
        std::vector<uint8_t> ops;
        const bool full = round == 0 || rnd(8) == 0;
        if (full) {
            model.clear();
            const size_t n = rnd(3) == 0 ? 0 : 1 + rnd(max_keys);
            std::set<std::string> ks;
            for (size_t i = 0; i < n; i++) ks.insert(rand_key());
            keys.assign(ks.begin(), ks.end());
            uint32_t r = 0;
            for (auto& k : ks) model[k] = r++;
            next_id = r;
            if (rnd(3) == 0) { // not a KV scan: shuffled and with a duplicate
                std::shuffle(keys.begin(), keys.end(), rng);
                if (!keys.empty()) keys.push_back(keys[0]);
            }
            n_rebuild++;
        } else {
            const size_t n = 1 + rnd(rnd(4) == 0 ? 2000 : 60);
            for (size_t i = 0; i < n; i++) {
                if (!model.empty() && rnd(2)) { // delete an existing key (or, rarely, a key that is not there)
                    auto it = model.begin();
                    std::advance(it, rnd(std::min<size_t>(model.size(), 500)));
                    keys.push_back(rnd(20) ? it->first : rand_key());
                    ops.push_back(1);
                } else {
                    keys.push_back(rnd(6) == 0 && !keys.empty() ? keys[rnd(keys.size())] : rand_key()); // sometimes a key of this very batch again
                    ops.push_back(0);
                }
            }
            uint32_t put_no = 0;
            for (size_t i = 0; i < keys.size(); i++) { // in order; a put's id = next_id + number of puts before it
                if (ops[i]) model.erase(keys[i]);
                else {
                    if (!model.count(keys[i])) model[keys[i]] = next_id + put_no;
                    put_no++;
                }
            }
            next_id += put_no;
            n_apply++;
        }
        std::vector<uint8_t> bytes;
        std::vector<uint32_t> off{0};
        for (auto& k : keys) {
            bytes.insert(bytes.end(), k.begin(), k.end());
            off.push_back((uint32_t)bytes.size());
        }
        bytes.resize(bytes.size() + 16, 0);
        const bool ok = full ? h.rebuild(bytes.data(), off.data(), (uint32_t)keys.size()) : h.apply(bytes.data(), off.data(), ops.data(), (uint32_t)keys.size());
        if (!ok) {
            fprintf(stderr, "round %d: %s failed: %s\n", round, full ? "rebuild" : "apply", h.error.c_str());
            return 1;
        }
        DistIndexStats st;
        h.stats(st);
        if (st.n_routes != model.size() || st.next_id != next_id) {
            fprintf(stderr, "round %d (%s): n_routes %llu (model %zu) next_id %u (model %u)\n", round, full ? "rebuild" : "apply",
                    (unsigned long long)st.n_routes, model.size(), st.next_id, next_id);
            return 1;
        }
        { // tenants with routes
            std::set<std::string> live;
            for (auto& e : model) {
                RouteKeyParts kp;
                decode_route_key(e.first, kp);
                live.insert(std::string(kp.tenant));
            }
            if (st.n_tenants != live.size()) {
                fprintf(stderr, "round %d: n_tenants %llu != %zu\n", round, (unsigned long long)st.n_tenants, live.size());
                return 1;
            }
        }
        // id -> key, single and batched
        std::vector<uint32_t> all_ids;
        for (auto& e : model) all_ids.push_back(e.second);
        size_t q = 0;
        for (auto& e : model) {
            if (q++ % (1 + model.size() / 300)) continue;
            std::string k;
            if (!h.route_key(e.second, k) || k != e.first) {
                fprintf(stderr, "round %d: route_key(%u) differs\n", round, e.second);
                return 1;
            }
        }
        {
            std::vector<uint8_t> kb;
            std::vector<uint64_t> ko;
            if (!h.route_keys(all_ids.data(), (uint32_t)all_ids.size(), kb, ko)) return 3;
            size_t i = 0;
            for (auto& e : model) {
                if (std::string_view((const char*)kb.data() + ko[i], ko[i + 1] - ko[i]) != e.first) {
                    fprintf(stderr, "round %d: route_keys entry %zu differs\n", round, i);
                    return 1;
                }
                i++;
            }
        }
        // ---- a generation change beside this index (bmq_compact_begin / _poll / _swap: DistIndex::reserve_like, import_snapshot, import_apply,
        // the log, its replay): chunks of random size, `h` mutated between a chunk's snapshot and its apply and between the chunks; afterwards
        // the new generation holds exactly the model's keys, with dense ids + at most the replayed ops' garbage.  `h` stays the index of the
        // following rounds (the engine would swap the two).
        if (h.built && rnd(4) == 0) {
            DistIndex<HostExec> g(hx);
            g.tiny = h.tiny;
            if (!g.reserve_like(h)) {
                fprintf(stderr, "round %d: reserve_like failed: %s\n", round, g.error.c_str());
                return 1;
            }
            h.defer_release = true;
            std::vector<std::string> log_keys;
            std::vector<uint8_t> log_ops;
            auto mutate = [&]() -> bool { // a small batch into h, the model and the log
                std::vector<std::string> mk;
                std::vector<uint8_t> mo;
                const size_t n = 1 + rnd(40);
                for (size_t i = 0; i < n; i++) {
                    if (!model.empty() && rnd(2)) {
                        auto it = model.begin();
                        std::advance(it, rnd(std::min<size_t>(model.size(), 500)));
                        mk.push_back(it->first);
                        mo.push_back(1);
                    } else {
                        mk.push_back(rnd(5) == 0 && !mk.empty() ? mk[rnd(mk.size())] : rand_key());
                        mo.push_back(0);
                    }
                }
                uint32_t put_no = 0;
                for (size_t i = 0; i < mk.size(); i++) {
                    if (mo[i]) model.erase(mk[i]);
                    else {
                        if (!model.count(mk[i])) model[mk[i]] = next_id + put_no;
                        put_no++;
                    }
                }
                next_id += put_no;
                std::vector<uint8_t> mb;
                std::vector<uint32_t> mf{0};
                for (auto& k : mk) {
                    mb.insert(mb.end(), k.begin(), k.end());
                    mf.push_back((uint32_t)mb.size());
                }
                mb.resize(mb.size() + 16, 0);
                log_keys.insert(log_keys.end(), mk.begin(), mk.end());
                log_ops.insert(log_ops.end(), mo.begin(), mo.end());
                n_apply++;
                return h.apply(mb.data(), mf.data(), mo.data(), (uint32_t)mk.size());
            };
            const uint32_t n_ids = h.id_bound();
            uint64_t carried = 0;
            for (uint32_t cursor = 0; cursor < n_ids;) {
                const uint32_t hi = (uint32_t)std::min<uint64_t>(n_ids, (uint64_t)cursor + 1 + rnd(rnd(3) == 0 ? 2000 : 150));
                uint32_t n_live = 0;
                if (!g.import_snapshot(h, cursor, hi) || (rnd(3) == 0 && !mutate()) || !g.import_apply(n_live)) {
                    fprintf(stderr, "round %d: generation change failed at id %u: %s | %s\n", round, cursor, g.error.c_str(), h.error.c_str());
                    return 1;
                }
                carried += n_live;
                cursor = hi;
                if (rnd(4) == 0 && !mutate()) {
                    fprintf(stderr, "round %d: apply during a generation change failed: %s\n", round, h.error.c_str());
                    return 1;
                }
            }
            if (!log_keys.empty()) { // the replay: ONE merged batch, in order
                std::vector<uint8_t> mb;
                std::vector<uint32_t> mf{0};
                for (auto& k : log_keys) {
                    mb.insert(mb.end(), k.begin(), k.end());
                    mf.push_back((uint32_t)mb.size());
                }
                mb.resize(mb.size() + 16, 0);
                if (!g.apply(mb.data(), mf.data(), log_ops.data(), (uint32_t)log_keys.size())) {
                    fprintf(stderr, "round %d: replay failed: %s\n", round, g.error.c_str());
                    return 1;
                }
            }
            h.defer_release = false;
            h.release_deferred();
            DistIndexStats gs;
            g.stats(gs);
            if (gs.n_routes != model.size() || gs.next_id - gs.n_routes > log_keys.size()) {
                fprintf(stderr, "round %d: new generation: %llu routes (model %zu), %u ids, %zu ops replayed, %llu carried\n", round,
                        (unsigned long long)gs.n_routes, model.size(), gs.next_id, log_keys.size(), (unsigned long long)carried);
                return 1;
            }
            std::vector<uint32_t> ids(gs.next_id);
            for (uint32_t i = 0; i < gs.next_id; i++) ids[i] = i;
            std::vector<uint8_t> kb;
            std::vector<uint64_t> ko;
            if (!g.route_keys(ids.data(), gs.next_id, kb, ko)) return 3;
            std::set<std::string> got;
            for (uint32_t i = 0; i < gs.next_id; i++)
                if (ko[i + 1] > ko[i]) got.insert(std::string((const char*)kb.data() + ko[i], ko[i + 1] - ko[i]));
            if (got.size() != model.size() || !std::equal(got.begin(), got.end(), model.begin(), [](const std::string& a, const auto& b) { return a == b.first; })) {
                fprintf(stderr, "round %d: the new generation's key set differs from the model (%zu keys against %zu)\n", round, got.size(), model.size());
                return 1;
            }
            n_generations++;
        }
        // decoded form of the model, for the brute force
        struct Dec {
            std::string tenant;
            std::vector<std::string> levels;
            uint32_t id;
        };
        std::vector<Dec> dec;
        std::map<std::pair<std::string, std::string>, std::vector<uint32_t>> by_filter;
        for (auto& e : model) {
            RouteKeyParts kp;
            if (!decode_route_key(e.first, kp)) return 2;
            dec.push_back({std::string(kp.tenant), split(kp.esc_filter, '\0'), e.second});
            std::string mq(kp.esc_filter);
            for (auto& c : mq)
                if (c == '\0') c = '/';
            by_filter[{std::string(kp.tenant), mq}].push_back(e.second);
        }
        { // exact filter lookup
            size_t i = 0;
            for (auto& e : by_filter) {
                if (i++ % (1 + by_filter.size() / 60)) continue;
                std::vector<uint32_t> got, want = e.second;
                std::sort(want.begin(), want.end());
                if (!h.find(e.first.first, e.first.second, got)) return 3;
                std::sort(got.begin(), got.end());
                if (got != want) {
                    fprintf(stderr, "round %d: find('%s','%s') gives %zu ids, model %zu\n", round, e.first.first.c_str(), e.first.second.c_str(), got.size(),
                            want.size());
                    return 1;
                }
            }
        }
        if (h.built) { // fan-out grouping of a random CSR (ids live, deleted and never handed out) against the model
            std::map<uint32_t, const std::string*> by_id;
            for (auto& e : model) by_id[e.second] = &e.first;
            const uint32_t n_rows = 1 + (uint32_t)rnd(rnd(4) == 0 ? 800 : 40); // now and then enough pairs for several host threads
            std::vector<uint32_t> row{0}, ids;
            for (uint32_t r = 0; r < n_rows; r++) {
                std::set<uint32_t> s;
                for (size_t k = rnd(4) ? rnd(30) : 0; k > 0; k--) s.insert((uint32_t)rnd(next_id + 3));
                ids.insert(ids.end(), s.begin(), s.end());
                row.push_back((uint32_t)ids.size());
            }
            const uint32_t total = (uint32_t)ids.size(), gcap = 64;
            std::vector<uint32_t> ot(total + 1), orr(total + 1), goff(gcap + 1), grep(gcap);
            ids.resize(total + 4);
            FanoutResult fr;
            if (!fo.group(row.data(), ids.data(), n_rows, total, ot.data(), orr.data(), goff.data(), grep.data(), gcap, fr) || fr.group_overflow) {
                fprintf(stderr, "round %d: fan-out grouping failed: %s\n", round, fo.error.c_str());
                return 1;
            }
            // model: DelivererKey (subBrokerId, delivererKey) / "shared" / "dead" -> pairs in (row, id) order
            std::map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> want, got;
            for (uint32_t r = 0; r < n_rows; r++)
                for (uint32_t k = row[r]; k < row[r + 1]; k++) {
                    auto it = by_id.find(ids[k]);
                    std::string name = "dead";
                    if (it != by_id.end()) {
                        RouteKeyParts kp;
                        decode_route_key(*it->second, kp);
                        if (kp.flag != 1) name = "shared";
                        else {
                            const auto parts = split(kp.receiver, '\0');
                            name = "k:" + parts[0] + "|" + parts[2];
                        }
                    }
                    want[name].push_back({r, ids[k]});
                }
            bool bad = goff[0] != 0 || goff[fr.n_groups] != total;
            for (uint32_t g = 0; g < fr.n_groups && !bad; g++) {
                std::string name = grep[g] == 0xFFFFFFFFu ? "dead" : grep[g] == 0xFFFFFFFEu ? "shared" : "";
                if (name.empty()) {
                    auto it = by_id.find(grep[g]);
                    if (it == by_id.end()) {
                        bad = true;
                        break;
                    }
                    RouteKeyParts kp;
                    decode_route_key(*it->second, kp);
                    const auto parts = split(kp.receiver, '\0');
                    name = "k:" + parts[0] + "|" + parts[2];
                }
                if (got.count(name)) bad = true;
                for (uint32_t k = goff[g]; k < goff[g + 1]; k++) got[name].push_back({ot[k], orr[k]});
            }
            const uint32_t sp = (want.count("shared") ? 1u : 0u) | (want.count("dead") ? 2u : 0u);
            if (bad || got != want || fr.special != sp) {
                fprintf(stderr, "round %d: fan-out groups differ from the model (%zu vs %zu groups, special %u vs %u)\n", round, got.size(), want.size(),
                        fr.special, sp);
                return 1;
            }
            fo_pairs += total;
        }
        for (int tq = 0; tq < 150; tq++) {
            const std::string& tn = tq % 25 == 24 ? std::string("nobody") : tenants[rnd(tenants.size())];
            const std::string topic = rand_topic();
            const auto tl = split(topic, '/');
            std::vector<uint32_t> want;
            for (auto& d : dec)
                if (d.tenant == tn && filter_matches(d.levels, tl)) want.push_back(d.id);
            std::sort(want.begin(), want.end());
            const auto got = image_match(h, tn, topic, nullptr, plus_stat);
            checks++;
            if (got != want) {
                fprintf(stderr, "round %d (%s): tenant '%s' topic '%s': image gives %zu ids, the rule %zu\n", round, full ? "rebuild" : "apply", tn.c_str(),
                        topic.c_str(), got.size(), want.size());
                for (uint32_t id : want)
                    if (!std::binary_search(got.begin(), got.end(), id)) fprintf(stderr, "  missing id %u\n", id);
                for (uint32_t id : got)
                    if (!std::binary_search(want.begin(), want.end(), id)) fprintf(stderr, "  extra id %u\n", id);
                return 1;
            }
        }
    }
    DistIndexStats st;
    h.stats(st);
    printf("layout v3: %llu of the %llu nodes the checks discovered through a '+' edge below a non-root node lay beside their parent\n", (unsigned long long)plus_stat[0], (unsigned long long)plus_stat[1]);
    printf("host_fuzz ok: seed %llu, %d rounds (%llu rebuilds, %llu applies, %llu generation changes), %llu topic checks, final %zu routes, %llu nodes, %llu tokens, "
           "%llu trie slots (%llu garbage), %llu id-list words (%llu garbage), %llu fan-out pairs grouped\n",
           (unsigned long long)seed, rounds, (unsigned long long)n_rebuild, (unsigned long long)n_apply, (unsigned long long)n_generations, (unsigned long long)checks, model.size(),
           (unsigned long long)st.n_nodes, (unsigned long long)st.n_tokens, (unsigned long long)st.trie_slots, (unsigned long long)st.trie_garbage_slots,
           (unsigned long long)st.id_list_words, (unsigned long long)st.id_list_garbage, (unsigned long long)fo_pairs);
    return 0;
}
