// retain_fuzz.cpp -- test tool, not product: the HOST side of the retain index (bmq_retain.cpp: rebuild, per-tenant add/remove,
// segment growth) under AddressSanitizer/UBSan, without a GPU.  After every step
//   * ids must enumerate (tenant, topic) in (tenant bytes, level-list bytes) order;
//   * a CPU walk over the HBM image (directory, breadth-first node array, edge hash, '$' runs) done the way k_retain_walk does
//     it -- frontier of node ranges, '+' = children range, '#' = subtree id ranges -- must return exactly the topics the rule of
//     SURVEY.md 8a-0 selects by brute force.
// Build + run: make -C bifromq_amd/csrc fuzz   (tests/test_host.py runs a short round)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../bifromq_amd/csrc/bmq_retain.h"

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

static std::vector<uint32_t> image_match(const RetainIndexHost& h, std::string_view tenant, std::string_view filter) {
    std::vector<uint32_t> ids;
    const uint32_t ttok = dict_find(h.dict, h.pool, tenant);
    if (ttok == TOK_UNKNOWN) return ids;
    const uint32_t mask = (uint32_t)h.tenants.size() - 1;
    uint32_t d = tenant_hash(ttok) & mask;
    while (h.tenants[d].token != ttok) {
        if (h.tenants[d].token == 0) return ids;
        d = (d + 1) & mask;
    }
    const RTenantSlot ten = h.tenants[d];
    auto node = [&](uint32_t local) -> const RNode& { return h.nodes[ten.node_base + local]; };
    const auto levels = split(filter, '/');
    std::vector<std::pair<uint32_t, uint32_t>> cur{{0, 1}}, nxt; // node ranges (begin, count), tenant-local
    auto emit = [&](uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; i++) ids.push_back(ten.id_base + i);
    };
    for (size_t li = 0; li < levels.size(); li++) {
        const std::string& lv = levels[li];
        if (lv == "#" && li + 1 == levels.size()) {
            for (auto& r : cur)
                for (uint32_t n = r.first; n < r.first + r.second; n++) {
                    const RNode& q = node(n);
                    if (li == 0) { // '#' at position 0: everything but the '$' topics
                        emit(q.sub_begin, std::min(q.sub_end, ten.sys_id_lo > q.sub_begin ? ten.sys_id_lo : q.sub_begin));
                        emit(std::max(q.sub_begin, ten.sys_id_hi), q.sub_end);
                    } else emit(q.sub_begin, q.sub_end);
                }
            std::sort(ids.begin(), ids.end());
            return ids;
        }
        nxt.clear();
        if (lv == "+") {
            for (auto& r : cur) {
                const uint32_t cb = node(r.first).child_begin;
                const RNode& last = node(r.first + r.second - 1);
                const uint32_t ce = last.child_begin + (last.child_count & ~RN_TERM);
                if (ce <= cb) continue;
                if (li == 0) { // skip the run of '$' children
                    if (ten.sys_node_lo > cb) nxt.push_back({cb, std::min(ce, ten.sys_node_lo) - cb});
                    if (ce > ten.sys_node_hi && ten.sys_node_hi >= cb) nxt.push_back({std::max(cb, ten.sys_node_hi), ce - std::max(cb, ten.sys_node_hi)});
                    if (ten.sys_node_lo == ten.sys_node_hi && ten.sys_node_lo <= cb) { nxt.clear(); nxt.push_back({cb, ce - cb}); }
                } else nxt.push_back({cb, ce - cb});
            }
        } else {
            const uint32_t tok = dict_find(h.dict, h.pool, lv);
            if (tok != TOK_UNKNOWN)
                for (auto& r : cur)
                    for (uint32_t n = r.first; n < r.first + r.second; n++) {
                        uint32_t bk = redge_bucket(n, tok, ten.edge_bucket_mask);
                        for (uint32_t probes = 0; probes <= ten.edge_bucket_mask; probes++) {
                            const REdge* e = &h.edges[ten.edge_base + 4 * (size_t)bk];
                            bool hit = false, hole = false;
                            for (int j = 0; j < 4; j++) {
                                if (e[j].parent == n && e[j].token == tok) {
                                    nxt.push_back({e[j].child, 1});
                                    hit = true;
                                }
                                hole |= e[j].parent == NONE;
                            }
                            if (hit || hole) break;
                            bk = (bk + 1) & ten.edge_bucket_mask;
                        }
                    }
        }
        cur.swap(nxt);
        if (cur.empty()) break;
    }
    for (auto& r : cur)
        for (uint32_t n = r.first; n < r.first + r.second; n++)
            if (node(n).child_count & RN_TERM) ids.push_back(ten.id_base + node(n).sub_begin);
    std::sort(ids.begin(), ids.end());
    return ids;
}

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 30;
    std::mt19937_64 rng(seed);
    const std::vector<std::string> tenants = {"t", "tenantB", "x", "a-much-longer-tenant-identifier"};
    const std::vector<std::string> alpha = {"a", "b", "c", "", "$sys", "$x", "a-level-longer-than-sixteen-bytes", "\xE4\xBD\xA0\xE5\xA5\xBD", "0"};
    auto rnd = [&](size_t n) { return (size_t)(rng() % n); };
    auto rand_topic = [&]() {
        std::string t;
        const size_t depth = 1 + rnd(5);
        for (size_t i = 0; i < depth; i++) t += (i ? "/" : "") + alpha[rnd(alpha.size())];
        return t;
    };
    auto rand_filter = [&]() {
        std::string f;
        const size_t depth = 1 + rnd(5);
        for (size_t i = 0; i < depth; i++) {
            if (i) f += '/';
            if (i + 1 == depth && rnd(3) == 0) f += "#";
            else if (rnd(3) == 0) f += "+";
            else f += alpha[rnd(alpha.size())];
        }
        return f;
    };
    std::map<std::string, std::set<std::string>> model; // tenant -> topics
    std::map<std::pair<std::string, std::string>, std::pair<uint64_t, uint32_t>> stamp; // (tenant, topic) -> (HLC timestamp, expirySeconds) of its LAST add
    RetainIndexHost h;
    uint64_t checks = 0;
    for (int round = 0; round < rounds; round++) {
        const bool full = round == 0 || rnd(8) == 0;
        bool ok;
        if (full) {
            model.clear();
            stamp.clear();
            std::vector<RetainIndexHost::Item> items;
            const size_t n = rnd(3) == 0 ? 0 : 1 + rnd(2500);
            for (size_t i = 0; i < n; i++) {
                const std::string& tn = tenants[rnd(tenants.size())];
                const std::string tp = rand_topic();
                model[tn].insert(tp);
                RetainIndexHost::Item it;
                it.tenant = tn;
                it.topic = tp;
                it.has_ts = rnd(4) != 0;
                it.ts = (uint64_t)(1000 + rnd(100000)) << 16 | rnd(65536);
                it.expiry = (uint32_t)rnd(500);
                stamp[{tn, tp}] = it.has_ts ? std::make_pair(it.ts, it.expiry) : std::make_pair<uint64_t, uint32_t>(0, 0xFFFFFFFFu);
                items.push_back(std::move(it));
            }
            ok = h.rebuild(std::move(items));
        } else {
            const std::string& tn = tenants[rnd(tenants.size())];
            std::vector<RetainIndexHost::Op> ops;
            const size_t n = 1 + rnd(rnd(4) == 0 ? 1500 : 40);
            auto& set = model[tn];
            for (size_t i = 0; i < n; i++) {
                if (!set.empty() && rnd(2)) {
                    auto it = set.begin();
                    std::advance(it, rnd(std::min<size_t>(set.size(), 300)));
                    const std::string tp = rnd(20) ? *it : rand_topic();
                    RetainIndexHost::Op o;
                    o.topic = tp;
                    o.op = 1;
                    ops.push_back(std::move(o));
                    set.erase(tp);
                    stamp.erase({tn, tp});
                } else {
                    const std::string tp = rand_topic();
                    RetainIndexHost::Op o;
                    o.topic = tp;
                    o.op = 0;
                    o.has_ts = rnd(4) != 0;
                    o.ts = (uint64_t)(1000 + rnd(100000)) << 16 | rnd(65536);
                    o.expiry = (uint32_t)rnd(500);
                    stamp[{tn, tp}] = o.has_ts ? std::make_pair(o.ts, o.expiry) : std::make_pair<uint64_t, uint32_t>(0, 0xFFFFFFFFu);
                    ops.push_back(std::move(o));
                    set.insert(tp);
                }
            }
            if (set.empty()) model.erase(tn);
            ok = h.apply(tn, std::move(ops));
        }
        if (!ok) {
            fprintf(stderr, "round %d: %s failed: %s\n", round, full ? "rebuild" : "apply", h.error.c_str());
            return 1;
        }
        // ids enumerate (tenant, level list) in order
        size_t total = 0;
        for (auto& e : model) total += e.second.size();
        if (h.n_topics != total) {
            fprintf(stderr, "round %d: n_topics %llu != %zu\n", round, (unsigned long long)h.n_topics, total);
            return 1;
        }
        std::vector<std::pair<std::string, std::vector<std::string>>> prev;
        for (uint32_t id = 0; id < total; id++) {
            std::string_view tn, tp;
            uint64_t ts = 0;
            uint32_t ex = 0;
            bool good = h.topic(id, tn, tp, &ts, &ex) && model.count(std::string(tn)) && model[std::string(tn)].count(std::string(tp));
            if (good) { // the stamp of the last add, and the expiry instant derived from it (RS/RetainStoreCoProc.java:298-304)
                const auto st = stamp[{std::string(tn), std::string(tp)}];
                good = st.first == ts && st.second == ex &&
                       h.expire_at[id] == ((ts == 0 && ex == 0xFFFFFFFFu) ? RETAIN_NEVER : (ts >> 16) + (uint64_t)ex * 1000);
            }
            if (!good) {
                fprintf(stderr, "round %d: topic(%u) is not in the model\n", round, id);
                return 1;
            }
            std::pair<std::string, std::vector<std::string>> key{std::string(tn), split(tp, '/')};
            if (!prev.empty() && !(prev.back() < key)) {
                fprintf(stderr, "round %d: ids are not in (tenant, level list) order at %u\n", round, id);
                return 1;
            }
            prev.clear();
            prev.push_back(std::move(key));
        }
        for (int q = 0; q < 120; q++) {
            const std::string tn = q % 25 == 24 ? std::string("nobody") : tenants[rnd(tenants.size())];
            const std::string filter = rand_filter();
            const auto fl = split(filter, '/');
            std::set<std::string> want;
            if (model.count(tn))
                for (auto& tp : model[tn])
                    if (filter_matches(fl, split(tp, '/'))) want.insert(tp);
            std::set<std::string> got;
            const auto ids = image_match(h, tn, filter);
            for (uint32_t id : ids) {
                std::string_view t2, tp;
                if (!h.topic(id, t2, tp) || t2 != tn || !got.insert(std::string(tp)).second) {
                    fprintf(stderr, "round %d: filter '%s': id %u is foreign or repeated\n", round, filter.c_str(), id);
                    return 1;
                }
            }
            checks++;
            if (got != want) {
                fprintf(stderr, "round %d (%s): tenant '%s' filter '%s': image gives %zu topics, the rule %zu\n", round, full ? "rebuild" : "apply", tn.c_str(),
                        filter.c_str(), got.size(), want.size());
                return 1;
            }
        }
    }
    printf("retain_fuzz ok: seed %llu, %d rounds, %llu filter checks\n", (unsigned long long)seed, rounds, (unsigned long long)checks);
    return 0;
}
