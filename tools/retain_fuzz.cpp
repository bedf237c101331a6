// retain_fuzz.cpp -- test tool, not product: the retain index without a GPU, under AddressSanitizer/UBSan and ThreadSanitizer.
//   * the bulk load (bmq_retain.cpp: RetainIndexHost::rebuild): ids must enumerate (tenant, topic) in (tenant bytes, level-list bytes)
//     order; a CPU walk over the HBM image (directory, breadth-first node array, edge hash, '$' runs) done the way k_retain_walk
//     does it -- frontier of node ranges, '+' = children range, '#' = subtree id ranges -- must return exactly the topics the rule
//     of SURVEY.md 8a-0 selects by brute force;
//   * the MUTATION path (bmq_retain_core.h through RetainDyn<HostExec>, bmq_retain_dyn.h): the very functions the gfx950 kernels
//     k_r_locate / k_r_commit / k_r_rank run, here on several host threads with minimal capacities (every growth path runs all the
//     time).  After every batch: ids are stable (an untouched topic keeps its id, a re-added one gets its id back, a new one a fresh
//     id), stamps are those of the last add, the image walk (dead ids dropped) + a walk over the overlay trie done the way the kernel
//     does it equal the brute force, the GC scan and the live-id listing equal the model, overlay ids resolve to their strings.
// Build + run: make -C bifromq_amd/csrc fuzz   (tests/test_host.py runs a short round of both builds)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../bifromq_amd/csrc/bmq_exec_host.h"
#include "../bifromq_amd/csrc/bmq_retain.h"
#include "../bifromq_amd/csrc/bmq_retain_dyn.h"

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
                                    nxt.push_back({e[j].child & ~RE_OVERFLOW, 1});
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

// the overlay trie walked the way k_retain_walk does it: frontier of single nodes, '+' / '#' over child lists, literal levels by hash
static std::vector<uint32_t> overlay_match(const RetainDynView& dv, std::string_view tenant, std::string_view filter) {
    std::vector<uint32_t> ids;
    if (!dv.ov_live) return ids;
    auto find = [&](uint32_t parent, std::string_view label) {
        const LevelHash h = hash_level(label);
        std::string padded(label);
        padded.append(16, '\0');
        return ov_find(dv.onodes, dv.oedges, dv.oedge_mask, dv.opool, parent, h.h1, h.h2, (uint32_t)label.size(), (const uint8_t*)padded.data(), 0);
    };
    auto live = [&](uint32_t node) {
        const uint32_t id = dv.onodes[node].topic_id;
        if (id != NONE && !id_dead(dv.dead_bits, id)) ids.push_back(id);
    };
    const uint32_t tnode = find(0, tenant);
    if (tnode == NONE) return ids;
    const auto levels = split(filter, '/');
    std::vector<uint32_t> cur{tnode}, nxt;
    for (size_t li = 0; li < levels.size() && !cur.empty(); li++) {
        const std::string& lv = levels[li];
        if (lv == "#" && li + 1 == levels.size()) {
            bool first = true;
            while (!cur.empty()) {
                nxt.clear();
                for (uint32_t n : cur) {
                    if (!(first && li == 0)) live(n);
                    for (uint32_t c = dv.onodes[n].first_child; c != NONE; c = dv.onodes[c].next_sibling)
                        if (!(first && li == 0 && (dv.onodes[c].str_len & ON_SYS))) nxt.push_back(c);
                }
                cur.swap(nxt);
                first = false;
            }
            std::sort(ids.begin(), ids.end());
            return ids;
        }
        nxt.clear();
        for (uint32_t n : cur) {
            if (lv == "+") {
                for (uint32_t c = dv.onodes[n].first_child; c != NONE; c = dv.onodes[c].next_sibling)
                    if (!(li == 0 && (dv.onodes[c].str_len & ON_SYS))) nxt.push_back(c);
            } else {
                const uint32_t c = find(n, lv);
                if (c != NONE) nxt.push_back(c);
            }
        }
        cur.swap(nxt);
    }
    for (uint32_t n : cur) live(n);
    std::sort(ids.begin(), ids.end());
    return ids;
}

#define FAIL(...)                     \
    do {                              \
        fprintf(stderr, __VA_ARGS__); \
        return 1;                     \
    } while (0)

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 30;
    const unsigned threads = argc > 3 ? (unsigned)atoi(argv[3]) : 4;
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
    using Key = std::pair<std::string, std::string>; // (tenant, topic)
    std::map<Key, uint32_t> live;                    // retained now -> id
    std::map<Key, uint32_t> ever;                    // every topic that got an id in this generation -> id (ids are never reused)
    std::map<Key, std::pair<uint64_t, uint32_t>> stamp; // (HLC timestamp, expirySeconds) of the LAST add
    RetainIndexHost h;
    HostExec hx;
    hx.threads = threads;
    RetainDyn<HostExec> rt(hx);
    rt.tiny = true;
    RetainIndexView bview{};
    uint64_t checks = 0, n_ops = 0, n_batches = 0;
    auto base_view = [&]() {
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
        return v;
    };
    for (int round = 0; round < rounds; round++) {
        const bool full = round == 0 || rnd(8) == 0;
        if (full) { // a bulk load: either fresh random content, or a COMPACTION of what is live (what bmq_retain_compact does)
            const bool compact = round != 0 && rnd(2);
            std::vector<RetainIndexHost::Item> items;
            if (compact) {
                for (auto& kv : live) {
                    RetainIndexHost::Item it;
                    it.tenant = kv.first.first;
                    it.topic = kv.first.second;
                    it.ts = stamp[kv.first].first;
                    it.expiry = stamp[kv.first].second;
                    it.has_ts = !(it.ts == 0 && it.expiry == 0xFFFFFFFFu);
                    items.push_back(std::move(it));
                }
            } else {
                live.clear();
                stamp.clear();
                const size_t n = rnd(3) == 0 ? 0 : 1 + rnd(2500);
                for (size_t i = 0; i < n; i++) {
                    RetainIndexHost::Item it;
                    it.tenant = tenants[rnd(tenants.size())];
                    it.topic = rand_topic();
                    it.has_ts = rnd(4) != 0;
                    it.ts = (uint64_t)(1000 + rnd(100000)) << 16 | rnd(65536);
                    it.expiry = (uint32_t)rnd(500);
                    stamp[{it.tenant, it.topic}] = it.has_ts ? std::make_pair(it.ts, it.expiry) : std::make_pair<uint64_t, uint32_t>(0, 0xFFFFFFFFu);
                    live[{it.tenant, it.topic}] = 0;
                    items.push_back(std::move(it));
                }
            }
            if (!h.rebuild(std::move(items))) FAIL("round %d: rebuild failed: %s\n", round, h.error.c_str());
            bview = base_view();
            if (!rt.reset(bview, h)) FAIL("round %d: reset failed: %s\n", round, rt.error.c_str());
            // ids enumerate (tenant, level list) in order
            if (h.n_topics != live.size()) FAIL("round %d: n_topics %llu != %zu\n", round, (unsigned long long)h.n_topics, live.size());
            ever.clear();
            std::pair<std::string, std::vector<std::string>> prev;
            for (uint32_t id = 0; id < h.n_topics; id++) {
                std::string_view tn, tp;
                if (!h.topic(id, tn, tp) || !live.count({std::string(tn), std::string(tp)})) FAIL("round %d: topic(%u) is not in the model\n", round, id);
                std::pair<std::string, std::vector<std::string>> key{std::string(tn), split(tp, '/')};
                if (id && !(prev < key)) FAIL("round %d: ids are not in (tenant, level list) order at %u\n", round, id);
                prev = std::move(key);
                live[{std::string(tn), std::string(tp)}] = id;
                ever[{std::string(tn), std::string(tp)}] = id;
            }
        } else { // one batch of adds / removes over several tenants, duplicates inside the batch included
            std::vector<std::string> tn_list;
            std::string tbytes, pbytes;
            std::vector<uint32_t> toff{0}, poff{0}, op_tenant;
            std::vector<uint8_t> op;
            std::vector<unsigned long long> ts;
            std::vector<uint32_t> ex;
            const size_t n_ten = 1 + rnd(tenants.size());
            for (size_t t = 0; t < n_ten; t++) {
                tbytes += tenants[(t + round) % tenants.size()];
                toff.push_back((uint32_t)tbytes.size());
                tn_list.push_back(tenants[(t + round) % tenants.size()]);
            }
            const size_t n = 1 + rnd(rnd(4) == 0 ? 1500 : 40);
            const bool with_ts = rnd(5) != 0, single = n_ten == 1 && rnd(2);
            std::vector<Key> keys;
            for (size_t i = 0; i < n; i++) {
                const uint32_t ti = (uint32_t)rnd(n_ten);
                std::string tp;
                uint8_t o;
                if (!live.empty() && rnd(2)) { // mostly removals of live topics (sometimes of another tenant's name or a random topic)
                    auto it = live.begin();
                    std::advance(it, rnd(std::min<size_t>(live.size(), 300)));
                    tp = rnd(20) ? it->first.second : rand_topic();
                    o = 1;
                } else if (!ever.empty() && rnd(4) == 0) { // re-add something that had an id before
                    auto it = ever.begin();
                    std::advance(it, rnd(std::min<size_t>(ever.size(), 300)));
                    tp = it->first.second;
                    o = 0;
                } else {
                    tp = rand_topic();
                    o = 0;
                }
                pbytes += tp;
                poff.push_back((uint32_t)pbytes.size());
                op_tenant.push_back(ti);
                op.push_back(o);
                ts.push_back((uint64_t)(1000 + rnd(100000)) << 16 | rnd(65536));
                ex.push_back((uint32_t)rnd(500));
                keys.push_back({tn_list[ti], tp});
            }
            pbytes.append(16, '\0');
            tbytes.append(16, '\0');
            std::vector<uint32_t> out(n);
            if (!rt.apply((const uint8_t*)tbytes.data(), toff.data(), (uint32_t)n_ten, single ? nullptr : op_tenant.data(), (const uint8_t*)pbytes.data(), poff.data(), op.data(),
                          with_ts ? ts.data() : nullptr, with_ts ? ex.data() : nullptr, (uint32_t)n, out.data()))
                FAIL("round %d: apply failed: %s\n", round, rt.error.c_str());
            n_ops += n;
            n_batches++;
            // the model: ops in order; the LAST op on a topic decides and is the one that reports the id
            std::map<Key, size_t> last;
            for (size_t i = 0; i < n; i++) {
                if (single) keys[i].first = tn_list[0];
                last[keys[i]] = i;
            }
            for (size_t i = 0; i < n; i++) {
                const Key& k = keys[i];
                if (last[k] != i) {
                    if (out[i] != NONE) FAIL("round %d: op %zu (%d '%s' '%s') was superseded by op %zu (%d) but reports id %u; n=%zu single=%d\n", round, i, op[i], k.first.c_str(), k.second.c_str(), last[k], op[last[k]], out[i], n, (int)single);
                    continue;
                }
                if (op[i] == 0) {
                    if (out[i] == NONE) FAIL("round %d: add %zu reports no id\n", round, i);
                    auto ev = ever.find(k);
                    if (ev != ever.end() && ev->second != out[i]) FAIL("round %d: re-added topic changed its id %u -> %u\n", round, ev->second, out[i]);
                    if (ev == ever.end()) {
                        for (auto& kv : ever)
                            if (kv.second == out[i]) FAIL("round %d: id %u handed out twice\n", round, out[i]);
                        if (out[i] < h.n_topics) FAIL("round %d: a new topic got a bulk-loaded id %u\n", round, out[i]);
                    }
                    ever[k] = out[i];
                    live[k] = out[i];
                    stamp[k] = with_ts ? std::make_pair((uint64_t)ts[i], ex[i]) : std::make_pair<uint64_t, uint32_t>(0, 0xFFFFFFFFu);
                } else {
                    auto ev = ever.find(k);
                    if (ev == ever.end()) {
                        if (out[i] != NONE) FAIL("round %d: removal of a topic that never existed reports id %u\n", round, out[i]);
                    } else if (out[i] != ev->second) FAIL("round %d: removal reports id %u, the topic's id is %u\n", round, out[i], ev->second);
                    live.erase(k);
                }
            }
        }
        // ---- after the step: counters, listing, stamps, strings -----------------------------------------------------------------------
        if (rt.info.n_live != live.size()) FAIL("round %d: n_live %llu != %zu\n", round, (unsigned long long)rt.info.n_live, live.size());
        uint64_t base_dead = 0;
        for (auto& kv : ever) base_dead += kv.second < h.n_topics && !live.count(kv.first);
        if (rt.info.base_dead != base_dead) FAIL("round %d: base_dead %llu != %llu\n", round, (unsigned long long)rt.info.base_dead, (unsigned long long)base_dead);
        std::vector<uint32_t> ids, want_ids;
        GcQuery q{};
        q.live_only = 1;
        q.override_expiry = -1;
        if (!rt.select(q, nullptr, 0, ids)) FAIL("round %d: select failed\n", round);
        for (auto& kv : live) want_ids.push_back(kv.second);
        std::sort(want_ids.begin(), want_ids.end());
        if (ids != want_ids) FAIL("round %d: live ids differ (%zu vs %zu)\n", round, ids.size(), want_ids.size());
        {
            std::vector<uint32_t> ov;
            for (auto& kv : ever)
                if (kv.second >= h.n_topics) ov.push_back(kv.second);
            std::vector<uint32_t> lens;
            std::vector<uint8_t> bytes;
            if (!rt.overlay_topics(ov.data(), (uint32_t)ov.size(), lens, bytes)) FAIL("round %d: overlay_topics failed\n", round);
            size_t off = 0, k = 0;
            for (auto& kv : ever) {
                if (kv.second < h.n_topics) continue;
                const std::string got((const char*)bytes.data() + off, lens[2 * k + 1]);
                if (lens[2 * k] != kv.first.first.size() || got != kv.first.first + kv.first.second)
                    FAIL("round %d: overlay id %u resolves to '%s', want '%s%s'\n", round, kv.second, got.c_str(), kv.first.first.c_str(), kv.first.second.c_str());
                off += lens[2 * k + 1];
                k++;
            }
        }
        for (int s = 0; s < 40 && !live.empty(); s++) { // stamps of the last add
            auto it = live.begin();
            std::advance(it, rnd(live.size()));
            unsigned long long ts = 0, at = 0;
            uint32_t ex = 0;
            bool lv = false;
            if (!rt.topic_info(it->second, ts, ex, at, lv) || !lv) FAIL("round %d: topic_info(%u) fails for a live topic\n", round, it->second);
            const auto st = stamp[it->first];
            if (st.first != ts || st.second != ex || at != ((ts == 0 && ex == 0xFFFFFFFFu) ? RETAIN_NEVER : (ts >> 16) + (uint64_t)ex * 1000))
                FAIL("round %d: stamp of id %u differs\n", round, it->second);
        }
        // the GC scan: one tenant (its '$' topics are out of reach of match(tenant, "#")) and all tenants, with and without an override
        for (int s = 0; s < 4; s++) {
            const uint64_t now = 1000 + rnd(160000);
            const long long over = s % 2 ? (long long)rnd(300) : -1;
            const bool all = s >= 2;
            const std::string tn = tenants[rnd(tenants.size())];
            GcQuery g{};
            g.now = now;
            g.override_expiry = over;
            g.skip_sys = 1;
            if (!all) {
                g.has_tenant = 1;
                auto f = h.by_name.find(tn);
                if (f != h.by_name.end()) {
                    g.t_lo = f->second->id_base, g.t_hi = f->second->id_base + (uint32_t)f->second->topics.size();
                    g.sys_lo = f->second->id_base + f->second->sys_id_lo, g.sys_hi = f->second->id_base + f->second->sys_id_hi;
                }
                g.t_node = NONE;
            }
            std::vector<uint32_t> got, want;
            if (!rt.select(g, all ? nullptr : (const uint8_t*)tn.data(), (uint32_t)tn.size(), got)) FAIL("round %d: gc select failed\n", round);
            for (auto& kv : live) {
                if (!all && (kv.first.first != tn || (!kv.first.second.empty() && kv.first.second[0] == '$'))) continue;
                const auto st = stamp[kv.first];
                uint64_t at = (st.first == 0 && st.second == 0xFFFFFFFFu) ? RETAIN_NEVER : (st.first >> 16) + (uint64_t)st.second * 1000;
                if (over >= 0) at = (st.first >> 16) + (uint64_t)over * 1000;
                if (at <= now) want.push_back(kv.second);
            }
            std::sort(want.begin(), want.end());
            if (got != want) FAIL("round %d: GC scan (%s, override %lld) gives %zu ids, the model %zu\n", round, all ? "all" : tn.c_str(), over, got.size(), want.size());
        }
        // ---- match: image walk (dead ids dropped) + overlay walk == brute force ----------------------------------------------------------
        RetainDynView dv = rt.view();
        for (int qn = 0; qn < 120; qn++) {
            const std::string tn = qn % 25 == 24 ? std::string("nobody") : tenants[rnd(tenants.size())];
            const std::string filter = rand_filter();
            const auto fl = split(filter, '/');
            std::vector<uint32_t> want, got;
            for (auto& kv : live)
                if (kv.first.first == tn && filter_matches(fl, split(kv.first.second, '/'))) want.push_back(kv.second);
            std::sort(want.begin(), want.end());
            for (uint32_t id : image_match(h, tn, filter))
                if (!id_dead(dv.dead_bits, id)) got.push_back(id);
            const auto ov = overlay_match(dv, tn, filter);
            if (!ov.empty() && !got.empty() && ov.front() <= got.back()) FAIL("round %d: overlay ids are not above the bulk-loaded ones\n", round);
            got.insert(got.end(), ov.begin(), ov.end());
            checks++;
            if (got != want) FAIL("round %d: tenant '%s' filter '%s': index gives %zu topics, the rule %zu\n", round, tn.c_str(), filter.c_str(), got.size(), want.size());
            // the live count of every matched range comes from the rank directory
            if (dv.use_dead) {
                const auto all_ids = image_match(h, tn, filter);
                for (size_t a = 0; a < all_ids.size();) {
                    size_t b = a;
                    while (b + 1 < all_ids.size() && all_ids[b + 1] == all_ids[b] + 1) b++;
                    const uint32_t lo = all_ids[a], hi = all_ids[b] + 1;
                    uint32_t dead = 0;
                    for (uint32_t id = lo; id < hi; id++) dead += id_dead(dv.dead_bits, id);
                    if (dead_before(dv.dead_bits, dv.dead_rank, hi) - dead_before(dv.dead_bits, dv.dead_rank, lo) != dead) FAIL("round %d: rank directory is off in [%u, %u)\n", round, lo, hi);
                    a = b + 1;
                }
            }
        }
    }
    printf("retain_fuzz ok: seed %llu, %d rounds, %u threads, %llu ops in %llu batches, %llu filter checks\n", (unsigned long long)seed, rounds, threads,
           (unsigned long long)n_ops, (unsigned long long)n_batches, (unsigned long long)checks);
    return 0;
}
