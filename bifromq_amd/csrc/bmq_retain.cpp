// bmq_retain.cpp -- host builder of the retained-topic index (see bmq_retain.h): one trie per tenant.
#include "bmq_retain.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

namespace bmq {

namespace {
// order of level lists: compare topics with '/' ranking below every other byte (== comparing level by level, a shorter
// list first) -- the order that makes subtrees contiguous id ranges
inline int cmp_topic(std::string_view a, std::string_view b) {
    const size_t n = std::min(a.size(), b.size());
    for (size_t i = 0; i < n; i++) {
        const int ca = a[i] == '/' ? -1 : (unsigned char)a[i], cb = b[i] == '/' ? -1 : (unsigned char)b[i];
        if (ca != cb) return ca < cb ? -1 : 1;
    }
    return a.size() < b.size() ? -1 : (a.size() > b.size() ? 1 : 0);
}
inline bool topic_less(const std::string& a, const std::string& b) { return cmp_topic(a, b) < 0; }

struct LocalBuild { // phase 1 result: a tenant's trie with tenant-local tokens
    RTenantState* st = nullptr;
    struct PNode {
        uint32_t parent, token, first_child, n_children, term, sub_begin, sub_end, bfs;
        bool sys;
    };
    std::vector<PNode> pn;          // preorder
    std::vector<uint32_t> order;    // bfs index -> preorder id
    std::vector<uint32_t> next_after;
    std::vector<std::string_view> local_strings;
    std::vector<uint32_t> l2g;
};

void build_local(LocalBuild& b) {
    RTenantState& t = *b.st;
    const size_t n = t.topics.size();
    HostDict local;
    auto& pn = b.pn;
    pn.clear();
    pn.reserve(n * 2 + 4);
    pn.push_back({NONE, 0, NONE, 0, 0, 0, 0, 0, false}); // root (label = tenant id, filled in globally)
    std::vector<uint32_t> stack{0};
    std::vector<std::string_view> prev_levels, levels;
    for (size_t i = 0; i < n; i++) {
        const std::string_view topic = t.topics[i];
        levels.clear();
        for (size_t s = 0, k = 0; k <= topic.size(); k++)
            if (k == topic.size() || topic[k] == '/') {
                levels.push_back(topic.substr(s, k - s));
                s = k + 1;
            }
        size_t reuse = 0;
        while (reuse < levels.size() && reuse < prev_levels.size() && levels[reuse] == prev_levels[reuse]) reuse++;
        stack.resize(reuse + 1);
        prev_levels.resize(reuse);
        for (size_t l = reuse; l < levels.size(); l++) {
            const uint32_t parent = stack.back();
            const uint32_t id = (uint32_t)pn.size();
            pn.push_back({parent, local.intern(levels[l]), NONE, 0, 0, (uint32_t)i, 0, 0, !levels[l].empty() && levels[l][0] == '$'});
            if (pn[parent].first_child == NONE) pn[parent].first_child = id;
            pn[parent].n_children++;
            stack.push_back(id);
            prev_levels.push_back(levels[l]);
        }
        pn[stack.back()].term = 1; // sorted unique input: each topic ends at a distinct node
    }
    // subtree id ranges + "next node after the subtree" (= next sibling), one stack pass over the preorder
    b.next_after.assign(pn.size(), (uint32_t)pn.size());
    {
        std::vector<uint32_t> st;
        for (uint32_t v = 0; v < pn.size(); v++) {
            while (!st.empty() && st.back() != pn[v].parent) {
                pn[st.back()].sub_end = pn[v].sub_begin;
                b.next_after[st.back()] = v;
                st.pop_back();
            }
            st.push_back(v);
        }
        while (!st.empty()) {
            pn[st.back()].sub_end = (uint32_t)n;
            st.pop_back();
        }
    }
    // breadth-first numbering; children of a node keep their (sorted) order
    b.order.clear();
    b.order.reserve(pn.size());
    b.order.push_back(0);
    for (size_t q = 0; q < b.order.size(); q++) {
        const uint32_t v = b.order[q];
        uint32_t c = pn[v].first_child;
        for (uint32_t k = 0; k < pn[v].n_children; k++) {
            b.order.push_back(c);
            c = b.next_after[c];
        }
    }
    for (uint32_t i = 0; i < b.order.size(); i++) pn[b.order[i]].bfs = i;
    b.local_strings.resize(local.entries.size());
    for (size_t i = 0; i < local.entries.size(); i++) b.local_strings[i] = local.entries[i].s;
}

void place_local(LocalBuild& b) { // phase 3: nodes + edge hash with global tokens
    RTenantState& t = *b.st;
    auto& pn = b.pn;
    const size_t nn = b.order.size();
    t.nodes.assign(nn, RNode{0, 0, 0, 0});
    const uint32_t eslots = pow2_at_least(std::max<uint64_t>(8, (uint64_t)nn * 2));
    const uint32_t bmask = eslots / 4 - 1;
    t.edges.assign(eslots, REdge{NONE, 0, 0, 0});
    uint32_t running = 1; // child_begin is monotone also over childless nodes: the children of the node range [a, b)
    for (uint32_t i = 0; i < nn; i++) { // are [child_begin(a), child_begin(b-1) + count(b-1))
        const auto& p = pn[b.order[i]];
        RNode& r = t.nodes[i];
        r.child_begin = running;
        running += p.n_children;
        r.child_count = p.n_children | (p.term ? RN_TERM : 0u);
        r.sub_begin = p.sub_begin;
        r.sub_end = p.sub_end;
        if (p.parent != NONE) {
            const uint32_t pb = pn[p.parent].bfs, tok = b.l2g[p.token - TOK_FIRST];
            const uint32_t home = redge_bucket(pb, tok, bmask);
            uint32_t bk = home, s = NONE;
            for (;;) {
                for (uint32_t j = 0; j < 4 && s == NONE; j++)
                    if (t.edges[4 * bk + j].parent == NONE) s = 4 * bk + j;
                if (s != NONE) break;
                bk = (bk + 1) & bmask;
            }
            t.edges[s] = REdge{pb, tok, i, p.sub_begin | (p.term ? RN_TERM : 0u)};
            if (bk != home) t.edges[4 * home].child |= RE_OVERFLOW; // (the home bucket is full: its first entry exists)
        }
    }
    // grandchild postings of the nodes with many children (RGp): (parent, token, child, child_topic) by (grandparent, token, parent)
    {
        struct PE {
            uint32_t gp, tok, parent, child, topic;
        };
        std::vector<PE> pe;
        for (uint32_t i = 1; i < nn; i++) {
            const auto& p = pn[b.order[i]];
            if (pn[p.parent].parent == NONE) continue; // a child of the root has no grandparent
            const auto& g = pn[pn[p.parent].parent];
            if (g.n_children < RPOST_MIN) continue;
            pe.push_back({g.bfs, b.l2g[p.token - TOK_FIRST], pn[p.parent].bfs, i, p.sub_begin | (p.term ? RN_TERM : 0u)});
        }
        std::sort(pe.begin(), pe.end(), [](const PE& x, const PE& y) {
            return x.gp != y.gp ? x.gp < y.gp : x.tok != y.tok ? x.tok < y.tok : x.parent < y.parent;
        });
        t.posts.resize(pe.size());
        size_t runs = 0;
        for (size_t i = 0; i < pe.size(); i++) {
            t.posts[i] = REdge{pe[i].parent, pe[i].tok, pe[i].child, pe[i].topic};
            if (i == 0 || pe[i].gp != pe[i - 1].gp || pe[i].tok != pe[i - 1].tok) runs++;
        }
        const uint32_t gslots = pow2_at_least(std::max<uint64_t>(4, (uint64_t)runs * 2));
        const uint32_t gmask = gslots / 4 - 1;
        t.gps.assign(gslots, RGp{NONE, 0, 0, 0});
        for (size_t i = 0; i < pe.size();) {
            size_t j = i;
            while (j < pe.size() && pe[j].gp == pe[i].gp && pe[j].tok == pe[i].tok) j++;
            const uint32_t home = rgp_bucket(pe[i].gp, pe[i].tok, gmask);
            uint32_t bk = home, sl = NONE;
            for (;;) {
                for (uint32_t k = 0; k < 4 && sl == NONE; k++)
                    if (t.gps[4 * bk + k].gp == NONE) sl = 4 * bk + k;
                if (sl != NONE) break;
                bk = (bk + 1) & gmask;
            }
            t.gps[sl] = RGp{pe[i].gp, pe[i].tok, (uint32_t)i, (uint32_t)(j - i)};
            if (bk != home) t.gps[4 * home].count |= RE_OVERFLOW;
            i = j;
        }
    }
    // the '$' children of the root form one contiguous run (children are sorted by label bytes)
    t.sys_node_lo = t.sys_node_hi = t.sys_id_lo = t.sys_id_hi = 0;
    uint32_t c = pn[0].first_child;
    bool in_run = false;
    for (uint32_t k = 0; k < pn[0].n_children; k++) {
        if (pn[c].sys) {
            if (!in_run) {
                t.sys_node_lo = pn[c].bfs;
                t.sys_id_lo = pn[c].sub_begin;
                in_run = true;
            }
            t.sys_node_hi = pn[c].bfs + 1;
            t.sys_id_hi = pn[c].sub_end;
        }
        c = b.next_after[c];
    }
}

template <class F> void parallel_for(size_t n, F&& f) {
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned nth = (unsigned)std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 64), std::max<size_t>(n, 1));
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) break;
            f(i);
        }
    };
    std::vector<std::thread> th;
    for (unsigned w = 1; w < nth; w++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
}
} // namespace

bool RetainIndexHost::rebuild(std::vector<Item>&& items) {
    error.clear();
    by_name.clear();
    order.clear();
    nodes.clear();
    edges.clear();
    dict_h = HostDict();
    strings.clear();
    node_free = edge_free = post_free = gp_free = 0;
    posts.clear();
    gps.clear();
    full_upload = dict_changed = true;
    dirty.clear();
    std::vector<RTenantState*> touched;
    std::map<RTenantState*, std::vector<Item*>> per;
    for (auto& it : items) {
        auto f = by_name.find(it.tenant);
        if (f == by_name.end()) {
            auto st = std::make_unique<RTenantState>();
            st->name = it.tenant;
            f = by_name.emplace(it.tenant, std::move(st)).first;
            touched.push_back(f->second.get());
        }
        per[f->second.get()].push_back(&it);
    }
    std::vector<std::vector<Item*>*> lists;
    for (RTenantState* t : touched) lists.push_back(&per[t]);
    parallel_for(touched.size(), [&](size_t i) {
        auto& v = *lists[i];
        std::stable_sort(v.begin(), v.end(), [](const Item* a, const Item* b) { return topic_less(a->topic, b->topic); });
        RTenantState& t = *touched[i];
        for (size_t k = 0; k < v.size(); k++) {
            if (k + 1 < v.size() && v[k + 1]->topic == v[k]->topic) continue; // the last add of a topic wins
            t.topics.push_back(std::move(v[k]->topic));
            t.ts.push_back(v[k]->has_ts ? v[k]->ts : 0);
            t.expiry.push_back(v[k]->has_ts ? v[k]->expiry : 0xFFFFFFFFu);
        }
    });
    return refresh(touched);
}

bool RetainIndexHost::refresh(std::vector<RTenantState*>& touched) {
    std::vector<RTenantState*> live;
    for (RTenantState* t : touched) {
        if (t->topics.empty()) {
            const std::string name = t->name;
            by_name.erase(name);
        } else live.push_back(t);
    }
    std::vector<LocalBuild> builds(live.size());
    for (size_t i = 0; i < live.size(); i++) builds[i].st = live[i];
    parallel_for(builds.size(), [&](size_t i) { build_local(builds[i]); });
    const size_t tokens_before = dict_h.entries.size();
    auto intern_owned = [&](std::string_view s) -> uint32_t {
        const size_t before = dict_h.entries.size();
        const uint32_t tok = dict_h.intern(s);
        if (dict_h.entries.size() != before) {
            strings.emplace_back(s);
            dict_h.entries.back().s = strings.back();
        }
        return tok;
    };
    parallel_for(builds.size(), [&](size_t bi) {
        LocalBuild& b = builds[bi];
        b.l2g.resize(b.local_strings.size());
        for (size_t i = 0; i < b.local_strings.size(); i++) b.l2g[i] = dict_h.find(b.local_strings[i]);
    });
    for (auto& b : builds) {
        b.st->token = intern_owned(b.st->name);
        for (size_t i = 0; i < b.local_strings.size(); i++)
            if (b.l2g[i] == TOK_UNKNOWN) b.l2g[i] = intern_owned(b.local_strings[i]);
    }
    if (dict_h.entries.size() != tokens_before) dict_changed = true;
    parallel_for(builds.size(), [&](size_t i) { place_local(builds[i]); });
    // segment allocation in the global arrays (with head-room; a tenant that outgrows its segment moves to the end)
    for (auto& b : builds) {
        RTenantState& t = *b.st;
        if (t.nodes.size() > t.node_cap) {
            t.node_base = node_free;
            t.node_cap = (uint32_t)(t.nodes.size() + t.nodes.size() / 4 + 4);
            node_free += t.node_cap;
        }
        if (t.edges.size() > t.edge_cap) {
            t.edge_base = edge_free;
            t.edge_cap = (uint32_t)t.edges.size();
            edge_free += t.edge_cap;
        }
        if (t.posts.size() > t.post_cap) {
            t.post_base = post_free;
            t.post_cap = (uint32_t)t.posts.size();
            post_free += t.post_cap;
        }
        if (t.gps.size() > t.gp_cap) {
            t.gp_base = gp_free;
            t.gp_cap = (uint32_t)t.gps.size();
            gp_free += t.gp_cap;
        }
    }
    if (node_free > nodes.size() || edge_free > edges.size() || post_free > posts.size() || gp_free > gps.size()) {
        nodes.resize(std::max<size_t>((size_t)node_free + node_free / 4, 16));
        edges.resize(std::max<size_t>((size_t)edge_free + edge_free / 4, 16), REdge{NONE, 0, 0, 0});
        posts.resize(std::max<size_t>((size_t)post_free + post_free / 4, 16), REdge{NONE, 0, 0, 0});
        gps.resize(std::max<size_t>((size_t)gp_free + gp_free / 4, 16), RGp{NONE, 0, 0, 0});
        full_upload = true;
        dirty.clear();
    }
    for (auto& b : builds) {
        RTenantState& t = *b.st;
        std::copy(t.nodes.begin(), t.nodes.end(), nodes.begin() + t.node_base);
        std::copy(t.edges.begin(), t.edges.end(), edges.begin() + t.edge_base);
        std::copy(t.posts.begin(), t.posts.end(), posts.begin() + t.post_base);
        std::copy(t.gps.begin(), t.gps.end(), gps.begin() + t.gp_base);
        if (!full_upload) dirty.push_back(&t);
    }
    order.clear();
    uint64_t id = 0;
    for (auto& e : by_name) {
        e.second->id_base = (uint32_t)id;
        id += e.second->topics.size();
        order.push_back(e.second.get());
    }
    if (id >= 0x7FFFFFF0ull) {
        error = "too many retained topics";
        return false;
    }
    n_topics = id;
    expire_at.assign(n_topics ? n_topics : 1, RETAIN_NEVER);
    for (RTenantState* t : order)
        for (size_t i = 0; i < t->topics.size(); i++)
            expire_at[t->id_base + i] = (t->ts[i] == 0 && t->expiry[i] == 0xFFFFFFFFu) ? RETAIN_NEVER : retain_expire_at(t->ts[i], t->expiry[i]);
    const uint32_t tslots = pow2_at_least((uint64_t)order.size() * 2);
    tenants.assign(tslots, RTenantSlot{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0}});
    for (RTenantState* t : order) {
        uint32_t d = tenant_hash(t->token) & (tslots - 1);
        while (tenants[d].token) d = (d + 1) & (tslots - 1);
        tenants[d] = RTenantSlot{t->token, t->node_base, t->edge_base, (uint32_t)t->edges.size() / 4 - 1, t->id_base,
                                 t->sys_node_lo, t->sys_node_hi, t->sys_id_lo, t->sys_id_hi, t->post_base, t->gp_base, (uint32_t)t->gps.size() / 4 - 1,
                                 {0, 0, 0, 0}};
    }
    if (dict_changed) flatten_dict(dict_h, dict, pool);
    if (nodes.empty()) nodes.resize(16);
    if (edges.empty()) edges.resize(16, REdge{NONE, 0, 0, 0});
    if (posts.empty()) posts.resize(16, REdge{NONE, 0, 0, 0});
    if (gps.empty()) gps.resize(16, RGp{NONE, 0, 0, 0});
    return true;
}

bool RetainIndexHost::topic(uint32_t id, std::string_view& tenant, std::string_view& topic, uint64_t* ts, uint32_t* expiry) const {
    if (id >= n_topics || order.empty()) return false;
    size_t lo = 0, hi = order.size();
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (order[mid]->id_base <= id) lo = mid;
        else hi = mid;
    }
    tenant = order[lo]->name;
    topic = order[lo]->topics[id - order[lo]->id_base];
    if (ts) *ts = order[lo]->ts[id - order[lo]->id_base];
    if (expiry) *expiry = order[lo]->expiry[id - order[lo]->id_base];
    return true;
}

} // namespace bmq
