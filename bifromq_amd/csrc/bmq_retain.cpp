// bmq_retain.cpp -- host builder of the retained-topic index (see bmq_retain.h).
#include "bmq_retain.h"

#include <algorithm>
#include <cstring>

#include "bmq_dict.h"

namespace bmq {

namespace {
// order of (tenant, level list): compare tenants bytewise, then topics with '/' ranking below every other byte
// (== comparing the level lists level by level, a shorter list first) -- the order that makes subtrees contiguous
inline int cmp_topic(std::string_view a, std::string_view b) {
    const size_t n = std::min(a.size(), b.size());
    for (size_t i = 0; i < n; i++) {
        const int ca = a[i] == '/' ? -1 : (unsigned char)a[i], cb = b[i] == '/' ? -1 : (unsigned char)b[i];
        if (ca != cb) return ca < cb ? -1 : 1;
    }
    return a.size() < b.size() ? -1 : (a.size() > b.size() ? 1 : 0);
}
} // namespace

void RetainIndexHost::assign(std::vector<std::pair<std::string, std::string>>&& items) {
    std::sort(items.begin(), items.end(), [](const auto& x, const auto& y) {
        if (x.first != y.first) return x.first < y.first;
        return cmp_topic(x.second, y.second) < 0;
    });
    items.erase(std::unique(items.begin(), items.end()), items.end());
    bytes.clear();
    off.assign(1, 0);
    tenant_len.clear();
    for (auto& it : items) {
        bytes.insert(bytes.end(), it.first.begin(), it.first.end());
        bytes.push_back(0);
        bytes.insert(bytes.end(), it.second.begin(), it.second.end());
        off.push_back(bytes.size());
        tenant_len.push_back((uint32_t)it.first.size());
    }
    if (bytes.empty()) bytes.push_back(0);
}

std::vector<std::pair<std::string, std::string>> RetainIndexHost::items() const {
    std::vector<std::pair<std::string, std::string>> v;
    v.reserve(size());
    for (size_t i = 0; i < size(); i++) v.emplace_back(std::string(tenant_of(i)), std::string(topic_of(i)));
    return v;
}

bool RetainIndexHost::build() {
    error.clear();
    const size_t n = size();
    if (n >= 0x7FFFFFF0ull) {
        error = "too many retained topics";
        return false;
    }
    struct PNode { // preorder construction
        uint32_t parent, token, first_child, n_children, term, sub_begin, sub_end, bfs;
        bool sys;
    };
    std::vector<PNode> pn;
    pn.reserve(n * 2 + 16);
    HostDict dict_h;
    std::vector<uint32_t> roots;
    std::vector<uint32_t> stack; // node per depth of the previous topic (stack[0] = tenant root)
    std::vector<std::string_view> prev_levels;
    std::string_view prev_tenant;
    bool have_prev = false;
    std::vector<std::string_view> levels;
    for (size_t i = 0; i < n; i++) {
        const std::string_view tenant = tenant_of(i), topic = topic_of(i);
        levels.clear();
        for (size_t s = 0, k = 0; k <= topic.size(); k++)
            if (k == topic.size() || topic[k] == '/') {
                levels.push_back(topic.substr(s, k - s));
                s = k + 1;
            }
        size_t reuse = 0;
        if (!have_prev || tenant != prev_tenant) {
            roots.push_back((uint32_t)pn.size());
            pn.push_back({NONE, dict_h.intern(tenant), NONE, 0, 0, (uint32_t)i, 0, 0, false});
            stack.assign(1, (uint32_t)pn.size() - 1);
            prev_levels.clear();
            prev_tenant = tenant;
            have_prev = true;
        } else {
            while (reuse < levels.size() && reuse < prev_levels.size() && levels[reuse] == prev_levels[reuse]) reuse++;
        }
        stack.resize(reuse + 1);
        prev_levels.resize(reuse);
        for (size_t l = reuse; l < levels.size(); l++) {
            const uint32_t parent = stack.back();
            const uint32_t id = (uint32_t)pn.size();
            pn.push_back({parent, dict_h.intern(levels[l]), NONE, 0, 0, (uint32_t)i, 0, 0,
                          !levels[l].empty() && levels[l][0] == '$'});
            if (pn[parent].first_child == NONE) pn[parent].first_child = id;
            pn[parent].n_children++;
            stack.push_back(id);
            prev_levels.push_back(levels[l]);
        }
        pn[stack.back()].term = 1; // sorted unique input: each topic ends at a distinct node
    }
    // subtree id ranges: preorder means a node's subtree is the node index range up to the next node that is not a
    // descendant; sub_begin was set to the first topic at/below the node, sub_end follows from the parent chain
    {
        std::vector<uint32_t> st;
        for (uint32_t v = 0; v < pn.size(); v++) {
            while (!st.empty() && st.back() != pn[v].parent) {
                pn[st.back()].sub_end = pn[v].sub_begin;
                st.pop_back();
            }
            st.push_back(v);
        }
        while (!st.empty()) {
            pn[st.back()].sub_end = (uint32_t)n;
            st.pop_back();
        }
    }
    // breadth-first numbering: roots first, then level by level; children of a node keep their (sorted) order.  The
    // children of node v in preorder: first_child, then repeatedly "the next node after that child's subtree".
    std::vector<uint32_t> order; // bfs index -> preorder id
    order.reserve(pn.size());
    for (uint32_t r : roots) order.push_back(r);
    // next sibling in preorder = first node after the subtree; subtree node-extent via a second stack pass
    std::vector<uint32_t> next_after(pn.size(), (uint32_t)pn.size());
    {
        std::vector<uint32_t> st;
        for (uint32_t v = 0; v < pn.size(); v++) {
            while (!st.empty() && st.back() != pn[v].parent) {
                next_after[st.back()] = v;
                st.pop_back();
            }
            st.push_back(v);
        }
    }
    for (size_t q = 0; q < order.size(); q++) {
        const uint32_t v = order[q];
        uint32_t c = pn[v].first_child;
        for (uint32_t k = 0; k < pn[v].n_children; k++) {
            order.push_back(c);
            c = next_after[c];
        }
    }
    for (uint32_t b = 0; b < order.size(); b++) pn[order[b]].bfs = b;
    nodes.assign(std::max<size_t>(order.size(), 1), RNode{0, 0, 0, 0});
    const uint32_t eslots = pow2_at_least(std::max<uint64_t>(8, (uint64_t)order.size() * 2));
    const uint32_t bmask = eslots / 4 - 1;
    edges.assign(eslots, REdge{NONE, 0, 0, 0});
    uint32_t running = (uint32_t)roots.size(); // child_begin is monotone also over childless nodes, so that the children
    for (uint32_t b = 0; b < order.size(); b++) { // of the node range [a, b) are [child_begin(a), child_begin(b-1) + count(b-1))
        const PNode& p = pn[order[b]];
        RNode& r = nodes[b];
        r.child_begin = running;
        running += p.n_children;
        r.child_count = p.n_children | (p.term ? RN_TERM : 0u);
        r.sub_begin = p.sub_begin;
        r.sub_end = p.sub_end;
        if (p.parent != NONE) {
            const uint32_t pb = pn[p.parent].bfs;
            uint32_t bk = redge_bucket(pb, p.token, bmask), s = NONE;
            for (;;) {
                for (uint32_t j = 0; j < 4 && s == NONE; j++)
                    if (edges[4 * bk + j].parent == NONE) s = 4 * bk + j;
                if (s != NONE) break;
                bk = (bk + 1) & bmask;
            }
            edges[s] = REdge{pb, p.token, b, 0};
        }
    }
    const uint32_t tslots = pow2_at_least((uint64_t)roots.size() * 2);
    tenants.assign(tslots, RTenantSlot{0, 0, 0, 0, 0, 0, {0, 0}});
    for (uint32_t r : roots) {
        RTenantSlot t{pn[r].token, pn[r].bfs, 0, 0, 0, 0, {0, 0}};
        // the '$' children of the root form one contiguous run (children are sorted by label bytes)
        uint32_t c = pn[r].first_child;
        bool in_run = false;
        for (uint32_t k = 0; k < pn[r].n_children; k++) {
            if (pn[c].sys) {
                if (!in_run) {
                    t.sys_node_lo = pn[c].bfs;
                    t.sys_id_lo = pn[c].sub_begin;
                    in_run = true;
                }
                t.sys_node_hi = pn[c].bfs + 1;
                t.sys_id_hi = pn[c].sub_end;
            }
            c = next_after[c];
        }
        uint32_t d = tenant_hash(t.token) & (tslots - 1);
        while (tenants[d].token) d = (d + 1) & (tslots - 1);
        tenants[d] = t;
    }
    flatten_dict(dict_h, dict, pool);
    n_topics = n;
    n_tenants = roots.size();
    return true;
}

} // namespace bmq
