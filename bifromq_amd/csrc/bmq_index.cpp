// bmq_index.cpp -- host-side builder: route keys -> dictionary + filter trie -> 32-byte slot tables.
//
// What it indexes is what the reference scans: the route keys of a KV range
// (SCHEMA/KVSchemaUtil.java:91-130).  The reference joins those keys against the expansion set of the
// publish topics (DW/cache/TenantRouteMatcher.java:88-156); this builder turns the same keys into the
// inverse structure (a trie of filters) once per rebuild so that the GPU can walk it per topic.
#include "bmq_index.h"

#include "bmq_dict.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

namespace bmq {

// ------------------------------------------------------------------------------------------------------------
// codec
// ------------------------------------------------------------------------------------------------------------
// Java String.hashCode over the UTF-16 code units of a UTF-8 string (used for the bucket byte,
// SCHEMA/KVSchemaUtil.java:127-130).
int32_t java_string_hash(std::string_view s) {
    uint32_t h = 0;
    const size_t n = s.size();
    size_t i = 0;
    while (i < n) {
        const uint32_t c = (uint8_t)s[i];
        uint32_t cp;
        if (c < 0x80) {
            cp = c;
            i += 1;
        } else if ((c & 0xE0) == 0xC0 && i + 1 < n) {
            cp = ((c & 0x1F) << 6) | ((uint8_t)s[i + 1] & 0x3F);
            i += 2;
        } else if ((c & 0xF0) == 0xE0 && i + 2 < n) {
            cp = ((c & 0x0F) << 12) | (((uint8_t)s[i + 1] & 0x3F) << 6) | ((uint8_t)s[i + 2] & 0x3F);
            i += 3;
        } else if ((c & 0xF8) == 0xF0 && i + 3 < n) {
            cp = ((c & 0x07) << 18) | (((uint8_t)s[i + 1] & 0x3F) << 12) | (((uint8_t)s[i + 2] & 0x3F) << 6) |
                 ((uint8_t)s[i + 3] & 0x3F);
            i += 4;
        } else {
            cp = 0xFFFD;
            i += 1;
        }
        if (cp >= 0x10000) { // surrogate pair
            cp -= 0x10000;
            h = 31u * h + (0xD800u + (cp >> 10));
            h = 31u * h + (0xDC00u + (cp & 0x3FF));
        } else {
            h = 31u * h + cp;
        }
    }
    return (int32_t)h;
}

static inline void put_u16be(std::string& s, size_t v) {
    s.push_back((char)((v >> 8) & 0xFF));
    s.push_back((char)(v & 0xFF));
}

// key = 0x00 | u16be(len tenant) | tenant | (level 0x00)* | 0x00 | bucket | flag | receiver | u16be(len receiver)
std::string encode_route_key(std::string_view tenant, std::string_view filter, uint8_t flag,
                             std::string_view receiver) {
    std::string k;
    k.reserve(tenant.size() + filter.size() + receiver.size() + 10);
    k.push_back('\0');
    put_u16be(k, tenant.size());
    k.append(tenant);
    for (char c : filter) k.push_back(c == '/' ? '\0' : c); // TopicUtil.escape: '/' -> NUL
    k.push_back('\0');                                       // terminates the last level
    k.push_back('\0');                                       // end of filter
    const uint32_t h = (uint32_t)java_string_hash(receiver);
    k.push_back((char)((h ^ (h >> 16)) & 0xFF));
    k.push_back((char)flag);
    k.append(receiver);
    put_u16be(k, receiver.size());
    return k;
}

// Parsed from both ends like RouteDetailCache.java:53-109.
bool decode_route_key(std::string_view k, RouteKeyParts& out) {
    if (k.size() < 3 + 2 + 2 + 2 || k[0] != 0) return false;
    const size_t tlen = ((size_t)(uint8_t)k[1] << 8) | (uint8_t)k[2];
    const size_t esc_start = 3 + tlen;
    const size_t rlen = ((size_t)(uint8_t)k[k.size() - 2] << 8) | (uint8_t)k[k.size() - 1];
    if (k.size() < esc_start + 4 + rlen + 2) return false;
    const size_t recv_start = k.size() - 2 - rlen;
    const size_t esc_end = recv_start - 4; // level-terminating NUL, filter-terminating NUL, bucket, flag
    if (k[esc_end] != 0 || k[esc_end + 1] != 0) return false;
    out.tenant = k.substr(3, tlen);
    out.esc_filter = k.substr(esc_start, esc_end - esc_start);
    out.bucket = (uint8_t)k[recv_start - 2];
    out.flag = (uint8_t)k[recv_start - 1];
    out.receiver = k.substr(recv_start, rlen);
    return out.flag >= 1 && out.flag <= 3;
}

// ------------------------------------------------------------------------------------------------------------
// KeySet
// ------------------------------------------------------------------------------------------------------------
namespace {
struct KeyRef {
    const uint8_t* p;
    uint32_t n;
};
inline bool key_less(const KeyRef& a, const KeyRef& b) {
    const uint32_t m = a.n < b.n ? a.n : b.n;
    const int c = m ? memcmp(a.p, b.p, m) : 0;
    return c < 0 || (c == 0 && a.n < b.n);
}
inline bool key_eq(const KeyRef& a, const KeyRef& b) { return a.n == b.n && (a.n == 0 || memcmp(a.p, b.p, a.n) == 0); }

void parallel_sort(std::vector<KeyRef>& v) {
    const size_t n = v.size();
    unsigned hw = std::thread::hardware_concurrency();
    size_t parts = 1;
    while (parts * 2 <= (hw ? hw : 1) && parts < 16) parts *= 2;
    if (n < (1u << 16) || parts == 1) {
        std::sort(v.begin(), v.end(), key_less);
        return;
    }
    std::vector<size_t> cut(parts + 1);
    for (size_t i = 0; i <= parts; i++) cut[i] = n * i / parts;
    {
        std::vector<std::thread> th;
        for (size_t i = 0; i < parts; i++)
            th.emplace_back([&, i] { std::sort(v.begin() + cut[i], v.begin() + cut[i + 1], key_less); });
        for (auto& t : th) t.join();
    }
    for (size_t w = 1; w < parts; w *= 2) {
        std::vector<std::thread> th;
        for (size_t i = 0; i + w < parts; i += 2 * w)
            th.emplace_back([&, i, w] {
                std::inplace_merge(v.begin() + cut[i], v.begin() + cut[i + w], v.begin() + cut[std::min(i + 2 * w, parts)],
                                   key_less);
            });
        for (auto& t : th) t.join();
    }
}
} // namespace

void KeySet::assign(const uint8_t* keys, const uint32_t* key_off, uint32_t n) {
    std::vector<KeyRef> refs(n);
    bool sorted = true;
    for (uint32_t i = 0; i < n; i++) {
        refs[i] = {keys + key_off[i], key_off[i + 1] - key_off[i]};
        if (i && !key_less(refs[i - 1], refs[i])) sorted = false; // strict: also catches duplicates
    }
    if (!sorted) {
        parallel_sort(refs);
        refs.erase(std::unique(refs.begin(), refs.end(), key_eq), refs.end());
    }
    uint64_t total = 0;
    for (auto& r : refs) total += r.n;
    std::vector<uint8_t> nb(total ? total : 1);
    std::vector<uint64_t> no(refs.size() + 1);
    uint64_t o = 0;
    for (size_t i = 0; i < refs.size(); i++) {
        no[i] = o;
        if (refs[i].n) memcpy(nb.data() + o, refs[i].p, refs[i].n);
        o += refs[i].n;
    }
    no[refs.size()] = o;
    bytes.swap(nb);
    off.swap(no);
}

int64_t KeySet::find(std::string_view k) const {
    size_t lo = 0, hi = size();
    const KeyRef kr{(const uint8_t*)k.data(), (uint32_t)k.size()};
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        const std::string_view m = key(mid);
        if (key_less(KeyRef{(const uint8_t*)m.data(), (uint32_t)m.size()}, kr)) lo = mid + 1;
        else hi = mid;
    }
    if (lo < size() && key(lo) == k) return (int64_t)lo;
    return -1;
}

void KeySet::apply(const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n) {
    // last op per key wins (ops are applied in order); then one merge pass over the sorted set
    std::vector<std::pair<KeyRef, uint32_t>> ops(n);
    for (uint32_t i = 0; i < n; i++) ops[i] = {KeyRef{keys + key_off[i], key_off[i + 1] - key_off[i]}, i};
    std::stable_sort(ops.begin(), ops.end(),
                     [](const auto& a, const auto& b) { return key_less(a.first, b.first); });
    std::vector<uint8_t> nb;
    nb.reserve(bytes.size() + (n ? key_off[n] : 0));
    std::vector<uint64_t> no;
    no.reserve(off.size() + n);
    no.push_back(0);
    auto emit = [&](const uint8_t* p, uint32_t len) {
        nb.insert(nb.end(), p, p + len);
        no.push_back(nb.size());
    };
    size_t i = 0, j = 0;
    const size_t m = size();
    while (i < m || j < ops.size()) {
        if (j == ops.size()) {
            const auto k = key(i++);
            emit((const uint8_t*)k.data(), (uint32_t)k.size());
            continue;
        }
        size_t j2 = j; // run of ops on the same key; the last one decides
        while (j2 + 1 < ops.size() && key_eq(ops[j2 + 1].first, ops[j].first)) j2++;
        const KeyRef& ok = ops[j2].first;
        const bool is_put = op[ops[j2].second] == 0;
        if (i < m) {
            const auto k = key(i);
            const KeyRef kr{(const uint8_t*)k.data(), (uint32_t)k.size()};
            if (key_less(kr, ok)) {
                emit(kr.p, kr.n);
                i++;
                continue;
            }
            if (key_eq(kr, ok)) i++; // replaced or deleted
        }
        if (is_put) emit(ok.p, ok.n);
        j = j2 + 1;
    }
    if (nb.empty()) nb.push_back(0);
    bytes.swap(nb);
    off.swap(no);
}

// ------------------------------------------------------------------------------------------------------------
// builder
// ------------------------------------------------------------------------------------------------------------
namespace {

// (parent node, token) -> node
struct ChildMap {
    std::vector<uint64_t> keys;
    std::vector<uint32_t> vals;
    uint64_t mask = 0, count = 0;
    static constexpr uint64_t EMPTY = ~0ull;
    ChildMap() { keys.assign(1024, EMPTY); vals.assign(1024, 0); mask = 1023; }
    static uint64_t mix(uint64_t k) {
        k ^= k >> 33;
        k *= 0xFF51AFD7ED558CCDull;
        k ^= k >> 33;
        k *= 0xC4CEB9FE1A85EC53ull;
        k ^= k >> 33;
        return k;
    }
    void grow() {
        std::vector<uint64_t> nk(keys.size() * 2, EMPTY);
        std::vector<uint32_t> nv(keys.size() * 2, 0);
        const uint64_t nm = nk.size() - 1;
        for (size_t i = 0; i < keys.size(); i++)
            if (keys[i] != EMPTY) {
                uint64_t j = mix(keys[i]) & nm;
                while (nk[j] != EMPTY) j = (j + 1) & nm;
                nk[j] = keys[i];
                nv[j] = vals[i];
            }
        keys.swap(nk);
        vals.swap(nv);
        mask = nm;
    }
    // returns reference to value slot; created = true if new (value uninitialised)
    uint32_t& get(uint32_t parent, uint32_t token, bool& created) {
        if ((count + 1) * 2 > keys.size()) grow();
        const uint64_t k = ((uint64_t)parent << 32) | token;
        uint64_t i = mix(k) & mask;
        while (keys[i] != EMPTY) {
            if (keys[i] == k) {
                created = false;
                return vals[i];
            }
            i = (i + 1) & mask;
        }
        keys[i] = k;
        count++;
        created = true;
        return vals[i];
    }
};

struct BuildNode {
    uint32_t parent; // node id, NONE for tenant roots
    uint32_t token;
    uint32_t own_group, hash_group; // route groups of "<path>" and "<path>/#" (NONE: none)
};

} // namespace

bool DistIndexHost::build(const KeySet& ks) {
    error.clear();
    const size_t n = ks.size();
    if (n >= 0x7FFFFFF0ull) {
        error = "too many routes";
        return false;
    }
    HostDict dict_h;
    ChildMap children;
    std::vector<BuildNode> nodes;
    nodes.reserve(n + 16);
    std::vector<uint32_t> group_count, group_first, group_last; // per route group (one per filter with routes)
    std::vector<uint32_t> route_group(n ? n : 1);
    std::vector<uint32_t> root_nodes; // node index of every tenant root, ascending (tenants are contiguous in key order)

    // path of the previous key, for prefix reuse (sorted keys share long prefixes)
    std::string_view prev_tenant;
    bool have_prev = false;
    uint32_t prev_root = NONE;
    std::vector<std::string_view> prev_levels;
    std::vector<uint32_t> prev_nodes; // node reached after consuming prev_levels[0..i]

    std::vector<std::string_view> levels;
    for (size_t r = 0; r < n; r++) {
        RouteKeyParts kp;
        if (!decode_route_key(ks.key(r), kp)) {
            error = "malformed route key at rank " + std::to_string(r);
            return false;
        }
        levels.clear();
        { // split escaped filter on NUL, keeping empty levels (TopicUtil.parse(escaped = true))
            size_t s = 0;
            const std::string_view f = kp.esc_filter;
            for (size_t i = 0; i <= f.size(); i++)
                if (i == f.size() || f[i] == '\0') {
                    levels.push_back(f.substr(s, i - s));
                    s = i + 1;
                }
        }
        uint32_t node;
        size_t reuse = 0;
        if (have_prev && kp.tenant == prev_tenant) {
            node = prev_root;
            while (reuse < levels.size() && reuse < prev_levels.size() && levels[reuse] == prev_levels[reuse] &&
                   prev_nodes[reuse] != NONE)
                reuse++;
            if (reuse) node = prev_nodes[reuse - 1];
        } else {
            const uint32_t ttok = dict_h.intern(kp.tenant);
            bool created;
            uint32_t& v = children.get(NONE, ttok, created);
            if (!created) {
                error = "keys of one tenant are not contiguous (input not sorted?)";
                return false;
            }
            v = (uint32_t)nodes.size();
            nodes.push_back({NONE, ttok, NONE, NONE});
            root_nodes.push_back(v);
            node = v;
            prev_root = node;
            prev_tenant = kp.tenant;
            have_prev = true;
            prev_levels.clear();
            prev_nodes.clear();
        }
        prev_levels.resize(reuse);
        prev_nodes.resize(reuse);
        bool is_hash = false;
        for (size_t li = reuse; li < levels.size(); li++) {
            const std::string_view lv = levels[li];
            if (lv == "#" && li + 1 == levels.size()) { // '#' is a wildcard only as the last level
                is_hash = true;
                prev_levels.push_back(lv);
                prev_nodes.push_back(NONE); // not a node; never reused
                break;
            }
            const uint32_t tok = (lv == "+") ? TOK_PLUS : dict_h.intern(lv);
            bool created;
            uint32_t& v = children.get(node, tok, created);
            if (created) {
                v = (uint32_t)nodes.size();
                nodes.push_back({node, tok, NONE, NONE});
            }
            node = v;
            prev_levels.push_back(lv);
            prev_nodes.push_back(node);
        }
        uint32_t& g = is_hash ? nodes[node].hash_group : nodes[node].own_group;
        if (g == NONE) {
            g = (uint32_t)group_count.size();
            group_count.push_back(0);
            group_first.push_back((uint32_t)r);
            group_last.push_back((uint32_t)r);
        }
        group_count[g]++;
        group_last[g] = (uint32_t)r;
        route_group[r] = g;
    }

    // ---- route groups: a group whose ids are one contiguous rank range is stored as that range (the overwhelmingly
    // common case); otherwise (SURVEY.md 8c quirk ii: keys of "x" interleave with keys of "x//...") its ids go to
    // route_pos[] and the range is flagged RANGE_INDIRECT ------------------------------------------------------------------
    std::vector<uint32_t> group_begin(group_count.size());
    uint64_t indirect_total = 0;
    for (size_t g = 0; g < group_count.size(); g++) {
        if (group_last[g] - group_first[g] + 1 == group_count[g]) {
            group_begin[g] = group_first[g];
        } else {
            group_begin[g] = (uint32_t)indirect_total;
            indirect_total += group_count[g];
            group_count[g] |= RANGE_INDIRECT;
        }
    }
    route_pos.assign(indirect_total ? indirect_total : 1, 0);
    if (indirect_total) {
        std::vector<uint32_t> cur(group_begin);
        for (size_t r = 0; r < n; r++) {
            const uint32_t g = route_group[r];
            if (group_count[g] & RANGE_INDIRECT) route_pos[cur[g]++] = (uint32_t)r;
        }
    }
    std::vector<uint32_t>().swap(route_group);

    // ---- one region of the slot table per tenant; bucketised linear probing, load factor <= 1/2 ---------------------
    double region_factor = 2.0; // slots per node; BMQ_REGION_FACTOR overrides for experiments
    if (const char* rf = getenv("BMQ_REGION_FACTOR")) region_factor = std::max(1.25, atof(rf));
    const size_t nt = root_nodes.size();
    std::vector<uint64_t> region_base(nt + 1, 0); // in slots
    std::vector<uint32_t> region_buckets(nt);
    for (size_t t = 0; t < nt; t++) {
        const uint64_t cnt = (t + 1 < nt ? root_nodes[t + 1] : nodes.size()) - root_nodes[t];
        region_buckets[t] = (uint32_t)std::max<uint64_t>(2, (uint64_t)(cnt * region_factor / 2.0) + 1);
        region_base[t + 1] = region_base[t] + 2ull * region_buckets[t];
    }
    const uint64_t slots = std::max<uint64_t>(region_base.back(), 2);
    if (slots >= 0xFFFFFFF0ull) {
        error = "trie too large";
        return false;
    }
    TrieSlot empty_slot{NONE, 0, 0, 0, 0, 0, NONE, 0};
    trie.assign(slots, empty_slot);
    const uint32_t tslots = pow2_at_least((uint64_t)nt * 2);
    tenants.assign(tslots, TenantSlot{0, NONE, 0, 1});
    {
        std::vector<uint32_t> slot_of(nodes.size());
        std::atomic<size_t> next{0};
        auto work = [&]() { // regions are independent: place them on all host cores (parents precede children)
            for (;;) {
                const size_t t = next.fetch_add(1);
                if (t >= nt) break;
                const size_t n0 = root_nodes[t], n1 = t + 1 < nt ? root_nodes[t + 1] : nodes.size();
                const uint32_t base = (uint32_t)region_base[t], nb = region_buckets[t];
                for (size_t i = n0; i < n1; i++) {
                    const BuildNode& b = nodes[i];
                    const uint32_t pslot = b.parent == NONE ? ROOT_PARENT : slot_of[b.parent];
                    uint32_t bk = edge_bucket(pslot, b.token, nb), s;
                    for (;;) {
                        s = base + 2 * bk;
                        if (trie[s].parent == NONE) break;
                        if (trie[++s].parent == NONE) break;
                        bk = (bk + 1 == nb) ? 0 : bk + 1;
                    }
                    TrieSlot& ts = trie[s];
                    ts.parent = pslot;
                    ts.token = b.token;
                    if (b.own_group != NONE) {
                        ts.own_begin = group_begin[b.own_group];
                        ts.own_count = group_count[b.own_group];
                    }
                    if (b.hash_group != NONE) {
                        ts.hash_begin = group_begin[b.hash_group];
                        ts.hash_count = group_count[b.hash_group];
                    }
                    slot_of[i] = s;
                    if (b.parent != NONE) {
                        TrieSlot& pr = trie[pslot];
                        if (b.token == TOK_PLUS) pr.plus_child = s;
                        else pr.lit_bloom |= 1u << bloom_bit(b.token);
                    }
                }
            }
        };
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned nth = (unsigned)std::min<size_t>(hw ? hw : 1, std::max<size_t>(nt, 1));
        std::vector<std::thread> th;
        for (unsigned w = 1; w < nth; w++) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
        for (size_t t = 0; t < nt; t++) {
            const uint32_t ttok = nodes[root_nodes[t]].token;
            uint32_t d = tenant_hash(ttok) & (tslots - 1);
            while (tenants[d].token) d = (d + 1) & (tslots - 1);
            tenants[d] = TenantSlot{ttok, slot_of[root_nodes[t]], (uint32_t)region_base[t], region_buckets[t]};
        }
    }

    flatten_dict(dict_h, dict, pool);

    n_routes = n;
    n_tenants = nt;
    n_nodes = nodes.size();
    n_tokens = dict_h.entries.size();
    return true;
}

uint32_t DistIndexHost::find_token(std::string_view level) const { return dict_find(dict, pool, level); }

const TenantSlot* DistIndexHost::find_tenant(uint32_t token) const {
    if (tenants.empty() || token == TOK_UNKNOWN) return nullptr;
    const uint32_t mask = (uint32_t)tenants.size() - 1;
    uint32_t d = tenant_hash(token) & mask;
    while (tenants[d].token) {
        if (tenants[d].token == token) return &tenants[d];
        d = (d + 1) & mask;
    }
    return nullptr;
}

uint32_t DistIndexHost::find_child(const TenantSlot& r, uint32_t parent_slot, uint32_t token) const {
    uint32_t bk = edge_bucket(parent_slot, token, r.buckets);
    for (;;) {
        bool full = true;
        for (uint32_t j = 0; j < 2; j++) {
            const TrieSlot& t = trie[r.base + 2 * bk + j];
            if (t.parent == NONE) full = false;
            else if (t.parent == parent_slot && t.token == token) return r.base + 2 * bk + j;
        }
        if (!full) return NONE;
        bk = (bk + 1 == r.buckets) ? 0 : bk + 1;
    }
}

// slot of the node for (tenant, filter); for "x/#" the node of "x" with is_hash = true
uint32_t DistIndexHost::find_filter_node(std::string_view tenant, std::string_view filter, bool& is_hash) const {
    is_hash = false;
    const TenantSlot* r = find_tenant(find_token(tenant));
    if (!r) return NONE;
    uint32_t slot = r->root;
    size_t s = 0;
    for (size_t i = 0; i <= filter.size() && slot != NONE; i++)
        if (i == filter.size() || filter[i] == '/') {
            const std::string_view lv = filter.substr(s, i - s);
            s = i + 1;
            if (lv == "#" && i == filter.size()) {
                is_hash = true;
                break;
            }
            const uint32_t tok = lv == "+" ? TOK_PLUS : find_token(lv);
            if (tok == TOK_UNKNOWN) return NONE;
            slot = find_child(*r, slot, tok);
        }
    return slot;
}

} // namespace bmq
