// bmq_index.cpp -- host-side dist index: route keys per tenant -> dictionary + per-tenant filter-trie regions.
//
// What it indexes is what the reference scans: the route keys of a KV range (SCHEMA/KVSchemaUtil.java:91-130).  The
// reference joins those keys against the expansion set of the publish topics on every call
// (DW/cache/TenantRouteMatcher.java:88-156); here the same keys become the inverse structure (a trie of filters) once,
// and a batch of route mutations rebuilds only the regions of the tenants it touches, on all host cores.
#include "bmq_index.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
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
inline bool sv_less(std::string_view a, std::string_view b) {
    const size_t m = a.size() < b.size() ? a.size() : b.size();
    const int c = m ? memcmp(a.data(), b.data(), m) : 0;
    return c < 0 || (c == 0 && a.size() < b.size());
}
} // namespace

void KeySet::assign(std::vector<std::string_view>& keys) {
    if (!std::is_sorted(keys.begin(), keys.end(), sv_less)) std::sort(keys.begin(), keys.end(), sv_less);
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    uint64_t total = 0;
    for (auto& k : keys) total += k.size();
    std::vector<uint8_t> nb(total ? total : 1);
    std::vector<uint64_t> no(keys.size() + 1);
    uint64_t o = 0;
    for (size_t i = 0; i < keys.size(); i++) {
        no[i] = o;
        if (!keys[i].empty()) memcpy(nb.data() + o, keys[i].data(), keys[i].size());
        o += keys[i].size();
    }
    no[keys.size()] = o;
    bytes.swap(nb);
    off.swap(no);
}

int64_t KeySet::find(std::string_view k) const {
    size_t lo = 0, hi = size();
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (sv_less(key(mid), k)) lo = mid + 1;
        else hi = mid;
    }
    return (lo < size() && key(lo) == k) ? (int64_t)lo : -1;
}

void KeySet::apply(std::vector<std::pair<std::string_view, uint8_t>>& ops, std::vector<int64_t>* src) {
    if (src) src->clear();
    // last op per key wins (ops are applied in order); then one merge pass over the sorted set
    std::vector<uint32_t> idx(ops.size());
    for (uint32_t i = 0; i < ops.size(); i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return sv_less(ops[a].first, ops[b].first); });
    std::vector<uint8_t> nb;
    nb.reserve(bytes.size() + 64 * ops.size());
    std::vector<uint64_t> no;
    no.reserve(off.size() + ops.size());
    no.push_back(0);
    auto emit = [&](std::string_view k, int64_t from) {
        nb.insert(nb.end(), k.begin(), k.end());
        no.push_back(nb.size());
        if (src) src->push_back(from);
    };
    size_t i = 0, j = 0;
    const size_t m = size();
    while (i < m || j < idx.size()) {
        if (j == idx.size()) {
            emit(key(i), (int64_t)i);
            i++;
            continue;
        }
        size_t j2 = j; // run of ops on the same key; the last one decides
        while (j2 + 1 < idx.size() && ops[idx[j2 + 1]].first == ops[idx[j]].first) j2++;
        const std::string_view ok = ops[idx[j2]].first;
        const bool is_put = ops[idx[j2]].second == 0;
        if (i < m) {
            const std::string_view k = key(i);
            if (sv_less(k, ok)) {
                emit(k, (int64_t)i);
                i++;
                continue;
            }
            if (k == ok) i++; // replaced or deleted
        }
        if (is_put) emit(ok, -(int64_t)idx[j2] - 1);
        j = j2 + 1;
    }
    if (nb.empty()) nb.push_back(0);
    bytes.swap(nb);
    off.swap(no);
}

// ------------------------------------------------------------------------------------------------------------
// per-tenant trie construction
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
    uint32_t parent; // node index, NONE for the root
    uint32_t token;  // LOCAL token while building, global after translation
    uint32_t own_group, hash_group;
};

// result of phase 1 for one tenant (everything tenant-local)
struct TenantBuild {
    TenantState* st = nullptr;
    std::vector<BuildNode> nodes;               // creation order: parents first
    std::vector<std::string_view> local_strings; // local token - TOK_FIRST -> level string (views into st->keys)
    std::vector<uint32_t> l2g;                   // local token index -> global token
    std::vector<uint32_t> group_begin, group_count;
    std::vector<uint32_t> indirect;
    std::string error;
    std::shared_ptr<void> hold; // keeps a persistent trie alive whose strings local_strings points into
};

// ---- EXPERIMENTAL (BMQ_INCREMENTAL=1): the tenant's trie kept between applies ---------------------------------------
// Local tokens, the child map and the nodes persist; every key remembers the node it hangs off.  A batch of mutations then
// walks / extends the trie for the keys it puts only, and the id ranges of all nodes are recomputed from key_node[] in one
// pass over the (merged) key set -- no key of the tenant is parsed again.  Nodes that lose their last route stay in the
// persistent trie (they may come back) but are left out of the image; when they pile up the state is dropped and the next
// refresh parses the keys from scratch.
constexpr uint32_t KEY_HASH = 0x80000000u; // key_node flag: the key is a route of "<node path>/#"
struct TenantInc {
    HostDict local;
    std::deque<std::string> local_store; // owned level strings (key bytes move when the key set is merged)
    ChildMap children;
    std::vector<BuildNode> nodes;        // creation order: parents first; own_group / hash_group unused here
    std::vector<uint32_t> key_node;      // per key, in rank order: node | KEY_HASH
    uint32_t intern(std::string_view s) {
        const size_t before = local.entries.size();
        const uint32_t tok = local.intern(s);
        if (local.entries.size() != before) {
            local_store.emplace_back(s);
            local.entries.back().s = local_store.back();
        }
        return tok;
    }
    // node of a route key (creating the path if needed); false: malformed key
    bool locate(std::string_view key, uint32_t& out) {
        RouteKeyParts kp;
        if (!decode_route_key(key, kp)) return false;
        const std::string_view f = kp.esc_filter;
        uint32_t node = 0;
        size_t s0 = 0;
        for (size_t i = 0; i <= f.size(); i++)
            if (i == f.size() || f[i] == '\0') {
                const std::string_view lv = f.substr(s0, i - s0);
                s0 = i + 1;
                if (lv == "#" && i == f.size()) {
                    out = node | KEY_HASH;
                    return true;
                }
                const uint32_t tok = (lv == "+") ? TOK_PLUS : intern(lv);
                bool created;
                uint32_t& v = children.get(node, tok, created);
                if (created) {
                    v = (uint32_t)nodes.size();
                    nodes.push_back({node, tok, NONE, NONE});
                }
                node = v;
            }
        out = node;
        return true;
    }
};

static bool incremental_enabled() {
    static const bool on = [] {
        const char* v = getenv("BMQ_INCREMENTAL");
        return v && atoi(v) != 0;
    }();
    return on;
}

// TenantBuild from the persistent trie + key_node[] (no key is parsed)
void build_from_inc(TenantBuild& b, TenantInc& inc) {
    const size_t n = inc.key_node.size(), nn = inc.nodes.size();
    // per (node, kind): count / first rank / last rank
    std::vector<uint32_t> cnt(2 * nn, 0), first(2 * nn, 0), last(2 * nn, 0);
    for (size_t r = 0; r < n; r++) {
        const uint32_t kn = inc.key_node[r], g = 2 * (kn & ~KEY_HASH) + (kn >> 31);
        if (cnt[g]++ == 0) first[g] = (uint32_t)r;
        last[g] = (uint32_t)r;
    }
    // liveness: a node is part of the image if it or a descendant has routes (children have larger indices than parents)
    std::vector<uint8_t> alive(nn, 0);
    alive[0] = 1;
    for (size_t i = nn; i-- > 0;) {
        if (cnt[2 * i] || cnt[2 * i + 1]) alive[i] = 1;
        if (alive[i] && inc.nodes[i].parent != NONE) alive[inc.nodes[i].parent] = 1;
    }
    std::vector<uint32_t> remap(nn, NONE);
    b.nodes.clear();
    b.group_begin.clear();
    b.group_count.clear();
    std::vector<uint32_t> group_of(2 * nn, NONE);
    uint32_t ind = 0;
    for (size_t i = 0; i < nn; i++) {
        if (!alive[i]) continue;
        remap[i] = (uint32_t)b.nodes.size();
        BuildNode bn = inc.nodes[i];
        bn.parent = bn.parent == NONE ? NONE : remap[bn.parent];
        bn.own_group = bn.hash_group = NONE;
        for (uint32_t kind = 0; kind < 2; kind++) {
            const size_t g = 2 * i + kind;
            if (!cnt[g]) continue;
            group_of[g] = (uint32_t)b.group_count.size();
            (kind ? bn.hash_group : bn.own_group) = group_of[g];
            if (last[g] - first[g] + 1 == cnt[g]) {
                b.group_begin.push_back(first[g]);
                b.group_count.push_back(cnt[g]);
            } else { // ids are not one contiguous rank range (SURVEY.md 8c quirk ii): listed in the indirect array
                b.group_begin.push_back(ind);
                b.group_count.push_back(cnt[g] | RANGE_INDIRECT);
                ind += cnt[g];
            }
        }
        b.nodes.push_back(bn);
    }
    b.indirect.assign(ind, 0);
    if (ind) {
        std::vector<uint32_t> cur(b.group_begin);
        for (size_t r = 0; r < n; r++) {
            const uint32_t kn = inc.key_node[r], g = group_of[2 * (kn & ~KEY_HASH) + (kn >> 31)];
            if (b.group_count[g] & RANGE_INDIRECT) b.indirect[cur[g]++] = (uint32_t)r;
        }
    }
    b.local_strings.resize(inc.local.entries.size());
    for (size_t i = 0; i < inc.local.entries.size(); i++) b.local_strings[i] = inc.local.entries[i].s;
    if (nn > 2 * b.nodes.size() + 64) { // mostly dead nodes: drop the state, the next refresh parses the keys again
        b.hold = b.st->inc;
        b.st->inc.reset();
    }
}

// Phase 1: keys -> trie nodes (local tokens), route groups.  Keys are sorted: consecutive keys share long prefixes.
void build_tenant_trie(TenantBuild& b) {
    const KeySet& ks = b.st->keys;
    const size_t n = ks.size();
    HostDict local;
    ChildMap children;
    b.nodes.clear();
    b.nodes.reserve(n + 8);
    b.nodes.push_back({NONE, 0, NONE, NONE}); // root; its token (the tenant id) is filled in globally
    std::vector<uint32_t> group_first, group_last;
    std::vector<uint32_t> route_group(n ? n : 1);
    std::vector<std::string_view> prev_levels, levels;
    std::vector<uint32_t> prev_nodes;
    std::shared_ptr<TenantInc> keep; // BMQ_INCREMENTAL=1: the trie outlives this call
    if (incremental_enabled()) {
        keep = std::make_shared<TenantInc>();
        keep->key_node.assign(n, 0);
    }
    for (size_t r = 0; r < n; r++) {
        RouteKeyParts kp;
        if (!decode_route_key(ks.key(r), kp)) {
            b.error = "malformed route key";
            return;
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
        uint32_t node = 0;
        size_t reuse = 0;
        while (reuse < levels.size() && reuse < prev_levels.size() && levels[reuse] == prev_levels[reuse] && prev_nodes[reuse] != NONE)
            reuse++;
        if (reuse) node = prev_nodes[reuse - 1];
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
            const uint32_t tok = (lv == "+") ? TOK_PLUS : local.intern(lv);
            bool created;
            uint32_t& v = children.get(node, tok, created);
            if (created) {
                v = (uint32_t)b.nodes.size();
                b.nodes.push_back({node, tok, NONE, NONE});
            }
            node = v;
            prev_levels.push_back(lv);
            prev_nodes.push_back(node);
        }
        if (keep) keep->key_node[r] = node | (is_hash ? KEY_HASH : 0u);
        uint32_t& g = is_hash ? b.nodes[node].hash_group : b.nodes[node].own_group;
        if (g == NONE) {
            g = (uint32_t)b.group_count.size();
            b.group_count.push_back(0);
            group_first.push_back((uint32_t)r);
            group_last.push_back((uint32_t)r);
        }
        b.group_count[g]++;
        group_last[g] = (uint32_t)r;
        route_group[r] = g;
    }
    // a group whose ids are one contiguous rank range is stored as that range (the overwhelmingly common case);
    // otherwise (SURVEY.md 8c quirk ii: keys of "x" interleave with keys of "x//...") its ids go to the indirect list
    b.group_begin.resize(b.group_count.size());
    uint32_t ind = 0;
    for (size_t g = 0; g < b.group_count.size(); g++) {
        if (group_last[g] - group_first[g] + 1 == b.group_count[g]) b.group_begin[g] = group_first[g];
        else {
            b.group_begin[g] = ind;
            ind += b.group_count[g];
            b.group_count[g] |= RANGE_INDIRECT;
        }
    }
    b.indirect.assign(ind, 0);
    if (ind) {
        std::vector<uint32_t> cur(b.group_begin);
        for (size_t r = 0; r < n; r++) {
            const uint32_t g = route_group[r];
            if (b.group_count[g] & RANGE_INDIRECT) b.indirect[cur[g]++] = (uint32_t)r;
        }
    }
    b.local_strings.resize(local.entries.size());
    for (size_t i = 0; i < local.entries.size(); i++) b.local_strings[i] = local.entries[i].s;
    if (keep) { // hand the trie over (level strings become owned: the key bytes they point into move on the next merge)
        for (auto& e : local.entries) {
            keep->local_store.emplace_back(e.s);
            e.s = keep->local_store.back();
        }
        keep->local = std::move(local);
        keep->children = std::move(children);
        keep->nodes = b.nodes;
        b.st->inc = keep;
    } else b.st->inc.reset();
}

// Phase 3: place the nodes (global tokens) into the tenant's region: bucketised first-free probing, parents first.
void place_tenant(const TenantBuild& b, SlotBuf& trie) {
    TenantState& t = *b.st;
    const TrieSlot empty_slot{NONE, 0, 0, 0, 0, 0, NONE, 0};
    for (uint32_t s = 0; s < t.cap_slots; s++) trie[t.base + s] = empty_slot;
    std::vector<uint32_t> slot_of(b.nodes.size());
    const uint32_t nb = t.buckets;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        const BuildNode& bn = b.nodes[i];
        const uint32_t pslot = bn.parent == NONE ? ROOT_PARENT : slot_of[bn.parent];
        const uint32_t tok = bn.parent == NONE ? t.token : (bn.token == TOK_PLUS ? TOK_PLUS : b.l2g[bn.token - TOK_FIRST]);
        uint32_t bk = edge_bucket(pslot, tok, nb), s;
        for (;;) {
            s = 2 * bk;
            if (trie[t.base + s].parent == NONE) break;
            if (trie[t.base + ++s].parent == NONE) break;
            bk = (bk + 1 == nb) ? 0 : bk + 1;
        }
        TrieSlot& ts = trie[t.base + s];
        ts.parent = pslot;
        ts.token = tok;
        if (bn.own_group != NONE) {
            ts.own_begin = b.group_begin[bn.own_group];
            ts.own_count = b.group_count[bn.own_group];
        }
        if (bn.hash_group != NONE) {
            ts.hash_begin = b.group_begin[bn.hash_group];
            ts.hash_count = b.group_count[bn.hash_group];
        }
        slot_of[i] = s;
        if (bn.parent != NONE) {
            TrieSlot& pr = trie[t.base + pslot];
            if (tok == TOK_PLUS) pr.plus_child = s;
            else pr.lit_bloom |= 1u << bloom_bit(tok);
        }
    }
    t.root_rel = slot_of[0];
    t.n_nodes = (uint32_t)b.nodes.size();
}

struct PhaseTimer { // BMQ_TIMING=1 prints host build phases to stderr
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = getenv("BMQ_TIMING") != nullptr;
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[bmq build] %-28s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

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

// ------------------------------------------------------------------------------------------------------------
// DistIndexHost
// ------------------------------------------------------------------------------------------------------------
bool DistIndexHost::rebuild(const uint8_t* keys, const uint32_t* key_off, uint32_t n) {
    error.clear();
    by_name.clear();
    order.clear();
    trie.clear();
    next_free = 0;
    dict_h = HostDict();
    strings.clear();
    full_upload = true;
    dict_changed = true;
    dirty.clear();
    PhaseTimer pt;
    // split by tenant: decode in parallel chunks (each thread fills its own per-tenant lists), then concatenate per tenant;
    // KeySet::assign sorts anyway, so the order inside a tenant's list does not matter
    struct Chunk {
        std::map<std::string_view, std::vector<std::string_view>> per;
        uint32_t bad = 0xFFFFFFFFu;
    };
    uint32_t per_chunk = 4096; // BMQ_SPLIT_CHUNK: test knob (tools/host_fuzz runs with tiny chunks)
    if (const char* v = getenv("BMQ_SPLIT_CHUNK")) per_chunk = (uint32_t)std::max(1, atoi(v));
    const uint32_t n_chunks = std::max(1u, std::min(256u, n / per_chunk));
    std::vector<Chunk> chunks(n_chunks);
    parallel_for(n_chunks, [&](size_t c) {
        Chunk& ch = chunks[c];
        const uint32_t lo = (uint32_t)((uint64_t)n * c / n_chunks), hi = (uint32_t)((uint64_t)n * (c + 1) / n_chunks);
        std::string_view last_tenant;
        std::vector<std::string_view>* last_list = nullptr;
        for (uint32_t i = lo; i < hi; i++) {
            const std::string_view k((const char*)keys + key_off[i], key_off[i + 1] - key_off[i]);
            RouteKeyParts kp;
            if (!decode_route_key(k, kp)) {
                ch.bad = i;
                return;
            }
            if (!last_list || kp.tenant != last_tenant) { // sorted input: long runs of one tenant
                last_list = &ch.per[kp.tenant];
                last_tenant = kp.tenant;
            }
            last_list->push_back(k);
        }
    });
    std::map<std::string_view, std::vector<std::vector<std::string_view>*>> per; // tenant -> its lists in the chunks
    for (auto& ch : chunks) {
        if (ch.bad != 0xFFFFFFFFu) {
            error = "malformed route key at position " + std::to_string(ch.bad);
            return false;
        }
        for (auto& e : ch.per) per[e.first].push_back(&e.second);
    }
    std::vector<TenantState*> touched;
    std::vector<std::vector<std::vector<std::string_view>*>*> lists;
    for (auto& e : per) {
        auto st = std::make_unique<TenantState>();
        st->name = std::string(e.first);
        touched.push_back(st.get());
        lists.push_back(&e.second);
        by_name.emplace(st->name, std::move(st));
    }
    pt.lap("split keys by tenant");
    parallel_for(touched.size(), [&](size_t i) {
        std::vector<std::vector<std::string_view>*>& parts = *lists[i];
        if (parts.size() == 1) touched[i]->keys.assign(*parts[0]);
        else {
            std::vector<std::string_view> all;
            size_t total = 0;
            for (auto* p : parts) total += p->size();
            all.reserve(total);
            for (auto* p : parts) all.insert(all.end(), p->begin(), p->end());
            touched[i]->keys.assign(all);
        }
    });
    pt.lap("per-tenant key sets");
    return refresh(touched);
}

bool DistIndexHost::apply(const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n) {
    error.clear();
    PhaseTimer pt;
    std::map<std::string_view, std::vector<std::pair<std::string_view, uint8_t>>> per;
    for (uint32_t i = 0; i < n; i++) {
        const std::string_view k((const char*)keys + key_off[i], key_off[i + 1] - key_off[i]);
        RouteKeyParts kp;
        if (op[i] > 1 || !decode_route_key(k, kp)) {
            error = "malformed route key or op in apply batch";
            return false;
        }
        per[kp.tenant].push_back({k, op[i]});
    }
    std::vector<TenantState*> touched;
    std::vector<std::vector<std::pair<std::string_view, uint8_t>>*> lists;
    for (auto& e : per) {
        auto it = by_name.find(std::string(e.first));
        if (it == by_name.end()) {
            auto st = std::make_unique<TenantState>();
            st->name = std::string(e.first);
            it = by_name.emplace(st->name, std::move(st)).first;
        }
        touched.push_back(it->second.get());
        lists.push_back(&e.second);
    }
    pt.lap("group ops by tenant");
    parallel_for(touched.size(), [&](size_t i) {
        TenantState& t = *touched[i];
        TenantInc* inc = incremental_enabled() ? (TenantInc*)t.inc.get() : nullptr;
        if (!inc) {
            t.keys.apply(*lists[i]);
            t.inc.reset();
            return;
        }
        // EXPERIMENTAL incremental path: merge the keys, carry every surviving key's node along, walk the trie for the new keys only
        std::vector<int64_t> src;
        t.keys.apply(*lists[i], &src);
        std::vector<uint32_t> kn(src.size());
        bool ok = true;
        for (size_t r = 0; r < src.size() && ok; r++) {
            if (src[r] >= 0) kn[r] = inc->key_node[(size_t)src[r]];
            else ok = inc->locate((*lists[i])[(size_t)(-src[r] - 1)].first, kn[r]);
        }
        if (ok) inc->key_node.swap(kn);
        else t.inc.reset(); // cannot happen (keys were decoded above); the full parse reports it
    });
    pt.lap("merge into key sets");
    return refresh(touched);
}

// Rebuilds the regions of the touched tenants, then the (small) global pieces: directory, id bases, route_pos, dictionary.
bool DistIndexHost::refresh(std::vector<TenantState*>& touched) {
    PhaseTimer pt;
    // tenants that lost their last route disappear
    std::vector<TenantState*> live;
    for (TenantState* t : touched) {
        if (t->keys.size() == 0) {
            const std::string name = t->name;
            by_name.erase(name); // its region becomes garbage until the next full rebuild
        } else live.push_back(t);
    }
    // phase 1 (parallel): tries with tenant-local tokens
    std::vector<TenantBuild> builds(live.size());
    for (size_t i = 0; i < live.size(); i++) builds[i].st = live[i];
    parallel_for(builds.size(), [&](size_t i) {
        TenantBuild& b = builds[i];
        if (incremental_enabled() && b.st->inc) build_from_inc(b, *(TenantInc*)b.st->inc.get()); // kept in sync by apply()
        else build_tenant_trie(b);
    });
    for (auto& b : builds)
        if (!b.error.empty()) {
            error = b.error;
            return false;
        }
    pt.lap("phase 1: tries (parallel)");
    // phase 2 (sequential): global dictionary, region allocation
    double region_factor = 2.0; // slots per node (load factor 1/2); BMQ_REGION_FACTOR overrides for experiments
    if (const char* rf = getenv("BMQ_REGION_FACTOR")) region_factor = std::max(1.25, atof(rf));
    const size_t tokens_before = dict_h.entries.size();
    auto intern_owned = [&](std::string_view s) -> uint32_t {
        const size_t before = dict_h.entries.size();
        const uint32_t tok = dict_h.intern(s);
        if (dict_h.entries.size() != before) { // new: give the entry storage that outlives the key bytes
            strings.emplace_back(s);
            dict_h.entries.back().s = strings.back();
        }
        return tok;
    };
    // most level strings of a touched tenant are already in the dictionary: resolve those in parallel (read-only), then
    // intern only the new ones sequentially
    parallel_for(builds.size(), [&](size_t bi) {
        TenantBuild& b = builds[bi];
        b.l2g.resize(b.local_strings.size());
        for (size_t i = 0; i < b.local_strings.size(); i++) b.l2g[i] = dict_h.find(b.local_strings[i]);
    });
    for (auto& b : builds) {
        TenantState& t = *b.st;
        t.token = intern_owned(t.name);
        for (size_t i = 0; i < b.local_strings.size(); i++)
            if (b.l2g[i] == TOK_UNKNOWN) b.l2g[i] = intern_owned(b.local_strings[i]);
        const uint32_t buckets = (uint32_t)std::max<uint64_t>(2, (uint64_t)(b.nodes.size() * region_factor / 2.0) + 1);
        if (2ull * buckets > t.cap_slots) { // (re)allocate the region at the end of the table, with room to grow
            const uint64_t cap = 2ull * buckets + (t.cap_slots ? buckets / 2 : buckets / 4); // head-room: growth stays in place
            if ((uint64_t)next_free + cap >= 0xFFFFFFF0ull) {
                error = "trie too large";
                return false;
            }
            t.base = next_free;
            t.cap_slots = (uint32_t)(cap + (cap & 1));
            next_free += t.cap_slots;
        }
        if (!full_upload) dirty.push_back({t.base, t.cap_slots}); // rewritten in place or freshly allocated
        t.buckets = buckets;
    }
    if (dict_h.entries.size() != tokens_before) dict_changed = true;
    pt.lap("  intern + allocate");
    if (next_free > trie.size()) { // the device table must grow: everything is re-uploaded (slots outside the regions
        // are never read, so plain zero fill is enough; place_tenant initialises every region it owns)
        // the regions of the tenants this call does not touch must survive (after rebuild() the table is empty: nothing to
        // copy).  Not tied to full_upload: that flag is only reset by an upload, which a host-only engine never does.
        const size_t keep = trie.size();
        if (!trie.grow(std::max<size_t>((size_t)next_free + next_free / 4, 64), keep)) {
            error = "out of host memory";
            return false;
        }
        full_upload = true;
        dirty.clear();
    }
    pt.lap("phase 2: dictionary, regions");
    // phase 3 (parallel): placement
    parallel_for(builds.size(), [&](size_t i) {
        place_tenant(builds[i], trie);
        builds[i].st->indirect.swap(builds[i].indirect);
    });
    pt.lap("phase 3: placement (parallel)");
    // global pieces
    order.clear();
    uint64_t rank = 0, rp = 0, nodes_total = 0;
    for (auto& e : by_name) {
        TenantState& t = *e.second;
        t.rank_base = (uint32_t)rank;
        t.rp_base = (uint32_t)rp;
        rank += t.keys.size();
        rp += t.indirect.size();
        nodes_total += t.n_nodes;
        order.push_back(&t);
    }
    if (rank >= 0x7FFFFFF0ull) {
        error = "too many routes";
        return false;
    }
    n_routes = rank;
    n_nodes = nodes_total;
    route_pos.assign(rp ? rp : 1, 0);
    for (TenantState* t : order)
        for (size_t i = 0; i < t->indirect.size(); i++) route_pos[t->rp_base + i] = t->rank_base + t->indirect[i];
    const uint32_t tslots = pow2_at_least((uint64_t)order.size() * 2);
    tenants.assign(tslots, EMPTY_TENANT);
    for (TenantState* t : order) {
        uint32_t d = tenant_hash(t->token) & (tslots - 1);
        while (tenants[d].token) d = (d + 1) & (tslots - 1);
        const TrieSlot& root = trie[t->base + t->root_rel];
        tenants[d] = TenantSlot{t->token, t->root_rel, t->base, t->buckets, t->rank_base, t->rp_base, root.hash_begin, root.hash_count,
                                root.plus_child, root.lit_bloom, {0, 0, 0, 0, 0, 0}};
    }
    if (dict_changed) flatten_dict(dict_h, dict, pool);
    if (trie.empty()) trie.grow(64, 0);
    pt.lap("directory, ids, dictionary");
    return true;
}

std::string_view DistIndexHost::route_key(uint32_t id) const {
    if (id >= n_routes || order.empty()) return {};
    size_t lo = 0, hi = order.size(); // last tenant with rank_base <= id
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (order[mid]->rank_base <= id) lo = mid;
        else hi = mid;
    }
    return order[lo]->keys.key(id - order[lo]->rank_base);
}

uint32_t DistIndexHost::find_child(const TenantState& t, uint32_t parent_rel, uint32_t token) const {
    uint32_t bk = edge_bucket(parent_rel, token, t.buckets);
    for (uint32_t probes = 0; probes < t.buckets; probes++) {
        bool full = true;
        for (uint32_t j = 0; j < 2; j++) {
            const TrieSlot& s = trie[t.base + 2 * bk + j];
            if (s.parent == NONE) full = false;
            else if (s.parent == parent_rel && s.token == token) return 2 * bk + j;
        }
        if (!full) return NONE;
        bk = (bk + 1 == t.buckets) ? 0 : bk + 1;
    }
    return NONE;
}

std::vector<uint32_t> DistIndexHost::find_filter(std::string_view tenant, std::string_view filter) const {
    std::vector<uint32_t> out;
    auto it = by_name.find(std::string(tenant));
    if (it == by_name.end()) return out;
    const TenantState& t = *it->second;
    uint32_t slot = t.root_rel;
    bool is_hash = false;
    size_t s = 0;
    for (size_t i = 0; i <= filter.size() && slot != NONE; i++)
        if (i == filter.size() || filter[i] == '/') {
            const std::string_view lv = filter.substr(s, i - s);
            s = i + 1;
            if (lv == "#" && i == filter.size()) {
                is_hash = true;
                break;
            }
            const uint32_t tok = lv == "+" ? TOK_PLUS : dict_find(dict, pool, lv);
            if (tok == TOK_UNKNOWN) return out;
            slot = find_child(t, slot, tok);
        }
    if (slot == NONE) return out;
    const TrieSlot& n = trie[t.base + slot];
    const uint32_t b = is_hash ? n.hash_begin : n.own_begin, cf = is_hash ? n.hash_count : n.own_count;
    const uint32_t c = cf & ~RANGE_INDIRECT;
    for (uint32_t i = 0; i < c; i++) out.push_back((cf & RANGE_INDIRECT) ? route_pos[t.rp_base + b + i] : t.rank_base + b + i);
    return out;
}

} // namespace bmq
