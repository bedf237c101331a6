// bmq_cache.cpp -- the route cache in front of the batching front: the engine-side counterpart of
//   ISubscriptionCache            DW/cache/ISubscriptionCache.java:30-40    get / isCached / refresh / reset
//   SubscriptionCache             DW/cache/SubscriptionCache.java:117-146   tenant -> TenantRouteCache
//   TenantRouteCache              DW/cache/TenantRouteCache.java:116-296    topic -> matched routes, loads through matchAll(singleton),
//                                                                            Add/RemoveRoutes tasks patch the entries index.match(filter) finds
//   TopicIndex                    DW/TopicIndex.java:39-156                 trie of the CACHED topics, queried with a topic filter
// Plain C++ over the C ABI (include/bmq.h): a miss goes through bmq_batcher_match_all (one GPU launch for everything missing right
// now), a hit never leaves the host.  Route mutations go to the engine (bmq_routes_apply) and then drop every cached topic the
// mutated filters match (the reference patches those entries in place; an invalidated entry reloads to the same set on its next
// access, which is what the reference's own reload computes).
//
// What keeps a stale result out of the cache: a load carries the engine epoch it was matched at; every mutation is logged per tenant
// with the epoch it produced.  An insert first checks the log for mutations newer than its match epoch whose filter matches the topic
// (the reference serialises loads and patches of a tenant in one task loop, TenantRouteCache.java:208-212 -- same effect, no queue).
//
// Fan-out caps: what the cache stores and serves is IMatchedRoutes.routes() (TenantRouteCache.java:299-301), i.e. the row AFTER
// MatchedRoutes' persistent / group fan-out caps (DW/cache/MatchedRoutes.java:87-141), applied in KV key order by bmq_routes_cap when
// the row is loaded; the throttle events of a load go to the event sink (IEventCollector.report).  The caps are a per-tenant setting
// (MaxPersistentFanout / MaxGroupFanout, asked per tenant in TenantRouteCache.java:174-175); an entry remembers the caps it was
// capped with and a hit under different caps follows MatchedRoutes.adjust (:150-200): reload when a raised cap could admit more or
// a lowered one is exceeded, otherwise just adopt the new caps.  The weigher counts the capped row (:108).
//
// Lifecycle: SubscriptionCache.java:79-107 drops a tenant's whole cache after 2 x the match expiry without a get(); here
// bmq_route_cache_expire unlinks and frees such tenants.  Lookups take no lock on the tenant table: they announce themselves in a
// striped reader count (a "big-reader" lock); the sweep waits for the readers that were inside when it unlinked.
#include "../../include/bmq.h"
#include "bmq_codec.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace bmq {
namespace cache {

inline std::vector<std::string_view> split(std::string_view s, char sep) { // TopicUtil.parse(topic, false): empty levels kept
    std::vector<std::string_view> out;
    size_t b = 0;
    for (size_t i = 0; i <= s.size(); i++)
        if (i == s.size() || s[i] == sep) {
            out.push_back(s.substr(b, i - b));
            b = i + 1;
        }
    return out;
}

// SURVEY.md 8a-0 on level lists: does `filter` match `topic`?
inline bool filter_matches(const std::vector<std::string_view>& f, const std::vector<std::string_view>& t) {
    const bool sys = !t.empty() && !t[0].empty() && t[0][0] == '$';
    for (size_t i = 0; i < f.size(); i++) {
        if (f[i] == "#" && i + 1 == f.size()) return !(i == 0 && sys);
        if (i >= t.size()) return false;
        if (f[i] == "+") {
            if (i == 0 && sys) return false;
            continue;
        }
        if (f[i] != t[i]) return false;
    }
    return f.size() == t.size();
}

struct Entry {
    std::string topic;
    std::vector<uint32_t> ids; // ascending
    uint64_t hash = 0;         // hash64(topic)
    uint64_t epoch = 0;        // engine epoch of the match
    uint64_t last_access_ms = 0;
    int32_t max_pf = INT_MAX, max_gf = INT_MAX; // the caps `ids` was capped with (MatchedRoutes.maxPersistentFanout / maxGroupFanout)
    uint32_t n_pers = ~0u, n_grp = ~0u;         // persistent / group routes among ids; ~0: not classified (at most ids.size() each)
    Entry* next_same_hash = nullptr;
    std::list<Entry*>::iterator lru;
    uint64_t weight() const { return ids.empty() ? 1 : ids.size(); } // TenantRouteCache.java:108-111
};

// TopicIndex (DW/TopicIndex.java:39-156): trie of the cached topics; match(filter) walks it as TopicMatcher's selector does.
// Children hang in a hash table keyed by the 64-bit hash of the level (chains through Node::next): a lookup hashes the
// string_view and compares labels, nothing is allocated per level.
inline uint64_t level_hash64(std::string_view s) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (unsigned char ch : s) h = (h ^ ch) * 0x100000001B3ull;
    return h ^ (h >> 29);
}
struct TopicIndex {
    struct Node {
        std::string label;
        std::unordered_map<uint64_t, Node*> children; // level hash -> chain
        Node* next = nullptr;                         // same hash, other label
        uint32_t n_children = 0;
        Entry* value = nullptr;
        ~Node() {
            for (auto& c : children)
                for (Node* n = c.second; n;) {
                    Node* nx = n->next;
                    delete n;
                    n = nx;
                }
        }
        Node* child(std::string_view lv) const {
            auto it = children.find(level_hash64(lv));
            for (Node* n = it == children.end() ? nullptr : it->second; n; n = n->next)
                if (n->label == lv) return n;
            return nullptr;
        }
        template <class F> void each_child(F&& f) const {
            for (auto& c : children)
                for (Node* n = c.second; n; n = n->next) f(*n);
        }
    };
    Node root;
    void add(const std::vector<std::string_view>& levels, Entry* e) {
        Node* n = &root;
        for (auto lv : levels) {
            Node* c = n->child(lv);
            if (!c) {
                c = new Node();
                c->label = std::string(lv);
                Node*& head = n->children[level_hash64(lv)];
                c->next = head;
                head = c;
                n->n_children++;
            }
            n = c;
        }
        n->value = e;
    }
    void remove(const std::vector<std::string_view>& levels) { remove_at(root, levels, 0); }
    template <class F> void match(const std::vector<std::string_view>& filter, F&& f) const { walk(root, filter, 0, f); }

private:
    static bool remove_at(Node& n, const std::vector<std::string_view>& levels, size_t i) { // true: n is empty now
        if (i == levels.size()) n.value = nullptr;
        else {
            auto it = n.children.find(level_hash64(levels[i]));
            if (it != n.children.end())
                for (Node** pp = &it->second; *pp; pp = &(*pp)->next)
                    if ((*pp)->label == levels[i]) {
                        if (remove_at(**pp, levels, i + 1)) {
                            Node* dead = *pp;
                            *pp = dead->next;
                            delete dead;
                            n.n_children--;
                            if (!it->second) n.children.erase(it);
                        }
                        break;
                    }
        }
        return !n.value && n.n_children == 0;
    }
    template <class F> static void all_below(const Node& n, F& f) {
        if (n.value) f(n.value);
        n.each_child([&](const Node& c) { all_below(c, f); });
    }
    template <class F> static void walk(const Node& n, const std::vector<std::string_view>& flt, size_t i, F& f) {
        if (i == flt.size()) {
            if (n.value) f(n.value);
            return;
        }
        const std::string_view lv = flt[i];
        const bool last = i + 1 == flt.size();
        if (lv == "#" && last) { // the node itself ("a/#" matches "a") and everything below; at level 0 not the '$' topics
            if (i > 0 && n.value) f(n.value);
            n.each_child([&](const Node& c) {
                if (!(i == 0 && !c.label.empty() && c.label[0] == '$')) all_below(c, f);
            });
            return;
        }
        if (lv == "+") {
            n.each_child([&](const Node& c) {
                if (!(i == 0 && !c.label.empty() && c.label[0] == '$')) walk(c, flt, i + 1, f);
            });
            return;
        }
        if (const Node* c = n.child(lv)) walk(*c, flt, i + 1, f);
    }
};

struct Mutation {
    uint64_t epoch;
    std::vector<std::string> filter;
};

inline uint64_t hash64(std::string_view s, uint64_t h = 0xCBF29CE484222325ull) {
    for (unsigned char ch : s) h = (h ^ ch) * 0x100000001B3ull;
    h ^= h >> 32;
    h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    return h;
}

// Hits come from hundreds of matcher threads and hold the lock for ~100 ns: spin briefly, then sleep in the kernel (glibc's adaptive
// mutex).  Pure spinning collapses when the threads outnumber the CPUs the process may use (a container's CPU quota: the GPU boxes of
// this project show 256 CPUs and grant 16), a plain mutex pays a futex hand-over for every short wait.
struct Spin {
    pthread_mutex_t m;
    Spin() {
        pthread_mutexattr_t a;
        pthread_mutexattr_init(&a);
        pthread_mutexattr_settype(&a, PTHREAD_MUTEX_ADAPTIVE_NP);
        pthread_mutex_init(&m, &a);
        pthread_mutexattr_destroy(&a);
    }
    ~Spin() { pthread_mutex_destroy(&m); }
    Spin(const Spin&) = delete;
    Spin& operator=(const Spin&) = delete;
    void lock() { pthread_mutex_lock(&m); }
    void unlock() { pthread_mutex_unlock(&m); }
};

// One slice of a tenant's cache: the topics whose hash falls into it, their LRU order, their trie.  A tenant has several, so that
// the publishes of one hot tenant do not queue behind one lock.  The weight bound is the TENANT's (TenantCache::weight); a shard
// publishes the access time of its least recently used entry so that an over-weight tenant evicts from the shard holding the oldest.
struct alignas(64) Shard {
    Spin mu;
    std::unordered_map<uint64_t, Entry*> table; // topic hash -> chain (next_same_hash)
    std::list<Entry*> lru;                      // front = most recently used
    uint64_t weight = 0;
    std::atomic<uint64_t> oldest_ms{~0ull};     // last access of lru.back(), ~0 when empty (read without the lock by the evictor)
    TopicIndex index;
    uint64_t hits = 0, misses = 0, evictions = 0, invalidations = 0, stale_loads = 0, expired = 0, entries = 0;

    void note_oldest() { oldest_ms.store(lru.empty() ? ~0ull : lru.back()->last_access_ms, std::memory_order_relaxed); }
    Entry* find(uint64_t h, std::string_view topic) const {
        auto it = table.find(h);
        for (Entry* en = it == table.end() ? nullptr : it->second; en; en = en->next_same_hash)
            if (en->topic == topic) return en;
        return nullptr;
    }
    void insert(Entry* en) {
        Entry*& head = table[en->hash];
        en->next_same_hash = head;
        head = en;
        lru.push_front(en);
        en->lru = lru.begin();
        weight += en->weight();
        index.add(split(en->topic, '/'), en);
        entries++;
        note_oldest();
    }
    uint64_t drop(Entry* en) { // returns the weight released
        const uint64_t w = en->weight();
        index.remove(split(en->topic, '/'));
        weight -= w;
        lru.erase(en->lru);
        auto it = table.find(en->hash);
        Entry** pp = &it->second;
        while (*pp != en) pp = &(*pp)->next_same_hash;
        *pp = en->next_same_hash;
        if (!it->second) table.erase(it);
        entries--;
        delete en;
        note_oldest();
        return w;
    }
    uint64_t clear() {
        uint64_t w = 0;
        while (!lru.empty()) w += drop(lru.back());
        return w;
    }
    ~Shard() { clear(); }
};

struct TenantCache { // TenantRouteCache
    std::string name;
    uint64_t hash = 0;
    std::atomic<TenantCache*> next{nullptr}; // chain of the tenant table bucket
    std::unique_ptr<Shard[]> shards;
    std::atomic<uint64_t> weight{0};         // sum of the shards' weights (the maximumWeight bound is the tenant's, TenantRouteCache.java:104-111)
    std::atomic<uint64_t> last_get_ms{0};    // SubscriptionCache.get refreshes the tenant's expiry (refreshExpiry), isCached / refresh do not
    std::atomic<int32_t> max_pf{INT_MAX}, max_gf{INT_MAX}; // the tenant's MaxPersistentFanout / MaxGroupFanout
    std::mutex log_mu;
    std::vector<Mutation> log; // ascending epochs; complete for epochs > log_floor
    uint64_t log_floor = 0;
};

} // namespace cache
} // namespace bmq

using namespace bmq;
using namespace bmq::cache;

struct bmq_route_cache {
    bmq_engine* e = nullptr;
    bmq_batcher* b = nullptr;
    uint64_t max_routes_per_tenant = 200000; // DistMaxCachedRoutesPerTenant
    uint64_t expiry_ms = 60000;              // DistTopicMatchExpirySeconds
    uint64_t tenant_idle_ms = 120000;        // SubscriptionCache.java:79: 2 x expiry without a get() and the tenant's cache is destroyed
    uint32_t log_keep = 4096;
    uint32_t n_shards = 16;                  // per tenant, power of two
    uint32_t direct_batch = 8192;            // get_batch: requests of at least this many topics are matched without the cache
    int32_t def_pf = INT_MAX, def_gf = 100;  // Setting.java:60-61: MaxPersistentFanout, MaxGroupFanout defaults
    // tenant table: chained hash with atomic bucket heads -- a lookup takes no lock and writes nothing but its reader stripe
    static constexpr uint32_t TBUCKETS = 1u << 14;
    std::unique_ptr<std::atomic<TenantCache*>[]> buckets;
    std::mutex table_mu; // writers of the table (publish / unlink), one at a time
    std::mutex sweep_mu; // one tenant sweep at a time (it drains the readers BEFORE it takes table_mu: a reader may be waiting for that one)
    std::mutex apply_mu; // one refresh at a time, in commit order (ISubscriptionCache.refresh comes from the range's apply thread)
    std::atomic<uint64_t> created_floor{0};
    std::atomic<bool> bypass{false}; // a rebuild is replacing the index: serve nothing from the cache, cache nothing
    std::atomic<uint64_t> async_inflight{0}; // misses of get_async still with the batching front
    std::atomic<uint64_t> cold_misses{0};    // gets for a tenant that has no cache yet
    // per-tenant caps that differ from the defaults (ISettingProvider.provide(MaxPersistentFanout / MaxGroupFanout, tenantId))
    std::mutex caps_mu;
    std::unordered_map<std::string, std::pair<int32_t, int32_t>> caps;
    // the event sink: callback and user pointer travel as ONE immutable pair behind one atomic pointer (round 4 kept two atomics: a load
    // between the two stores of a replacement could pair the new callback with the old user pointer); a pair is never freed before the
    // cache is -- a load that read it just before a replacement may still call it
    struct EvSink {
        bmq_route_cache_event_cb cb;
        void* user;
    };
    std::atomic<const EvSink*> ev{nullptr};
    std::mutex ev_mu;
    std::vector<std::unique_ptr<EvSink>> ev_all;
    // counters of tenants whose cache has been destroyed (the cache-wide statistics keep counting them)
    std::mutex retired_mu;
    bmq_route_cache_stats retired{};
    std::atomic<uint64_t> tenants_live{0}, tenants_expired{0};
    // big-reader lock: a reader bumps the stripe of its thread while it holds a TenantCache*; the sweep raises `sweeping`, waits for
    // every stripe to drain, unlinks and frees.  Readers never hold a stripe across a call into the batching front.
    static constexpr uint32_t STRIPES = 64;
    struct alignas(64) Stripe {
        std::atomic<int64_t> n{0};
    };
    Stripe stripes[STRIPES];
    std::atomic<bool> sweeping{false};

    bmq_route_cache() : buckets(new std::atomic<TenantCache*>[TBUCKETS]) {
        for (uint32_t i = 0; i < TBUCKETS; i++) buckets[i].store(nullptr, std::memory_order_relaxed);
    }
    ~bmq_route_cache() {
        for (uint32_t i = 0; i < TBUCKETS; i++)
            for (TenantCache* t = buckets[i].load(); t;) {
                TenantCache* nx = t->next.load();
                delete t;
                t = nx;
            }
    }
    static uint32_t my_stripe() {
        static std::atomic<uint32_t> next{0};
        thread_local uint32_t s = next.fetch_add(1, std::memory_order_relaxed) % STRIPES;
        return s;
    }
    void rd_enter(uint32_t st) {
        for (;;) {
            stripes[st].n.fetch_add(1, std::memory_order_seq_cst);
            if (!sweeping.load(std::memory_order_seq_cst)) return;
            stripes[st].n.fetch_sub(1, std::memory_order_seq_cst);
            while (sweeping.load(std::memory_order_acquire)) std::this_thread::yield();
        }
    }
    void rd_exit(uint32_t st) { stripes[st].n.fetch_sub(1, std::memory_order_release); }
    TenantCache* find(std::string_view tenant, uint64_t h) const { // reader stripe (or table_mu) held
        for (TenantCache* t = buckets[h & (TBUCKETS - 1)].load(std::memory_order_seq_cst); t; t = t->next.load(std::memory_order_acquire))
            if (t->hash == h && t->name == tenant) return t;
        return nullptr;
    }
    std::pair<int32_t, int32_t> caps_of(std::string_view tenant) {
        std::lock_guard<std::mutex> g(caps_mu);
        if (!caps.empty()) {
            auto it = caps.find(std::string(tenant));
            if (it != caps.end()) return it->second;
        }
        return {def_pf, def_gf};
    }
    TenantCache* obtain(std::string_view tenant, uint64_t now_ms) { // reader stripe held
        const uint64_t h = hash64(tenant);
        if (TenantCache* t = find(tenant, h)) return t;
        auto nt = std::make_unique<TenantCache>();
        nt->name = std::string(tenant);
        nt->hash = h;
        nt->shards.reset(new Shard[n_shards]);
        nt->last_get_ms.store(now_ms, std::memory_order_relaxed);
        TenantCache* t;
        {
            std::lock_guard<std::mutex> g(table_mu);
            if ((t = find(tenant, h))) return t; // somebody else published it meanwhile
            std::atomic<TenantCache*>& head = buckets[h & (TBUCKETS - 1)];
            nt->next.store(head.load(std::memory_order_relaxed), std::memory_order_relaxed);
            t = nt.release();
            head.store(t, std::memory_order_seq_cst);
            tenants_live.fetch_add(1, std::memory_order_relaxed);
        }
        // Mutations applied before the tenant's cache was visible were never logged for it.  The floor is read AFTER the publication
        // (both seq_cst): a bmq_route_cache_apply that did not see the tenant had stored its epoch before we read -- either apply
        // finds the tenant, or the tenant starts above apply's epoch; a load matched before it is then not cached.
        {
            std::lock_guard<std::mutex> lg(t->log_mu);
            t->log_floor = std::max(t->log_floor, created_floor.load(std::memory_order_seq_cst));
        }
        { // the same hand-shake for the caps: set_caps updates the map, then the live tenant, under caps_mu
            std::lock_guard<std::mutex> g(caps_mu);
            auto it = caps.empty() ? caps.end() : caps.find(t->name);
            t->max_pf.store(it == caps.end() ? def_pf : it->second.first, std::memory_order_relaxed);
            t->max_gf.store(it == caps.end() ? def_gf : it->second.second, std::memory_order_relaxed);
        }
        return t;
    }
    template <class F> void for_each_tenant(F&& f) { // reader stripe (or table_mu) held
        for (uint32_t i = 0; i < TBUCKETS; i++)
            for (TenantCache* t = buckets[i].load(std::memory_order_acquire); t; t = t->next.load(std::memory_order_acquire)) f(*t);
    }
    Shard& shard_of(TenantCache& t, uint64_t topic_hash) const { return t.shards[(topic_hash >> 40) & (n_shards - 1)]; }
    void touch(TenantCache& t, uint64_t now_ms) const { // written only when it moved on noticeably: the line stays shared among the getters
        const uint64_t last = t.last_get_ms.load(std::memory_order_relaxed);
        if (now_ms > last && now_ms - last >= std::max<uint64_t>(1, tenant_idle_ms / 16)) t.last_get_ms.store(now_ms, std::memory_order_relaxed);
    }
};

namespace {
struct SpinGuard {
    Spin& s;
    explicit SpinGuard(Spin& sp) : s(sp) { s.lock(); }
    ~SpinGuard() { s.unlock(); }
};
struct ReadGuard { // while alive, TenantCache pointers found in the table stay valid
    bmq_route_cache* c;
    uint32_t st;
    bool in = false;
    explicit ReadGuard(bmq_route_cache* rc) : c(rc), st(bmq_route_cache::my_stripe()) { enter(); }
    ~ReadGuard() { leave(); }
    void enter() {
        if (!in) c->rd_enter(st), in = true;
    }
    void leave() {
        if (in) c->rd_exit(st), in = false;
    }
};
} // namespace

extern "C" {

int bmq_route_cache_create(bmq_engine* e, bmq_batcher* b, const bmq_route_cache_config* cfg, bmq_route_cache** out) {
    if (!e || !b || !out) return BMQ_E_INVAL;
    *out = nullptr;
    auto c = std::make_unique<bmq_route_cache>();
    c->e = e;
    c->b = b;
    if (cfg) {
        if (cfg->struct_size < 8 || cfg->struct_size > sizeof(bmq_route_cache_config)) return BMQ_E_INVAL;
        bmq_route_cache_config k{};
        memcpy(&k, cfg, cfg->struct_size);
        if (k.max_routes_per_tenant) c->max_routes_per_tenant = k.max_routes_per_tenant;
        if (k.expiry_ms) c->expiry_ms = k.expiry_ms;
        if (k.mutation_log_entries) c->log_keep = k.mutation_log_entries;
        if (k.direct_batch_topics) c->direct_batch = (uint32_t)std::min<uint64_t>(k.direct_batch_topics, 0xFFFFFFFFull);
        if (k.shards_per_tenant) {
            if (k.shards_per_tenant > 1024 || (k.shards_per_tenant & (k.shards_per_tenant - 1))) return BMQ_E_INVAL;
            c->n_shards = (uint32_t)k.shards_per_tenant;
        }
        if (k.default_max_persistent_fanout < 0 || k.default_max_group_fanout < 0) return BMQ_E_INVAL;
        if (k.default_max_persistent_fanout) c->def_pf = k.default_max_persistent_fanout;
        if (k.default_max_group_fanout) c->def_gf = k.default_max_group_fanout;
        c->tenant_idle_ms = k.tenant_idle_ms ? k.tenant_idle_ms : 2 * c->expiry_ms;
    } else c->tenant_idle_ms = 2 * c->expiry_ms;
    bmq_index_info info{};
    if (bmq_index_info_get(e, &info) == BMQ_OK) c->created_floor = info.epoch;
    *out = c.release();
    return BMQ_OK;
}

void bmq_route_cache_destroy(bmq_route_cache* c) {
    if (!c) return;
    // futures still on their way complete first (the batcher's dispatcher thread calls them back): it outlives the cache
    while (c->async_inflight.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    delete c;
}

int bmq_route_cache_set_caps(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, int32_t max_pf, int32_t max_gf) {
    if (!c || (tenant_len && !tenant) || max_pf < 0 || max_gf < 0) return BMQ_E_INVAL;
    const std::string_view tn((const char*)tenant, tenant_len);
    ReadGuard rg(c);
    std::lock_guard<std::mutex> g(c->caps_mu);
    if (max_pf == c->def_pf && max_gf == c->def_gf) c->caps.erase(std::string(tn));
    else c->caps[std::string(tn)] = {max_pf, max_gf};
    if (TenantCache* t = c->find(tn, hash64(tn))) {
        t->max_pf.store(max_pf, std::memory_order_relaxed);
        t->max_gf.store(max_gf, std::memory_order_relaxed);
    }
    return BMQ_OK;
}

int bmq_route_cache_set_event_sink(bmq_route_cache* c, bmq_route_cache_event_cb cb, void* user) {
    if (!c) return BMQ_E_INVAL;
    const bmq_route_cache::EvSink* next = nullptr;
    if (cb) {
        auto p = std::make_unique<bmq_route_cache::EvSink>(bmq_route_cache::EvSink{cb, user});
        next = p.get();
        std::lock_guard<std::mutex> g(c->ev_mu);
        c->ev_all.push_back(std::move(p));
    }
    c->ev.store(next, std::memory_order_release);
    return BMQ_OK;
}

namespace {
struct Caps {
    int32_t pf, gf;
};
struct RowClass {
    uint32_t n_pers = ~0u, n_grp = ~0u;
};
// MatchedRoutes over freshly matched rows (CSR rp / ids, rewritten in place: rows only shrink): row r belongs to tenant row_tenant[r]
// (an index into caps / the tenant table of the call) and to topic row_topic[r] (for the events).  Rows no longer than either cap are
// left alone -- nothing to look up; the others go through bmq_routes_cap, one call per tenant that has such rows.
int cap_rows_inplace(bmq_route_cache* c, const Caps* caps, const uint8_t* tenants, const uint32_t* tenant_off, const uint32_t* row_tenant,
                     const uint8_t* topics, const uint32_t* topic_off, const uint32_t* row_topic, uint32_t n_rows, uint32_t* rp, uint32_t* ids,
                     RowClass* cls /* may be null */) {
    std::vector<uint32_t> lng; // rows a cap can bind on
    for (uint32_t r = 0; r < n_rows; r++) {
        const Caps k = caps[row_tenant ? row_tenant[r] : 0];
        if ((int64_t)(rp[r + 1] - rp[r]) > std::min<int64_t>(k.pf, k.gf)) lng.push_back(r);
    }
    if (lng.empty()) return BMQ_OK;
    std::stable_sort(lng.begin(), lng.end(), [&](uint32_t x, uint32_t y) { return (row_tenant ? row_tenant[x] : 0) < (row_tenant ? row_tenant[y] : 0); });
    std::vector<uint32_t> new_len(lng.size());
    std::vector<uint32_t> kept_all; // the capped rows, in lng order
    std::vector<uint32_t> s_rp, s_ids, o_rp, o_ids, o_cls;
    std::vector<int32_t> evs;
    const bmq_route_cache::EvSink* const sink = c->ev.load(std::memory_order_acquire);
    const bmq_route_cache_event_cb cb = sink ? sink->cb : nullptr;
    void* const cb_user = sink ? sink->user : nullptr;
    for (size_t a = 0; a < lng.size();) {
        const uint32_t ti = row_tenant ? row_tenant[lng[a]] : 0;
        size_t z = a;
        s_rp.assign(1, 0);
        s_ids.clear();
        for (; z < lng.size() && (row_tenant ? row_tenant[lng[z]] : 0) == ti; z++) {
            s_ids.insert(s_ids.end(), ids + rp[lng[z]], ids + rp[lng[z] + 1]);
            s_rp.push_back((uint32_t)s_ids.size());
        }
        const uint32_t m = (uint32_t)(z - a);
        o_rp.resize((size_t)m + 1);
        o_ids.resize(s_ids.size());
        o_cls.resize(2 * (size_t)m);
        evs.resize(4 * s_ids.size());
        uint32_t n_ev = 0;
        const int rc = bmq_routes_cap(c->e, s_rp.data(), s_ids.data(), m, caps[ti].pf, caps[ti].gf, o_rp.data(), o_ids.data(), o_cls.data(), evs.data(),
                                      (uint32_t)s_ids.size(), &n_ev);
        if (rc != BMQ_OK) return rc;
        for (uint32_t k = 0; k < m; k++) {
            new_len[a + k] = o_rp[k + 1] - o_rp[k];
            kept_all.insert(kept_all.end(), o_ids.begin() + o_rp[k], o_ids.begin() + o_rp[k + 1]);
            if (cls) cls[lng[a + k]] = RowClass{o_cls[2 * k], o_cls[2 * k + 1]};
        }
        if (cb) // IEventCollector.report(PersistentFanoutThrottled / GroupFanoutThrottled), MatchedRoutes.java:95-101,124-130
            for (uint32_t k = 0; k < n_ev; k++) {
                const uint32_t r = lng[a + (size_t)evs[4 * k + 1]], tp = row_topic ? row_topic[r] : r;
                cb(cb_user, tenants + tenant_off[ti], tenant_off[ti + 1] - tenant_off[ti], topics + topic_off[tp], topic_off[tp + 1] - topic_off[tp], evs[4 * k],
                   (uint32_t)evs[4 * k + 2], evs[4 * k + 3]);
            }
        a = z;
    }
    // compact the CSR front to back (a write position never passes its read position: rows only shrink)
    std::vector<uint32_t> order(lng.size());
    for (size_t k = 0; k < lng.size(); k++) order[k] = (uint32_t)k;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return lng[x] < lng[y]; });
    std::vector<uint64_t> kept_off(lng.size() + 1, 0);
    for (size_t k = 0; k < lng.size(); k++) kept_off[k + 1] = kept_off[k] + new_len[k];
    size_t nx = 0;
    uint32_t w = 0, rd = rp[0];
    for (uint32_t r = 0; r < n_rows; r++) {
        const uint32_t len = rp[r + 1] - rd, beg = rd;
        rd = rp[r + 1];
        rp[r] = w;
        if (nx < order.size() && lng[order[nx]] == r) {
            const uint32_t k = order[nx++];
            if (new_len[k]) memcpy(ids + w, kept_all.data() + kept_off[k], (size_t)new_len[k] * 4);
            w += new_len[k];
        } else {
            if (w != beg && len) memmove(ids + w, ids + beg, (size_t)len * 4);
            w += len;
        }
    }
    rp[n_rows] = w;
    return BMQ_OK;
}

// Evict least recently used entries until the tenant fits its weight bound again.  No shard lock may be held by the caller.
void enforce_budget(bmq_route_cache* c, TenantCache* t) {
    for (uint32_t guard = 0; t->weight.load(std::memory_order_relaxed) > c->max_routes_per_tenant && guard < (1u << 20); guard++) {
        uint32_t best = 0;
        uint64_t best_ms = ~0ull;
        for (uint32_t s = 0; s < c->n_shards; s++) {
            const uint64_t o = t->shards[s].oldest_ms.load(std::memory_order_relaxed);
            if (o < best_ms) best_ms = o, best = s;
        }
        if (best_ms == ~0ull) return; // nothing cached
        Shard& sh = t->shards[best];
        SpinGuard g(sh.mu);
        if (sh.lru.empty()) continue;
        if (t->weight.load(std::memory_order_relaxed) <= c->max_routes_per_tenant) return;
        sh.evictions++;
        t->weight.fetch_sub(sh.drop(sh.lru.back()), std::memory_order_relaxed);
    }
}

// A loaded (and capped) row goes into the cache unless a mutation that could change it has been applied since it was matched.
void store_loaded(bmq_route_cache* c, TenantCache* t, Shard& sh, std::string_view tp, uint64_t th, std::vector<uint32_t>&& ids, uint64_t epoch,
                  uint64_t now_ms, Caps caps, RowClass cls, bool cold) {
    auto en = std::make_unique<Entry>(); // built outside the lock
    en->topic = std::string(tp);
    en->ids = std::move(ids);
    en->hash = th;
    en->epoch = epoch;
    en->last_access_ms = now_ms;
    en->max_pf = caps.pf, en->max_gf = caps.gf;
    en->n_pers = cls.n_pers, en->n_grp = cls.n_grp;
    const auto tl = split(tp, '/');
    {
        SpinGuard g(sh.mu);
        if (cold) { // the miss was counted cache-wide because the tenant had no cache yet: it is the tenant's (MqttRouteCacheMissCount)
            sh.misses++;
            c->cold_misses.fetch_sub(1, std::memory_order_relaxed);
        }
        bool stale;
        { // shard lock, then log lock: a mutation logs first and invalidates the shards afterwards, so it either shows up here or finds the entry
            std::lock_guard<std::mutex> lg(t->log_mu);
            stale = epoch < t->log_floor; // mutations in (epoch, log_floor] are unknown here
            for (size_t k = t->log.size(); !stale && k-- > 0 && t->log[k].epoch > epoch;) {
                std::vector<std::string_view> fl(t->log[k].filter.begin(), t->log[k].filter.end());
                stale = filter_matches(fl, tl);
            }
        }
        if (stale) { // correct as of its epoch (the caller gets it), but not what the next caller should see
            sh.stale_loads++;
            return;
        }
        Entry* have = sh.find(th, tp);
        if (have && have->epoch >= epoch && have->max_pf == caps.pf && have->max_gf == caps.gf) return; // another thread loaded the same topic meanwhile
        if (have) t->weight.fetch_sub(sh.drop(have), std::memory_order_relaxed);
        t->weight.fetch_add(en->weight(), std::memory_order_relaxed);
        sh.insert(en.release());
    }
    if (t->weight.load(std::memory_order_relaxed) > c->max_routes_per_tenant) enforce_budget(c, t); // maximumWeight: least recently used first
}
// MatchedRoutes.adjust (MatchedRoutes.java:150-200) for an entry met under caps other than the ones it was capped with: true = the
// entry stays (it now carries the new caps), false = it has to be re-matched (a raised cap could admit routes the old one threw away;
// a lowered cap is exceeded -- the reference removes arbitrary routes there, a reload removes the last ones in key order).
bool adjust_entry(Entry* en, Caps now) {
    const uint32_t ub = (uint32_t)en->ids.size();
    const uint32_t pers = en->n_pers == ~0u ? ub : en->n_pers, grp = en->n_grp == ~0u ? ub : en->n_grp; // unknown: at most the row
    if (en->max_pf < now.pf && (int64_t)pers >= (int64_t)en->max_pf) return false;
    if (en->max_gf < now.gf && (int64_t)grp >= (int64_t)en->max_gf) return false;
    if (en->max_pf > now.pf && (int64_t)pers > (int64_t)now.pf) return false;
    if (en->max_gf > now.gf && (int64_t)grp > (int64_t)now.gf) return false;
    en->max_pf = now.pf, en->max_gf = now.gf;
    return true;
}
// a live entry, touched -- or nullptr (an expired one, or one the tenant's new caps invalidate, is dropped on the way).  Shard lock held.
Entry* lookup_live(bmq_route_cache* c, TenantCache* t, Shard& sh, std::string_view tp, uint64_t th, uint64_t now_ms, Caps caps) {
    Entry* en = sh.find(th, tp);
    if (!en) return nullptr;
    if (now_ms >= en->last_access_ms && now_ms - en->last_access_ms >= c->expiry_ms) { // expireAfterAccess
        sh.expired++;
        t->weight.fetch_sub(sh.drop(en), std::memory_order_relaxed);
        return nullptr;
    }
    if ((en->max_pf != caps.pf || en->max_gf != caps.gf) && !adjust_entry(en, caps)) {
        t->weight.fetch_sub(sh.drop(en), std::memory_order_relaxed);
        return nullptr;
    }
    sh.hits++;
    en->last_access_ms = now_ms;
    if (en->lru != sh.lru.begin()) {
        sh.lru.splice(sh.lru.begin(), sh.lru, en->lru);
        sh.note_oldest();
    } else if (sh.lru.size() == 1) sh.note_oldest();
    return en;
}
Caps tenant_caps(bmq_route_cache* c, TenantCache* t, std::string_view tn) {
    if (t) return Caps{t->max_pf.load(std::memory_order_relaxed), t->max_gf.load(std::memory_order_relaxed)};
    const auto p = c->caps_of(tn);
    return Caps{p.first, p.second};
}
struct AsyncLoad { // a miss of bmq_route_cache_get_async on its way through the batching front
    bmq_route_cache* c;
    std::string tenant, topic;
    uint64_t th, now_ms;
    bool bypass, cold;
    Caps caps;
    bmq_route_cache_cb cb;
    void* user;
};
void async_loaded(void* user, int status, const uint32_t* ids, uint32_t n, uint64_t epoch) { // on the batcher's dispatcher thread
    std::unique_ptr<AsyncLoad> a((AsyncLoad*)user);
    bmq_route_cache* c = a->c;
    std::vector<uint32_t> row;
    if (status == BMQ_OK) {
        row.assign(ids, ids + n);
        uint32_t rp[2] = {0, n};
        const uint32_t toff[2] = {0, (uint32_t)a->tenant.size()}, poff[2] = {0, (uint32_t)a->topic.size()};
        RowClass cls;
        status = cap_rows_inplace(c, &a->caps, (const uint8_t*)a->tenant.data(), toff, nullptr, (const uint8_t*)a->topic.data(), poff, nullptr, 1, rp, row.data(), &cls);
        row.resize(status == BMQ_OK ? rp[1] : 0);
        if (status == BMQ_OK && !a->bypass && !c->bypass.load(std::memory_order_acquire)) {
            ReadGuard rg(c);
            TenantCache* t = c->obtain(a->tenant, a->now_ms);
            store_loaded(c, t, c->shard_of(*t, a->th), a->topic, a->th, std::vector<uint32_t>(row), epoch, a->now_ms, a->caps, cls, a->cold);
        }
    }
    a->cb(a->user, status, row.data(), (uint32_t)row.size(), epoch);
    c->async_inflight.fetch_sub(1, std::memory_order_acq_rel); // last touch of the cache: bmq_route_cache_destroy waits for this
}
} // namespace

int bmq_route_cache_get(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len, uint64_t now_ms,
                        uint32_t* out_route_ids, uint32_t cap, uint32_t* out_n, uint64_t* out_epoch) {
    if (!c || !out_n || (tenant_len && !tenant) || (topic_len && !topic)) return BMQ_E_INVAL;
    const std::string_view tn((const char*)tenant, tenant_len), tp((const char*)topic, topic_len);
    const uint64_t th = hash64(tp);
    const bool bypass = c->bypass.load(std::memory_order_acquire);
    ReadGuard rg(c);
    TenantCache* t = c->find(tn, hash64(tn)); // a tenant gets its cache with its first loaded row, not with its first question
    const Caps caps = tenant_caps(c, t, tn);
    const bool cold = !bypass && !t;
    if (!bypass && t) {
        c->touch(*t, now_ms);
        Shard& sh = c->shard_of(*t, th);
        SpinGuard g(sh.mu);
        if (Entry* en = lookup_live(c, t, sh, tp, th, now_ms, caps)) {
            *out_n = (uint32_t)en->ids.size();
            if (out_epoch) *out_epoch = en->epoch;
            if (en->ids.size() > cap) return BMQ_E_NOSPACE;
            if (!en->ids.empty()) memcpy(out_route_ids, en->ids.data(), en->ids.size() * 4);
            return BMQ_OK;
        }
        sh.misses++;
    } else if (!bypass) c->cold_misses.fetch_add(1, std::memory_order_relaxed);
    rg.leave(); // the load waits for a GPU launch: the tenant is looked up again afterwards
    // load: matchAll(singleton(topic), maxPersistentFanout, maxGroupFanout) through the batching front (TenantRouteCache.java:172-193)
    const uint32_t off[2] = {0, topic_len}, toff[2] = {0, tenant_len};
    uint32_t row[2] = {0, 0};
    std::vector<uint32_t> ids(64);
    uint64_t needed = 0, epoch = 0;
    int rc = bmq_batcher_match_all(c->b, tenant, tenant_len, topic, off, 1, row, ids.data(), ids.size(), &needed, &epoch);
    if (rc == BMQ_E_NOSPACE) {
        ids.resize(needed);
        rc = bmq_batcher_match_all(c->b, tenant, tenant_len, topic, off, 1, row, ids.data(), ids.size(), &needed, &epoch);
    }
    if (rc != BMQ_OK) return rc;
    ids.resize(needed);
    row[1] = (uint32_t)needed;
    RowClass cls;
    if ((rc = cap_rows_inplace(c, &caps, tenant, toff, nullptr, topic, off, nullptr, 1, row, ids.data(), &cls)) != BMQ_OK) return rc;
    ids.resize(row[1]);
    *out_n = (uint32_t)ids.size();
    if (out_epoch) *out_epoch = epoch;
    const bool fits = ids.size() <= cap;
    if (fits && !ids.empty()) memcpy(out_route_ids, ids.data(), ids.size() * 4);
    if (!bypass && !c->bypass.load(std::memory_order_acquire)) {
        rg.enter();
        t = c->obtain(tn, now_ms);
        store_loaded(c, t, c->shard_of(*t, th), tp, th, std::move(ids), epoch, now_ms, caps, cls, cold);
    }
    return fits ? BMQ_OK : BMQ_E_NOSPACE;
}

int bmq_route_cache_get_async(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len, uint64_t now_ms,
                              bmq_route_cache_cb cb, void* user) {
    if (!c || !cb || (tenant_len && !tenant) || (topic_len && !topic)) return BMQ_E_INVAL;
    const std::string_view tn((const char*)tenant, tenant_len), tp((const char*)topic, topic_len);
    const uint64_t th = hash64(tp);
    const bool bypass = c->bypass.load(std::memory_order_acquire);
    Caps caps;
    bool cold;
    {
        ReadGuard rg(c);
        TenantCache* t = c->find(tn, hash64(tn));
        caps = tenant_caps(c, t, tn);
        cold = !bypass && !t;
        if (!bypass && t) {
            c->touch(*t, now_ms);
            Shard& sh = c->shard_of(*t, th);
            std::vector<uint32_t> ids; // copied out: the callback runs without the lock
            uint64_t epoch = 0;
            bool hit = false;
            {
                SpinGuard g(sh.mu);
                if (Entry* en = lookup_live(c, t, sh, tp, th, now_ms, caps)) {
                    ids = en->ids;
                    epoch = en->epoch;
                    hit = true;
                } else sh.misses++;
            }
            if (hit) { // a completed future: the callback runs on the caller's thread, before this call returns
                rg.leave();
                cb(user, BMQ_OK, ids.data(), (uint32_t)ids.size(), epoch);
                return BMQ_OK;
            }
        } else if (!bypass) c->cold_misses.fetch_add(1, std::memory_order_relaxed);
    }
    auto a = std::make_unique<AsyncLoad>(AsyncLoad{c, std::string(tn), std::string(tp), th, now_ms, bypass, cold, caps, cb, user});
    c->async_inflight.fetch_add(1, std::memory_order_acq_rel);
    const int rc = bmq_batcher_submit(c->b, tenant, tenant_len, topic, topic_len, async_loaded, a.get());
    if (rc == BMQ_OK) a.release(); // async_loaded owns it now
    else c->async_inflight.fetch_sub(1, std::memory_order_acq_rel);
    return rc;
}

int bmq_route_cache_get_batch(bmq_route_cache* c, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                              const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, uint64_t now_ms, uint32_t* out_row_ptr,
                              uint32_t* out_route_ids, uint64_t out_capacity, uint64_t* out_needed, uint8_t* out_hit) {
    if (!c || !out_row_ptr || !out_needed || (n_topics && (!topics || !topic_off || !topic_tenant)) || (n_tenants && (!tenants || !tenant_off)))
        return BMQ_E_INVAL;
    *out_needed = 0;
    out_row_ptr[0] = 0;
    if (n_topics == 0) return BMQ_OK;
    for (uint32_t i = 0; i < n_topics; i++)
        if (topic_tenant[i] >= n_tenants) return BMQ_E_INVAL;
    ReadGuard rg(c);
    std::vector<TenantCache*> tcache(n_tenants);
    std::vector<uint64_t> thash(n_tenants);
    std::vector<Caps> tcaps(n_tenants);
    for (uint32_t t = 0; t < n_tenants; t++) {
        const std::string_view tn((const char*)tenants + tenant_off[t], tenant_off[t + 1] - tenant_off[t]);
        thash[t] = hash64(tn);
        tcache[t] = c->find(tn, thash[t]);
        tcaps[t] = tenant_caps(c, tcache[t], tn);
    }
    if (n_topics >= c->direct_batch) {
        // A big request is cheaper to match than to look up: one cache probe costs ~0.1-1 us of one host thread, the engine resolves
        // 10 k topics in 0.09 ms and a million in 0.4 ms.  Straight to one launch; the cache is not touched.
        // The rows are capped like loaded ones (MatchedRoutes), in place.
        rg.leave();
        uint64_t epoch = 0;
        if (out_hit) memset(out_hit, 0, n_topics);
        const int rc = bmq_batcher_match_batch(c->b, tenants, tenant_off, n_tenants, topic_tenant, topics, topic_off, n_topics, out_row_ptr, out_route_ids,
                                               out_capacity, out_needed, &epoch);
        if (rc != BMQ_OK) return rc;
        const int rcap = cap_rows_inplace(c, tcaps.data(), tenants, tenant_off, topic_tenant, topics, topic_off, nullptr, n_topics, out_row_ptr, out_route_ids, nullptr);
        *out_needed = out_row_ptr[n_topics];
        return rcap;
    }
    const bool bypass = c->bypass.load(std::memory_order_acquire);
    // pass 1: the cache.  Hits are copied out at once (an entry may be gone a moment later); misses are listed, identical ones once.
    struct Row {
        uint64_t off = 0; // into hit_ids, or (miss) the row of the launch
        uint32_t n = 0;
        bool hit = false;
    };
    std::vector<Row> rows(n_topics);
    std::vector<uint32_t> hit_ids;
    std::vector<uint32_t> miss_first;                  // launch row -> first topic index asking for it
    std::unordered_map<uint64_t, uint32_t> miss_row;   // hash of (tenant, topic) -> launch row (bytes compared)
    std::vector<uint64_t> phash(n_topics);
    std::vector<uint8_t> miss_cold; // launch row: its tenant had no cache when the row was probed
    for (uint32_t i = 0; i < n_topics; i++) {
        const uint32_t ti = topic_tenant[i];
        const std::string_view tp((const char*)topics + topic_off[i], topic_off[i + 1] - topic_off[i]);
        const uint64_t th = phash[i] = hash64(tp);
        if (!bypass && tcache[ti]) {
            c->touch(*tcache[ti], now_ms);
            Shard& sh = c->shard_of(*tcache[ti], th);
            SpinGuard g(sh.mu);
            if (Entry* en = lookup_live(c, tcache[ti], sh, tp, th, now_ms, tcaps[ti])) {
                rows[i].hit = true;
                rows[i].off = hit_ids.size();
                rows[i].n = (uint32_t)en->ids.size();
                hit_ids.insert(hit_ids.end(), en->ids.begin(), en->ids.end());
                continue;
            }
            sh.misses++;
        } else if (!bypass) c->cold_misses.fetch_add(1, std::memory_order_relaxed);
        uint64_t key = th ^ (thash[ti] * 0x9E3779B97F4A7C15ull);
        for (;; key++) { // open chaining on the 64-bit key: a different pair with the same key takes the next one
            auto it = miss_row.find(key);
            if (it == miss_row.end()) {
                miss_row.emplace(key, (uint32_t)miss_first.size());
                rows[i].off = miss_first.size();
                miss_first.push_back(i);
                miss_cold.push_back(!bypass && !tcache[ti]);
                break;
            }
            const uint32_t j = miss_first[it->second];
            if (topic_tenant[j] == ti && topic_off[j + 1] - topic_off[j] == tp.size() &&
                (tp.empty() || !memcmp(topics + topic_off[j], tp.data(), tp.size()))) {
                rows[i].off = it->second;
                break;
            }
        }
    }
    rg.leave(); // the launch below waits for the GPU: tenants are looked up again afterwards
    // pass 2: ONE launch for everything missing
    std::vector<uint32_t> m_row, m_ids;
    uint64_t epoch = 0;
    const uint32_t n_miss = (uint32_t)miss_first.size();
    if (n_miss) {
        std::vector<uint8_t> m_topics;
        std::vector<uint32_t> m_off{0}, m_tenant(n_miss);
        for (uint32_t r = 0; r < n_miss; r++) {
            const uint32_t i = miss_first[r];
            m_topics.insert(m_topics.end(), topics + topic_off[i], topics + topic_off[i + 1]);
            m_off.push_back((uint32_t)m_topics.size());
            m_tenant[r] = topic_tenant[i];
        }
        m_topics.resize(m_topics.size() + 16, 0); // the engine reads whole 16-byte groups
        m_row.resize((size_t)n_miss + 1);
        m_ids.resize(std::max<size_t>(1024, (size_t)n_miss * 16));
        uint64_t need = 0;
        int rc = BMQ_OK;
        for (int attempt = 0; attempt < 3; attempt++) {
            rc = bmq_batcher_match_batch(c->b, tenants, tenant_off, n_tenants, m_tenant.data(), m_topics.data(), m_off.data(), n_miss, m_row.data(),
                                         m_ids.data(), m_ids.size(), &need, &epoch);
            if (rc != BMQ_E_NOSPACE) break;
            m_ids.resize(need + need / 8 + 64);
        }
        if (rc != BMQ_OK) return rc;
        std::vector<RowClass> m_cls(n_miss);
        if ((rc = cap_rows_inplace(c, tcaps.data(), tenants, tenant_off, m_tenant.data(), topics, topic_off, miss_first.data(), n_miss, m_row.data(), m_ids.data(),
                                   m_cls.data())) != BMQ_OK)
            return rc;
        if (!bypass && !c->bypass.load(std::memory_order_acquire)) {
            rg.enter();
            for (uint32_t r = 0; r < n_miss; r++) {
                const uint32_t i = miss_first[r], ti = topic_tenant[i];
                TenantCache* t = c->obtain(std::string_view((const char*)tenants + tenant_off[ti], tenant_off[ti + 1] - tenant_off[ti]), now_ms);
                const std::string_view tp((const char*)topics + topic_off[i], topic_off[i + 1] - topic_off[i]);
                store_loaded(c, t, c->shard_of(*t, phash[i]), tp, phash[i], std::vector<uint32_t>(m_ids.begin() + m_row[r], m_ids.begin() + m_row[r + 1]), epoch, now_ms,
                             tcaps[ti], m_cls[r], miss_cold[r] != 0);
            }
            rg.leave();
        }
    }
    // pass 3: the CSR
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_topics; i++) {
        if (!rows[i].hit) rows[i].n = m_row[rows[i].off + 1] - m_row[rows[i].off];
        total += rows[i].n;
        if (total >= 0xFFFFFFFFull) return BMQ_E_RANGE;
        out_row_ptr[i + 1] = (uint32_t)total;
        if (out_hit) out_hit[i] = rows[i].hit ? 1 : 0;
    }
    *out_needed = total;
    if (total > out_capacity || (total && !out_route_ids)) return BMQ_E_NOSPACE;
    for (uint32_t i = 0; i < n_topics; i++)
        if (rows[i].n) memcpy(out_route_ids + out_row_ptr[i], rows[i].hit ? hit_ids.data() + rows[i].off : m_ids.data() + m_row[rows[i].off], (size_t)rows[i].n * 4);
    return BMQ_OK;
}

int bmq_route_cache_is_cached(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filter, uint32_t filter_len) {
    if (!c) return BMQ_E_INVAL;
    const std::string_view tn((const char*)tenant, tenant_len);
    ReadGuard rg(c);
    TenantCache* t = c->find(tn, hash64(tn));
    if (!t) return 0;
    const auto fl = split(std::string_view((const char*)filter, filter_len), '/');
    bool any = false;
    for (uint32_t s = 0; s < c->n_shards && !any; s++) {
        SpinGuard g(t->shards[s].mu);
        t->shards[s].index.match(fl, [&](Entry*) { any = true; });
    }
    return any ? 1 : 0;
}

int bmq_route_cache_apply(bmq_route_cache* c, const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n) {
    if (!c || (n && (!keys || !key_off))) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> ag(c->apply_mu);
    const int rc = bmq_routes_apply(c->e, keys, key_off, op, n);
    if (rc != BMQ_OK) return rc;
    bmq_index_info info{};
    const int ri = bmq_index_info_get(c->e, &info);
    if (ri != BMQ_OK) return ri;
    // per tenant that has a cache: log the filters, then drop the cached topics they match
    std::unordered_map<std::string, std::vector<std::vector<std::string>>> by_tenant;
    for (uint32_t i = 0; i < n; i++) {
        RouteKeyParts kp;
        if (!decode_route_key(std::string_view((const char*)keys + key_off[i], key_off[i + 1] - key_off[i]), kp)) continue; // the engine refused it already
        std::vector<std::string> fl;
        for (auto lv : split(kp.esc_filter, '\0')) fl.emplace_back(lv);
        by_tenant[std::string(kp.tenant)].push_back(std::move(fl));
    }
    c->created_floor.store(info.epoch, std::memory_order_seq_cst); // before the tenants are looked up: see obtain()
    ReadGuard rg(c);
    for (auto& bt : by_tenant) {
        TenantCache* t = c->find(bt.first, hash64(bt.first));
        if (!t) continue;
        {
            std::lock_guard<std::mutex> lg(t->log_mu);
            for (auto& fl : bt.second) t->log.push_back({info.epoch, fl});
            if (t->log.size() > c->log_keep) { // forget the oldest: loads older than what is left are not cached
                const size_t cut = t->log.size() - c->log_keep / 2;
                t->log_floor = std::max(t->log_floor, t->log[cut - 1].epoch);
                t->log.erase(t->log.begin(), t->log.begin() + (long)cut);
            }
        }
        for (uint32_t s = 0; s < c->n_shards; s++) {
            Shard& sh = t->shards[s];
            SpinGuard g(sh.mu);
            if (sh.lru.empty()) continue;
            std::vector<Entry*> hit;
            for (auto& fl : bt.second) {
                std::vector<std::string_view> fv(fl.begin(), fl.end());
                sh.index.match(fv, [&](Entry* en) { hit.push_back(en); });
            }
            std::sort(hit.begin(), hit.end()); // a topic may be hit by several filters of the batch
            hit.erase(std::unique(hit.begin(), hit.end()), hit.end());
            for (Entry* en : hit) {
                sh.invalidations++;
                t->weight.fetch_sub(sh.drop(en), std::memory_order_relaxed);
            }
        }
    }
    return BMQ_OK;
}

namespace {
// drop every entry and forget the log: nothing matched before `floor_epoch` is cached afterwards.  Tenant objects stay (they go with
// bmq_route_cache_expire).
void clear_all(bmq_route_cache* c, uint64_t floor_epoch) {
    c->created_floor.store(floor_epoch, std::memory_order_seq_cst);
    ReadGuard rg(c);
    c->for_each_tenant([&](TenantCache& t) {
        {
            std::lock_guard<std::mutex> lg(t.log_mu);
            t.log.clear();
            t.log_floor = std::max(t.log_floor, floor_epoch);
        }
        for (uint32_t s = 0; s < c->n_shards; s++) {
            SpinGuard g(t.shards[s].mu);
            t.weight.fetch_sub(t.shards[s].clear(), std::memory_order_relaxed);
        }
    });
}
void add_shard_stats(bmq_route_cache_stats& o, const Shard& sh) {
    o.hits += sh.hits;
    o.misses += sh.misses;
    o.evictions += sh.evictions;
    o.invalidations += sh.invalidations;
    o.stale_loads += sh.stale_loads;
    o.expired += sh.expired;
    o.entries += sh.entries;
    o.cached_routes += sh.weight;
}
} // namespace

int bmq_route_cache_reset(bmq_route_cache* c) {
    if (!c) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> ag(c->apply_mu);
    bmq_index_info info{};
    const int ri = bmq_index_info_get(c->e, &info);
    clear_all(c, ri == BMQ_OK ? info.epoch : ~0ull);
    return BMQ_OK;
}

int bmq_route_cache_rebuild(bmq_route_cache* c, const uint8_t* keys, const uint32_t* key_off, uint32_t n_keys) {
    if (!c) return BMQ_E_INVAL;
    std::lock_guard<std::mutex> ag(c->apply_mu);
    c->bypass.store(true, std::memory_order_release); // route ids of the old generation must not be served once the engine holds the new one
    const int rc = bmq_rebuild(c->e, keys, key_off, n_keys);
    bmq_index_info info{};
    const int ri = bmq_index_info_get(c->e, &info);
    clear_all(c, ri == BMQ_OK ? info.epoch : ~0ull);
    c->bypass.store(false, std::memory_order_release);
    return rc;
}

int bmq_route_cache_expire(bmq_route_cache* c, uint64_t now_ms, uint64_t* out_dropped) {
    if (!c) return BMQ_E_INVAL;
    uint64_t dropped = 0;
    bool idle_tenants = false;
    {
        ReadGuard rg(c);
        c->for_each_tenant([&](TenantCache& t) {
            for (uint32_t s = 0; s < c->n_shards; s++) {
                Shard& sh = t.shards[s];
                SpinGuard g(sh.mu);
                // the LRU order is the order of last access: the idle entries are at the back, the first live one ends the sweep
                while (!sh.lru.empty()) {
                    Entry* en = sh.lru.back();
                    if (!(now_ms >= en->last_access_ms && now_ms - en->last_access_ms >= c->expiry_ms)) break;
                    sh.expired++;
                    t.weight.fetch_sub(sh.drop(en), std::memory_order_relaxed);
                    dropped++;
                }
            }
            const uint64_t last = t.last_get_ms.load(std::memory_order_relaxed);
            if (now_ms >= last && now_ms - last >= c->tenant_idle_ms) idle_tenants = true;
        });
    }
    if (idle_tenants) {
        // SubscriptionCache.java:79-107: a tenant nobody asked about for 2 x expiry loses its cache (ITenantRouteCache.destroy).  Getters are
        // kept out while the idle tenants are unlinked (microseconds); a getter that was inside before `sweeping` went up is waited for.
        std::vector<TenantCache*> dead;
        {
            std::lock_guard<std::mutex> sg(c->sweep_mu);
            c->sweeping.store(true, std::memory_order_seq_cst);
            for (auto& st : c->stripes)
                while (st.n.load(std::memory_order_seq_cst) != 0) std::this_thread::yield();
            std::lock_guard<std::mutex> tg(c->table_mu); // nobody is inside any more: uncontended
            for (uint32_t i = 0; i < bmq_route_cache::TBUCKETS; i++) {
                std::atomic<TenantCache*>* link = &c->buckets[i];
                for (TenantCache* t = link->load(std::memory_order_relaxed); t; t = link->load(std::memory_order_relaxed)) {
                    const uint64_t last = t->last_get_ms.load(std::memory_order_relaxed);
                    if (now_ms >= last && now_ms - last >= c->tenant_idle_ms) {
                        link->store(t->next.load(std::memory_order_relaxed), std::memory_order_relaxed);
                        dead.push_back(t);
                    } else link = &t->next;
                }
            }
            c->sweeping.store(false, std::memory_order_seq_cst);
        }
        for (TenantCache* t : dead) {
            {
                std::lock_guard<std::mutex> rg(c->retired_mu);
                for (uint32_t s = 0; s < c->n_shards; s++) {
                    dropped += t->shards[s].entries;
                    bmq_route_cache_stats one{};
                    add_shard_stats(one, t->shards[s]);
                    c->retired.hits += one.hits, c->retired.misses += one.misses, c->retired.evictions += one.evictions;
                    c->retired.invalidations += one.invalidations, c->retired.stale_loads += one.stale_loads, c->retired.expired += one.expired + one.entries;
                }
            }
            delete t;
            c->tenants_live.fetch_sub(1, std::memory_order_relaxed);
            c->tenants_expired.fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (out_dropped) *out_dropped = dropped;
    return BMQ_OK;
}

int bmq_route_cache_stats_get(bmq_route_cache* c, bmq_route_cache_stats* out) {
    if (!c || !out) return BMQ_E_INVAL;
    {
        std::lock_guard<std::mutex> g(c->retired_mu);
        *out = c->retired;
    }
    out->misses += c->cold_misses.load(std::memory_order_relaxed);
    ReadGuard rg(c);
    c->for_each_tenant([&](TenantCache& t) {
        for (uint32_t s = 0; s < c->n_shards; s++) {
            Shard& sh = t.shards[s];
            SpinGuard g(sh.mu);
            add_shard_stats(*out, sh);
        }
    });
    out->tenants = c->tenants_live.load(std::memory_order_relaxed);
    out->tenants_expired = c->tenants_expired.load(std::memory_order_relaxed);
    return BMQ_OK;
}

int bmq_route_cache_tenant_stats_get(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, bmq_route_cache_tenant_stats* out) {
    if (!c || !out || (tenant_len && !tenant)) return BMQ_E_INVAL;
    memset(out, 0, sizeof(*out));
    const std::string_view tn((const char*)tenant, tenant_len);
    ReadGuard rg(c);
    TenantCache* t = c->find(tn, hash64(tn));
    if (!t) return BMQ_E_STATE; // no cache for this tenant (never loaded, or destroyed after 2 x expiry idle): its meters are gone with it
    for (uint32_t s = 0; s < c->n_shards; s++) {
        Shard& sh = t->shards[s];
        SpinGuard g(sh.mu);
        out->hits += sh.hits;
        out->misses += sh.misses;
        out->evictions += sh.evictions;
        out->entries += sh.entries;
        out->cached_routes += sh.weight;
    }
    out->last_get_ms = t->last_get_ms.load(std::memory_order_relaxed);
    out->max_persistent_fanout = t->max_pf.load(std::memory_order_relaxed);
    out->max_group_fanout = t->max_gf.load(std::memory_order_relaxed);
    return BMQ_OK;
}

} // extern "C"
