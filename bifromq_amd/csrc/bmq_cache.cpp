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
#include "../../include/bmq.h"
#include "bmq_codec.h"

#include <algorithm>
#include <atomic>
#include <chrono>
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
    Entry* next_same_hash = nullptr;
    std::list<Entry*>::iterator lru;
    uint64_t weight() const { return ids.empty() ? 1 : ids.size(); } // TenantRouteCache.java:108-111
};

// TopicIndex (DW/TopicIndex.java:39-156): trie of the cached topics; match(filter) walks it as TopicMatcher's selector does
struct TopicIndex {
    struct Node {
        std::unordered_map<std::string, std::unique_ptr<Node>> children;
        Entry* value = nullptr;
    };
    Node root;
    void add(const std::vector<std::string_view>& levels, Entry* e) {
        Node* n = &root;
        for (auto lv : levels) {
            auto& c = n->children[std::string(lv)];
            if (!c) c = std::make_unique<Node>();
            n = c.get();
        }
        n->value = e;
    }
    void remove(const std::vector<std::string_view>& levels) { remove_at(root, levels, 0); }
    template <class F> void match(const std::vector<std::string_view>& filter, F&& f) const { walk(root, filter, 0, f); }

private:
    static bool remove_at(Node& n, const std::vector<std::string_view>& levels, size_t i) { // true: n is empty now
        if (i == levels.size()) n.value = nullptr;
        else {
            auto it = n.children.find(std::string(levels[i]));
            if (it != n.children.end() && remove_at(*it->second, levels, i + 1)) n.children.erase(it);
        }
        return !n.value && n.children.empty();
    }
    template <class F> static void all_below(const Node& n, F& f) {
        if (n.value) f(n.value);
        for (auto& c : n.children) all_below(*c.second, f);
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
            for (auto& c : n.children)
                if (!(i == 0 && !c.first.empty() && c.first[0] == '$')) all_below(*c.second, f);
            return;
        }
        if (lv == "+") {
            for (auto& c : n.children)
                if (!(i == 0 && !c.first.empty() && c.first[0] == '$')) walk(*c.second, flt, i + 1, f);
            return;
        }
        auto it = n.children.find(std::string(lv));
        if (it != n.children.end()) walk(*it->second, flt, i + 1, f);
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
// the publishes of one hot tenant do not queue behind one lock.
struct alignas(64) Shard {
    Spin mu;
    std::unordered_map<uint64_t, Entry*> table; // topic hash -> chain (next_same_hash)
    std::list<Entry*> lru;                      // front = most recently used
    uint64_t weight = 0;
    TopicIndex index;
    uint64_t hits = 0, misses = 0, evictions = 0, invalidations = 0, stale_loads = 0, expired = 0, entries = 0;

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
    }
    void drop(Entry* en) {
        index.remove(split(en->topic, '/'));
        weight -= en->weight();
        lru.erase(en->lru);
        auto it = table.find(en->hash);
        Entry** pp = &it->second;
        while (*pp != en) pp = &(*pp)->next_same_hash;
        *pp = en->next_same_hash;
        if (!it->second) table.erase(it);
        entries--;
        delete en;
    }
    void clear() {
        while (!lru.empty()) drop(lru.back());
    }
    ~Shard() { clear(); }
};

struct TenantCache { // TenantRouteCache
    std::string name;
    uint64_t hash = 0;
    TenantCache* next = nullptr; // chain of the tenant table bucket (immutable once published)
    std::unique_ptr<Shard[]> shards;
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
    uint32_t log_keep = 4096;
    uint32_t n_shards = 16;                  // per tenant, power of two
    uint32_t direct_batch = 8192;            // get_batch: requests of at least this many topics are matched without the cache
    // tenant table: insert-only chained hash with atomic bucket heads -- a lookup takes no lock and writes nothing
    static constexpr uint32_t TBUCKETS = 1u << 14;
    std::unique_ptr<std::atomic<TenantCache*>[]> buckets;
    std::mutex apply_mu; // one refresh at a time, in commit order (ISubscriptionCache.refresh comes from the range's apply thread)
    std::atomic<uint64_t> created_floor{0};
    std::atomic<bool> bypass{false}; // a rebuild is replacing the index: serve nothing from the cache, cache nothing
    std::atomic<uint64_t> async_inflight{0}; // misses of get_async still with the batching front
    std::atomic<uint64_t> cold_misses{0};    // gets for a tenant that has no cache yet

    bmq_route_cache() : buckets(new std::atomic<TenantCache*>[TBUCKETS]) {
        for (uint32_t i = 0; i < TBUCKETS; i++) buckets[i].store(nullptr, std::memory_order_relaxed);
    }
    ~bmq_route_cache() {
        for (uint32_t i = 0; i < TBUCKETS; i++)
            for (TenantCache* t = buckets[i].load(); t;) {
                TenantCache* nx = t->next;
                delete t;
                t = nx;
            }
    }
    TenantCache* find(std::string_view tenant, uint64_t h) const {
        for (TenantCache* t = buckets[h & (TBUCKETS - 1)].load(std::memory_order_acquire); t; t = t->next)
            if (t->hash == h && t->name == tenant) return t;
        return nullptr;
    }
    TenantCache* obtain(std::string_view tenant) {
        const uint64_t h = hash64(tenant);
        if (TenantCache* t = find(tenant, h)) return t;
        auto nt = std::make_unique<TenantCache>();
        nt->name = std::string(tenant);
        nt->hash = h;
        nt->shards.reset(new Shard[n_shards]);
        nt->log_floor = created_floor.load(); // mutations before the tenant cache existed were never logged for it
        std::atomic<TenantCache*>& head = buckets[h & (TBUCKETS - 1)];
        TenantCache* old = head.load(std::memory_order_acquire);
        for (;;) {
            for (TenantCache* t = old; t; t = t->next)
                if (t->hash == h && t->name == tenant) return t; // somebody else published it meanwhile
            nt->next = old;
            if (head.compare_exchange_weak(old, nt.get(), std::memory_order_release, std::memory_order_acquire)) return nt.release();
        }
    }
    template <class F> void for_each_tenant(F&& f) {
        for (uint32_t i = 0; i < TBUCKETS; i++)
            for (TenantCache* t = buckets[i].load(std::memory_order_acquire); t; t = t->next) f(*t);
    }
    Shard& shard_of(TenantCache& t, uint64_t topic_hash) const { return t.shards[(topic_hash >> 40) & (n_shards - 1)]; }
    uint64_t shard_budget() const { return std::max<uint64_t>(1, max_routes_per_tenant / n_shards); }
};

namespace {
struct SpinGuard {
    Spin& s;
    explicit SpinGuard(Spin& sp) : s(sp) { s.lock(); }
    ~SpinGuard() { s.unlock(); }
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
    }
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

namespace {
// A loaded row goes into the cache unless a mutation that could change it has been applied since it was matched.
void store_loaded(bmq_route_cache* c, TenantCache* t, Shard& sh, std::string_view tp, uint64_t th, std::vector<uint32_t>&& ids, uint64_t epoch,
                  uint64_t now_ms) {
    auto en = std::make_unique<Entry>(); // built outside the lock
    en->topic = std::string(tp);
    en->ids = std::move(ids);
    en->hash = th;
    en->epoch = epoch;
    en->last_access_ms = now_ms;
    const auto tl = split(tp, '/');
    SpinGuard g(sh.mu);
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
    if (have && have->epoch >= epoch) return; // another thread loaded the same topic meanwhile
    if (have) sh.drop(have);
    sh.insert(en.release());
    const uint64_t budget = c->shard_budget();
    while (sh.weight > budget && sh.lru.size() > 1) { // maximumWeight: least recently used first
        sh.evictions++;
        sh.drop(sh.lru.back());
    }
}
// a live entry, touched -- or nullptr (an expired one is dropped on the way).  Shard lock held.
Entry* lookup_live(bmq_route_cache* c, Shard& sh, std::string_view tp, uint64_t th, uint64_t now_ms) {
    Entry* en = sh.find(th, tp);
    if (!en) return nullptr;
    if (now_ms >= en->last_access_ms && now_ms - en->last_access_ms >= c->expiry_ms) { // expireAfterAccess
        sh.expired++;
        sh.drop(en);
        return nullptr;
    }
    sh.hits++;
    en->last_access_ms = now_ms;
    if (en->lru != sh.lru.begin()) sh.lru.splice(sh.lru.begin(), sh.lru, en->lru);
    return en;
}
struct AsyncLoad { // a miss of bmq_route_cache_get_async on its way through the batching front
    bmq_route_cache* c;
    std::string tenant, topic;
    uint64_t th, now_ms;
    bool bypass;
    bmq_route_cache_cb cb;
    void* user;
};
void async_loaded(void* user, int status, const uint32_t* ids, uint32_t n, uint64_t epoch) { // on the batcher's dispatcher thread
    std::unique_ptr<AsyncLoad> a((AsyncLoad*)user);
    if (status == BMQ_OK && !a->bypass && !a->c->bypass.load(std::memory_order_acquire)) {
        TenantCache* t = a->c->obtain(a->tenant);
        store_loaded(a->c, t, a->c->shard_of(*t, a->th), a->topic, a->th, std::vector<uint32_t>(ids, ids + n), epoch, a->now_ms);
    }
    a->cb(a->user, status, ids, n, epoch);
    a->c->async_inflight.fetch_sub(1, std::memory_order_acq_rel); // last touch of the cache: bmq_route_cache_destroy waits for this
}
} // namespace

int bmq_route_cache_get(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len, uint64_t now_ms,
                        uint32_t* out_route_ids, uint32_t cap, uint32_t* out_n, uint64_t* out_epoch) {
    if (!c || !out_n || (tenant_len && !tenant) || (topic_len && !topic)) return BMQ_E_INVAL;
    const std::string_view tn((const char*)tenant, tenant_len), tp((const char*)topic, topic_len);
    TenantCache* t = c->find(tn, hash64(tn)); // a tenant gets its cache with its first loaded row, not with its first question
    const uint64_t th = hash64(tp);
    const bool bypass = c->bypass.load(std::memory_order_acquire);
    if (!bypass && t) {
        Shard& sh = c->shard_of(*t, th);
        SpinGuard g(sh.mu);
        if (Entry* en = lookup_live(c, sh, tp, th, now_ms)) {
            *out_n = (uint32_t)en->ids.size();
            if (out_epoch) *out_epoch = en->epoch;
            if (en->ids.size() > cap) return BMQ_E_NOSPACE;
            if (!en->ids.empty()) memcpy(out_route_ids, en->ids.data(), en->ids.size() * 4);
            return BMQ_OK;
        }
        sh.misses++;
    } else if (!bypass) c->cold_misses.fetch_add(1, std::memory_order_relaxed);
    // load: matchAll(singleton(topic)) through the batching front (TenantRouteCache.java:180-193)
    const uint32_t off[2] = {0, topic_len};
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
    *out_n = (uint32_t)ids.size();
    if (out_epoch) *out_epoch = epoch;
    const bool fits = ids.size() <= cap;
    if (fits && !ids.empty()) memcpy(out_route_ids, ids.data(), ids.size() * 4);
    if (!bypass && !c->bypass.load(std::memory_order_acquire)) {
        if (!t) t = c->obtain(tn);
        store_loaded(c, t, c->shard_of(*t, th), tp, th, std::move(ids), epoch, now_ms);
    }
    return fits ? BMQ_OK : BMQ_E_NOSPACE;
}

int bmq_route_cache_get_async(bmq_route_cache* c, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len, uint64_t now_ms,
                              bmq_route_cache_cb cb, void* user) {
    if (!c || !cb || (tenant_len && !tenant) || (topic_len && !topic)) return BMQ_E_INVAL;
    const std::string_view tn((const char*)tenant, tenant_len), tp((const char*)topic, topic_len);
    TenantCache* t = c->find(tn, hash64(tn));
    const uint64_t th = hash64(tp);
    const bool bypass = c->bypass.load(std::memory_order_acquire);
    if (!bypass && t) {
        Shard& sh = c->shard_of(*t, th);
        std::vector<uint32_t> ids; // copied out: the callback runs without the lock
        uint64_t epoch = 0;
        bool hit = false;
        {
            SpinGuard g(sh.mu);
            if (Entry* en = lookup_live(c, sh, tp, th, now_ms)) {
                ids = en->ids;
                epoch = en->epoch;
                hit = true;
            } else sh.misses++;
        }
        if (hit) { // a completed future: the callback runs on the caller's thread, before this call returns
            cb(user, BMQ_OK, ids.data(), (uint32_t)ids.size(), epoch);
            return BMQ_OK;
        }
    } else if (!bypass) c->cold_misses.fetch_add(1, std::memory_order_relaxed);
    auto a = std::make_unique<AsyncLoad>(AsyncLoad{c, std::string(tn), std::string(tp), th, now_ms, bypass, cb, user});
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
    if (n_topics >= c->direct_batch) {
        // A big request is cheaper to match than to look up: one cache probe costs ~0.1-1 us of one host thread, the engine resolves
        // 10 k topics in 0.09 ms and a million in 0.4 ms.  Straight to one launch; the cache is not touched.
        uint64_t epoch = 0;
        if (out_hit) memset(out_hit, 0, n_topics);
        return bmq_batcher_match_batch(c->b, tenants, tenant_off, n_tenants, topic_tenant, topics, topic_off, n_topics, out_row_ptr, out_route_ids,
                                       out_capacity, out_needed, &epoch);
    }
    const bool bypass = c->bypass.load(std::memory_order_acquire);
    // pass 1: the cache.  Hits are copied out at once (an entry may be gone a moment later); misses are listed, identical ones once.
    std::vector<TenantCache*> tcache(n_tenants);
    std::vector<uint64_t> thash(n_tenants);
    for (uint32_t t = 0; t < n_tenants; t++) {
        const std::string_view tn((const char*)tenants + tenant_off[t], tenant_off[t + 1] - tenant_off[t]);
        thash[t] = hash64(tn);
        tcache[t] = c->find(tn, thash[t]);
    }
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
    for (uint32_t i = 0; i < n_topics; i++) {
        const uint32_t ti = topic_tenant[i];
        const std::string_view tp((const char*)topics + topic_off[i], topic_off[i + 1] - topic_off[i]);
        const uint64_t th = phash[i] = hash64(tp);
        if (!bypass && tcache[ti]) {
            Shard& sh = c->shard_of(*tcache[ti], th);
            SpinGuard g(sh.mu);
            if (Entry* en = lookup_live(c, sh, tp, th, now_ms)) {
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
        if (!bypass && !c->bypass.load(std::memory_order_acquire))
            for (uint32_t r = 0; r < n_miss; r++) {
                const uint32_t i = miss_first[r], ti = topic_tenant[i];
                if (!tcache[ti]) tcache[ti] = c->obtain(std::string_view((const char*)tenants + tenant_off[ti], tenant_off[ti + 1] - tenant_off[ti]));
                const std::string_view tp((const char*)topics + topic_off[i], topic_off[i + 1] - topic_off[i]);
                store_loaded(c, tcache[ti], c->shard_of(*tcache[ti], phash[i]), tp, phash[i], std::vector<uint32_t>(m_ids.begin() + m_row[r], m_ids.begin() + m_row[r + 1]),
                             epoch, now_ms);
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
    c->created_floor = info.epoch;
    for (auto& bt : by_tenant) {
        TenantCache* t = c->find(bt.first, hash64(bt.first));
        if (!t) continue;
        {
            std::lock_guard<std::mutex> lg(t->log_mu);
            for (auto& fl : bt.second) t->log.push_back({info.epoch, fl});
            if (t->log.size() > c->log_keep) { // forget the oldest: loads older than what is left are not cached
                const size_t cut = t->log.size() - c->log_keep / 2;
                t->log_floor = t->log[cut - 1].epoch;
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
                sh.drop(en);
            }
        }
    }
    return BMQ_OK;
}

namespace {
// drop every entry and forget the log: nothing matched before `floor_epoch` is cached afterwards.  Tenant objects stay (other threads
// may hold pointers to them).
void clear_all(bmq_route_cache* c, uint64_t floor_epoch) {
    c->created_floor = floor_epoch;
    c->for_each_tenant([&](TenantCache& t) {
        {
            std::lock_guard<std::mutex> lg(t.log_mu);
            t.log.clear();
            t.log_floor = floor_epoch;
        }
        for (uint32_t s = 0; s < c->n_shards; s++) {
            SpinGuard g(t.shards[s].mu);
            t.shards[s].clear();
        }
    });
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
    c->for_each_tenant([&](TenantCache& t) {
        for (uint32_t s = 0; s < c->n_shards; s++) {
            Shard& sh = t.shards[s];
            SpinGuard g(sh.mu);
            // the LRU order is the order of last access: the idle entries are at the back, the first live one ends the sweep
            while (!sh.lru.empty()) {
                Entry* en = sh.lru.back();
                if (!(now_ms >= en->last_access_ms && now_ms - en->last_access_ms >= c->expiry_ms)) break;
                sh.expired++;
                sh.drop(en);
                dropped++;
            }
        }
    });
    if (out_dropped) *out_dropped = dropped;
    return BMQ_OK;
}

int bmq_route_cache_stats_get(bmq_route_cache* c, bmq_route_cache_stats* out) {
    if (!c || !out) return BMQ_E_INVAL;
    memset(out, 0, sizeof(*out));
    out->misses = c->cold_misses.load(std::memory_order_relaxed);
    c->for_each_tenant([&](TenantCache& t) {
        for (uint32_t s = 0; s < c->n_shards; s++) {
            Shard& sh = t.shards[s];
            SpinGuard g(sh.mu);
            out->hits += sh.hits;
            out->misses += sh.misses;
            out->evictions += sh.evictions;
            out->invalidations += sh.invalidations;
            out->stale_loads += sh.stale_loads;
            out->expired += sh.expired;
            out->entries += sh.entries;
            out->cached_routes += sh.weight;
        }
    });
    return BMQ_OK;
}

} // extern "C"
