// cache_fuzz.cpp -- test tool, not product: the route cache (bifromq_amd/csrc/bmq_cache.cpp) under AddressSanitizer/UBSan and
// ThreadSanitizer, without a GPU.  The engine and the batching front behind the cache are replaced by a stand-in that matches by brute
// force over a std::map model and counts epochs exactly as the engine does, so that every answer can be checked:
//   1. TopicIndex against the reference's golden table (DWT/TopicIndexTest.java:41-73) and against the matching rule on random input;
//   2. the reference's own cache scenarios (DWT/cache/TenantRouteCacheTest.java: load + index on first access, reuse without reload, Add /
//      RemoveRoutes tasks, a task overtaking a load, weight bound with empty rows, index clean-up on expiry) and more single-threaded behaviour: miss -> hit, isCached, invalidation by a matching mutation only, weight-bounded LRU eviction,
//      expire-after-access, rebuild;
//   3. getter threads against a mutator thread: every answer equals the brute force at the epoch it reports, and once everything has
//      settled no cached entry differs from the brute force on the final model (a load overtaken by a mutation must not be cached).
// Build + run: make -C bifromq_amd/csrc cachefuzz   (tests/test_host.py runs both builds)
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../bifromq_amd/csrc/bmq_cache.cpp"
#include "../bifromq_amd/csrc/bmq_codec.h"

// ---- stand-in engine ---------------------------------------------------------------------------------------------------------------
struct bmq_engine {
    std::mutex mu;
    std::map<std::string, uint32_t> model; // route key -> id
    uint32_t next_id = 0;
    uint64_t epoch = 1, generation = 1;
    std::vector<std::map<std::string, uint32_t>> history{{}, {}}; // history[epoch] = model at that epoch
    std::atomic<int> match_delay_us{0};
    std::atomic<uint64_t> n_match{0}, n_launch{0};
    std::atomic<bool> hold_loads{false}; // a load that has read the model waits here until released (TenantRouteCacheTest.java:257-292)
    std::atomic<int> loads_waiting{0};
    std::atomic<uint64_t> n_cap_calls{0};
};
struct bmq_batcher { // blocking side: matches inline; asynchronous side: a dispatcher thread, as the real front has
    bmq_engine* e;
    struct Req {
        std::string tenant, topic;
        bmq_batcher_cb cb;
        void* user;
    };
    std::mutex qm;
    std::condition_variable qcv;
    std::vector<Req> queue;
    std::thread dispatcher;
    bool stop = false;
    explicit bmq_batcher(bmq_engine* eng) : e(eng) {}
    ~bmq_batcher() {
        {
            std::lock_guard<std::mutex> g(qm);
            stop = true;
        }
        qcv.notify_all();
        if (dispatcher.joinable()) dispatcher.join();
    }
};

static std::vector<uint32_t> brute(const std::map<std::string, uint32_t>& model, std::string_view tenant, std::string_view topic) {
    std::vector<uint32_t> out;
    const auto tl = bmq::cache::split(topic, '/');
    for (auto& kv : model) {
        bmq::RouteKeyParts kp;
        if (!bmq::decode_route_key(kv.first, kp) || kp.tenant != tenant) continue;
        if (bmq::cache::filter_matches(bmq::cache::split(kp.esc_filter, '\0'), tl)) out.push_back(kv.second);
    }
    std::sort(out.begin(), out.end());
    return out;
}

extern "C" {
int bmq_batcher_match_all(bmq_batcher* b, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                          uint32_t* out_row_ptr, uint32_t* out_route_ids, uint64_t out_capacity, uint64_t* out_needed, uint64_t* out_epoch) {
    if (n_topics != 1) return BMQ_E_INVAL;
    std::vector<uint32_t> ids;
    {
        std::lock_guard<std::mutex> g(b->e->mu);
        ids = brute(b->e->model, std::string_view((const char*)tenant, tenant_len),
                    std::string_view((const char*)topics + topic_off[0], topic_off[1] - topic_off[0]));
        *out_epoch = b->e->epoch;
    }
    b->e->n_match++;
    if (const int d = b->e->match_delay_us.load()) std::this_thread::sleep_for(std::chrono::microseconds(d)); // the answer travels a while
    if (b->e->hold_loads.load()) {
        b->e->loads_waiting++;
        while (b->e->hold_loads.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
        b->e->loads_waiting--;
    }
    out_row_ptr[0] = 0;
    out_row_ptr[1] = (uint32_t)ids.size();
    *out_needed = ids.size();
    if (ids.size() > out_capacity) return BMQ_E_NOSPACE;
    for (size_t i = 0; i < ids.size(); i++) out_route_ids[i] = ids[i];
    return BMQ_OK;
}
int bmq_batcher_match_batch(bmq_batcher* b, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                            const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, uint32_t* out_row_ptr, uint32_t* out_route_ids,
                            uint64_t out_capacity, uint64_t* out_needed, uint64_t* out_epoch) {
    std::vector<std::vector<uint32_t>> rows(n_topics);
    {
        std::lock_guard<std::mutex> g(b->e->mu); // one launch: one epoch
        for (uint32_t i = 0; i < n_topics; i++) {
            if (topic_tenant[i] >= n_tenants) return BMQ_E_INVAL;
            rows[i] = brute(b->e->model, std::string_view((const char*)tenants + tenant_off[topic_tenant[i]], tenant_off[topic_tenant[i] + 1] - tenant_off[topic_tenant[i]]),
                            std::string_view((const char*)topics + topic_off[i], topic_off[i + 1] - topic_off[i]));
        }
        *out_epoch = b->e->epoch;
    }
    b->e->n_match += n_topics;
    b->e->n_launch++;
    if (const int d = b->e->match_delay_us.load()) std::this_thread::sleep_for(std::chrono::microseconds(d));
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_topics; i++) {
        out_row_ptr[i] = (uint32_t)total;
        total += rows[i].size();
    }
    out_row_ptr[n_topics] = (uint32_t)total;
    *out_needed = total;
    if (total > out_capacity) return BMQ_E_NOSPACE;
    for (uint32_t i = 0; i < n_topics; i++)
        for (size_t k = 0; k < rows[i].size(); k++) out_route_ids[out_row_ptr[i] + k] = rows[i][k];
    return BMQ_OK;
}
int bmq_batcher_submit(bmq_batcher* b, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topic, uint32_t topic_len, bmq_batcher_cb cb, void* user) {
    std::lock_guard<std::mutex> g(b->qm);
    if (b->stop) return BMQ_E_STATE;
    if (!b->dispatcher.joinable())
        b->dispatcher = std::thread([b] {
            for (;;) {
                std::vector<bmq_batcher::Req> batch;
                {
                    std::unique_lock<std::mutex> lk(b->qm);
                    b->qcv.wait(lk, [&] { return b->stop || !b->queue.empty(); });
                    if (b->queue.empty()) return;
                    batch.swap(b->queue);
                }
                std::vector<std::vector<uint32_t>> rows;
                uint64_t epoch;
                {
                    std::lock_guard<std::mutex> eg(b->e->mu); // one "launch": every row of the batch sees the same epoch
                    for (auto& r : batch) rows.push_back(brute(b->e->model, r.tenant, r.topic));
                    epoch = b->e->epoch;
                }
                b->e->n_match += batch.size();
                if (const int d = b->e->match_delay_us.load()) std::this_thread::sleep_for(std::chrono::microseconds(d));
                for (size_t i = 0; i < batch.size(); i++) batch[i].cb(batch[i].user, BMQ_OK, rows[i].data(), (uint32_t)rows[i].size(), epoch);
            }
        });
    b->queue.push_back({std::string((const char*)tenant, tenant_len), std::string((const char*)topic, topic_len), cb, user});
    b->qcv.notify_one();
    return BMQ_OK;
}
int bmq_routes_apply(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n) {
    std::lock_guard<std::mutex> g(e->mu);
    for (uint32_t i = 0; i < n; i++) {
        const std::string k((const char*)keys + key_off[i], key_off[i + 1] - key_off[i]);
        if (op && op[i]) e->model.erase(k);
        else if (!e->model.count(k)) e->model[k] = e->next_id++;
    }
    e->epoch++;
    e->history.push_back(e->model);
    return BMQ_OK;
}
int bmq_rebuild(bmq_engine* e, const uint8_t* keys, const uint32_t* key_off, uint32_t n) {
    std::lock_guard<std::mutex> g(e->mu);
    std::set<std::string> ks;
    for (uint32_t i = 0; i < n; i++) ks.emplace((const char*)keys + key_off[i], key_off[i + 1] - key_off[i]);
    e->model.clear();
    e->next_id = 0;
    for (auto& k : ks) e->model[k] = e->next_id++;
    e->epoch++;
    e->generation++;
    e->history.push_back(e->model);
    return BMQ_OK;
}
// MatchedRoutes over rows of ids, as the engine's bmq_routes_cap does it: keys of the ids from the model, KV key order, caps first-come
int bmq_routes_cap(bmq_engine* e, const uint32_t* row_ptr, const uint32_t* route_ids, uint32_t n_rows, int32_t max_pf, int32_t max_gf, uint32_t* out_row_ptr,
                   uint32_t* out_route_ids, uint32_t* out_class_counts, int32_t* out_events, uint32_t events_cap, uint32_t* out_n_events) {
    std::map<uint32_t, std::string> by_id;
    {
        std::lock_guard<std::mutex> g(e->mu);
        for (auto& kv : e->model) by_id[kv.second] = kv.first;
    }
    e->n_cap_calls++;
    uint32_t total = 0, n_ev = 0;
    for (uint32_t r = 0; r < n_rows; r++) {
        out_row_ptr[r] = total;
        std::vector<std::pair<std::string, uint32_t>> row;
        for (uint32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
            auto it = by_id.find(route_ids[k]);
            if (it != by_id.end()) row.emplace_back(it->second, route_ids[k]);
        }
        std::sort(row.begin(), row.end());
        int64_t pers = 0, grp = 0;
        std::vector<uint32_t> kept;
        for (auto& en : row) {
            bmq::RouteKeyParts kp;
            bmq::decode_route_key(en.first, kp);
            int ev = -1;
            if (kp.flag == 1) {
                if (kp.receiver.substr(0, 2) == std::string_view("1\0", 2)) {
                    if (pers < max_pf) pers++;
                    else ev = 0;
                }
            } else if (grp + 1 <= max_gf) grp++;
            else ev = 1;
            if (ev < 0) kept.push_back(en.second);
            else {
                if (out_events && n_ev < events_cap) {
                    int32_t* o = out_events + 4 * (size_t)n_ev;
                    o[0] = ev, o[1] = (int32_t)r, o[2] = (int32_t)en.second, o[3] = ev == 0 ? max_pf : max_gf;
                }
                n_ev++;
            }
        }
        std::sort(kept.begin(), kept.end());
        for (uint32_t id : kept) out_route_ids[total++] = id;
        if (out_class_counts) out_class_counts[2 * r] = (uint32_t)pers, out_class_counts[2 * r + 1] = (uint32_t)grp;
    }
    out_row_ptr[n_rows] = total;
    if (out_n_events) *out_n_events = n_ev;
    return BMQ_OK;
}
int bmq_index_info_get(const bmq_engine* ce, bmq_index_info* out) {
    bmq_engine* e = const_cast<bmq_engine*>(ce);
    std::lock_guard<std::mutex> g(e->mu);
    memset(out, 0, sizeof(*out));
    out->epoch = e->epoch;
    out->generation = e->generation;
    out->n_routes = e->model.size();
    return BMQ_OK;
}
}

static int g_fail = 0;
#define EXPECT(c)                                                       \
    do {                                                                \
        if (!(c)) {                                                     \
            fprintf(stderr, "cache_fuzz: %s (line %d)\n", #c, __LINE__); \
            g_fail++;                                                   \
        }                                                               \
    } while (0)

// ---- 1. TopicIndex ------------------------------------------------------------------------------------------------------------------
static std::set<std::string> index_match(const bmq::cache::TopicIndex& ix, const std::string& filter) {
    std::set<std::string> out;
    ix.match(bmq::cache::split(filter, '/'), [&](bmq::cache::Entry* e) { out.insert(e->topic); });
    return out;
}
static void test_topic_index(uint64_t seed) {
    using namespace bmq::cache;
    // DWT/TopicIndexTest.java:41-73
    const std::vector<std::string> topics = {"/", "/a", "/b", "a", "a/", "a/b", "a/b/c", "$a", "$a/", "$a/b"};
    std::vector<std::unique_ptr<Entry>> es;
    TopicIndex ix;
    for (auto& t : topics) {
        es.push_back(std::make_unique<Entry>());
        es.back()->topic = t;
        ix.add(split(es.back()->topic, '/'), es.back().get());
    }
    using S = std::set<std::string>;
    const std::vector<std::pair<std::string, S>> table = {
        {"/", {"/"}}, {"/a", {"/a"}}, {"/b", {"/b"}}, {"a", {"a"}}, {"a/", {"a/"}}, {"a/b", {"a/b"}}, {"a/b/c", {"a/b/c"}}, {"$a", {"$a"}},
        {"$a/", {"$a/"}}, {"$a/b", {"$a/b"}}, {"", {}}, {"fakeTopic", {}},
        {"#", {"/", "/a", "/b", "a", "a/", "a/b", "a/b/c"}}, {"+", {"a"}}, {"+/#", {"/", "/a", "/b", "a", "a/", "a/b", "a/b/c"}},
        {"+/+", {"/", "/a", "/b", "a/", "a/b"}}, {"+/+/#", {"/", "/a", "/b", "a/", "a/b", "a/b/c"}},
        {"/+", {"/", "/a", "/b"}}, {"/+/#", {"/", "/a", "/b"}}, {"/#", {"/", "/a", "/b"}},
        {"a/+", {"a/", "a/b"}}, {"a/#", {"a", "a/", "a/b", "a/b/c"}},
        {"$a/+", {"$a/", "$a/b"}}, {"$a/+/#", {"$a/", "$a/b"}}, {"$a/#", {"$a", "$a/", "$a/b"}}};
    for (auto& row : table) EXPECT(index_match(ix, row.first) == row.second);
    // remove: DWT/TopicIndexTest.java (remove then match)
    ix.remove(split("a/b", '/'));
    EXPECT(index_match(ix, "a/#") == (S{"a", "a/", "a/b/c"}));
    ix.remove(split("a/b/c", '/'));
    EXPECT(index_match(ix, "a/#") == (S{"a", "a/"}));
    // random: the trie walk == the rule applied to every topic
    std::mt19937_64 rng(seed);
    const std::vector<std::string> alpha = {"a", "b", "", "$s", "c", "dd"};
    auto level = [&]() { return alpha[rng() % alpha.size()]; };
    for (int round = 0; round < 50; round++) {
        TopicIndex rx;
        std::vector<std::unique_ptr<Entry>> re;
        std::set<std::string> ts;
        for (int i = 0; i < 40; i++) {
            std::string t;
            for (size_t d = 1 + rng() % 4, k = 0; k < d; k++) t += (k ? "/" : "") + level();
            ts.insert(t);
        }
        for (auto& t : ts) {
            re.push_back(std::make_unique<Entry>());
            re.back()->topic = t;
            rx.add(split(re.back()->topic, '/'), re.back().get());
        }
        for (int q = 0; q < 60; q++) {
            std::string f;
            const size_t d = 1 + rng() % 4;
            for (size_t k = 0; k < d; k++) {
                const int r = (int)(rng() % 10);
                f += (k ? "/" : "") + (r < 2 ? std::string("+") : (r == 2 && k + 1 == d ? std::string("#") : level()));
            }
            S want;
            for (auto& t : ts)
                if (filter_matches(split(f, '/'), split(t, '/'))) want.insert(t);
            EXPECT(index_match(rx, f) == want);
        }
        // removing half leaves exactly the other half reachable
        size_t i = 0;
        S left;
        for (auto& t : ts)
            if (i++ % 2) rx.remove(split(t, '/'));
            else left.insert(t);
        S sys_free;
        for (auto& t : left)
            if (t.empty() || t[0] != '$') sys_free.insert(t);
        EXPECT(index_match(rx, "#") == sys_free);
    }
}

// ---- helpers --------------------------------------------------------------------------------------------------------------------------
struct Packed {
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off{0};
    std::vector<uint8_t> op;
    void add(const std::string& k, uint8_t o) {
        bytes.insert(bytes.end(), k.begin(), k.end());
        off.push_back((uint32_t)bytes.size());
        op.push_back(o);
    }
};
static std::string key_of(const std::string& tenant, const std::string& filter, int rid) {
    return bmq::encode_route_key(tenant, filter, 1, std::string("0\0", 2) + "inbox" + std::to_string(rid) + std::string("\0d", 2));
}
static bool cache_get(bmq_route_cache* c, const std::string& tenant, const std::string& topic, uint64_t now, std::vector<uint32_t>& ids, uint64_t& epoch,
                      size_t first_cap = 4) {
    ids.assign(first_cap, 0); // small on purpose: the NOSPACE protocol runs all the time
    for (;;) {
        uint32_t n = 0;
        const int rc = bmq_route_cache_get(c, (const uint8_t*)tenant.data(), (uint32_t)tenant.size(), (const uint8_t*)topic.data(), (uint32_t)topic.size(),
                                           now, ids.data(), (uint32_t)ids.size(), &n, &epoch);
        if (rc == BMQ_E_NOSPACE) { // the row may have grown again by the next call
            ids.assign(n, 0);
            continue;
        }
        ids.resize(n);
        return rc == BMQ_OK;
    }
}
static int is_cached(bmq_route_cache* c, const std::string& tenant, const std::string& filter) {
    return bmq_route_cache_is_cached(c, (const uint8_t*)tenant.data(), (uint32_t)tenant.size(), (const uint8_t*)filter.data(), (uint32_t)filter.size());
}

// ---- 2. single-threaded behaviour --------------------------------------------------------------------------------------------------
static void test_behaviour() {
    bmq_engine e;
    bmq_batcher b(&e);
    bmq_route_cache_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.max_routes_per_tenant = 12;
    cfg.expiry_ms = 1000;
    cfg.shards_per_tenant = 1; // one slice: the LRU order below is the tenant's
    bmq_route_cache* c = nullptr;
    EXPECT(bmq_route_cache_create(&e, &b, &cfg, &c) == BMQ_OK);
    Packed p;
    for (int i = 0; i < 5; i++) p.add(key_of("t", "a/+", i), 0);
    p.add(key_of("t", "a/b", 9), 0);
    p.add(key_of("u", "#", 1), 0);
    EXPECT(bmq_route_cache_apply(c, p.bytes.data(), p.off.data(), p.op.data(), (uint32_t)p.op.size()) == BMQ_OK);
    std::vector<uint32_t> ids;
    uint64_t ep = 0;
    bmq_route_cache_stats st{};
    EXPECT(cache_get(c, "t", "a/b", 10, ids, ep, 16) && ids.size() == 6 && ep == 2); // 5 x a/+ and a/b
    EXPECT(cache_get(c, "t", "a/b", 20, ids, ep, 16) && ids.size() == 6);
    EXPECT(cache_get(c, "t", "x", 20, ids, ep, 16) && ids.empty());
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.hits == 1 && st.misses == 2 && st.entries == 2 && st.cached_routes == 7); // an empty row weighs 1
    EXPECT(is_cached(c, "t", "a/+") == 1 && is_cached(c, "t", "#") == 1 && is_cached(c, "t", "b/#") == 0 && is_cached(c, "nobody", "#") == 0);
    // a mutation of a filter that matches no cached topic leaves the cache alone; one that matches drops exactly those topics
    Packed q;
    q.add(key_of("t", "b/c", 1), 0);
    EXPECT(bmq_route_cache_apply(c, q.bytes.data(), q.off.data(), q.op.data(), 1) == BMQ_OK);
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.invalidations == 0 && st.entries == 2);
    Packed r;
    r.add(key_of("t", "a/#", 7), 0);
    r.add(key_of("t", "a/+", 0), 1);
    EXPECT(bmq_route_cache_apply(c, r.bytes.data(), r.off.data(), r.op.data(), 2) == BMQ_OK);
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.invalidations == 1 && st.entries == 1 && is_cached(c, "t", "a/b") == 0 && is_cached(c, "t", "x") == 1);
    EXPECT(cache_get(c, "t", "a/b", 30, ids, ep, 16) && ids.size() == 6 && ep == 4); // -1 a/+ route, +1 a/#
    // weight-bounded LRU: rows of 6 (a/b), 5, 5 ids: the third load evicts the least recently used
    EXPECT(cache_get(c, "t", "a/c", 31, ids, ep, 16) && ids.size() == 5);
    EXPECT(cache_get(c, "t", "a/b", 32, ids, ep, 16)); // touch a/b: "x" (weight 1) and a/c are older
    EXPECT(cache_get(c, "t", "a/d", 33, ids, ep, 16) && ids.size() == 5);
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.evictions >= 1 && st.cached_routes <= 12 && is_cached(c, "t", "a/b") == 1 && is_cached(c, "t", "a/d") == 1);
    // expire after access
    const uint64_t loads = e.n_match;
    EXPECT(cache_get(c, "t", "a/b", 900, ids, ep, 16) && e.n_match == loads);      // hit, access time moves on
    EXPECT(cache_get(c, "t", "a/b", 1850, ids, ep, 16) && e.n_match == loads);     // 950 ms after the last access: still there
    EXPECT(cache_get(c, "t", "a/b", 2900, ids, ep, 16) && e.n_match == loads + 1); // 1050 ms: expired, reloaded
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.expired == 1);
    { // a whole BatchDistRequest in one call: hits from the cache, all misses (identical ones once) in ONE launch
        const std::string tn_bytes = "tu";
        const uint32_t tenant_off[3] = {0, 1, 2};
        const std::vector<std::string> tps = {"a/b", "a/zz", "a/zz", "zz", "a/zz", "q"};
        const uint32_t tt[6] = {0, 0, 0, 1, 1, 0}; // "a/zz" twice for t and once for u: two launch rows
        std::string bytes;
        std::vector<uint32_t> off{0};
        for (auto& t : tps) {
            bytes += t;
            off.push_back((uint32_t)bytes.size());
        }
        bytes.append(16, '\0');
        uint32_t row[7];
        uint8_t hit[6];
        std::vector<uint32_t> out(2);
        uint64_t need = 0;
        const uint64_t launches = e.n_launch, matched = e.n_match;
        int rc = bmq_route_cache_get_batch(c, (const uint8_t*)tn_bytes.data(), tenant_off, 2, tt, (const uint8_t*)bytes.data(), off.data(), 6, 2950, row, out.data(),
                                           out.size(), &need, hit);
        EXPECT(rc == BMQ_E_NOSPACE && need == 6 + 5 + 5 + 1 + 1 + 0 && e.n_launch == launches + 1 && e.n_match == matched + 4); // 4 distinct misses
        out.resize(need);
        rc = bmq_route_cache_get_batch(c, (const uint8_t*)tn_bytes.data(), tenant_off, 2, tt, (const uint8_t*)bytes.data(), off.data(), 6, 2960, row, out.data(),
                                       out.size(), &need, hit);
        EXPECT(rc == BMQ_OK && e.n_launch == launches + 1); // everything was cached by the first call
        for (int i = 0; i < 6; i++) {
            EXPECT(hit[i] == 1);
            const std::vector<uint32_t> got(out.begin() + row[i], out.begin() + row[i + 1]);
            EXPECT(got == brute(e.model, tt[i] ? "u" : "t", tps[i]));
        }
    }
    { // the sweep: everything idle for a second goes, what was touched since stays
        uint64_t dropped = 0;
        bmq_route_cache_stats_get(c, &st);
        const uint64_t before = st.entries;
        EXPECT(cache_get(c, "t", "a/d", 3500, ids, ep, 16)); // touch (or load) a/d at 3500
        EXPECT(bmq_route_cache_expire(c, 4100, &dropped) == BMQ_OK); // idle since 2960 or earlier: more than a second
        bmq_route_cache_stats_get(c, &st);
        EXPECT(dropped >= 1 && st.entries + dropped >= before && is_cached(c, "t", "a/d") == 1 && is_cached(c, "t", "a/b") == 0); // a/b idle since 2960
    }
    { // a request of direct_batch_topics topics and more goes straight to one launch: no probes, nothing cached
        bmq_route_cache_config big{};
        big.struct_size = sizeof(big);
        big.direct_batch_topics = 4;
        bmq_route_cache* c2 = nullptr;
        EXPECT(bmq_route_cache_create(&e, &b, &big, &c2) == BMQ_OK);
        const uint32_t tenant_off[2] = {0, 1};
        const std::string bytes = std::string("a/ba/ca/da/e") + std::string(16, '\0');
        const uint32_t off[5] = {0, 3, 6, 9, 12}, tt[4] = {0, 0, 0, 0};
        uint32_t row[5], out[64];
        uint8_t hit[4] = {9, 9, 9, 9};
        uint64_t need = 0;
        const uint64_t launches = e.n_launch;
        EXPECT(bmq_route_cache_get_batch(c2, (const uint8_t*)"t", tenant_off, 1, tt, (const uint8_t*)bytes.data(), off, 4, 5000, row, out, 64, &need, hit) == BMQ_OK);
        EXPECT(e.n_launch == launches + 1 && hit[0] == 0 && hit[3] == 0);
        for (int i = 0; i < 4; i++) EXPECT(std::vector<uint32_t>(out + row[i], out + row[i + 1]) == brute(e.model, "t", bytes.substr(off[i], 3)));
        bmq_route_cache_stats s2{};
        bmq_route_cache_stats_get(c2, &s2);
        EXPECT(s2.entries == 0 && s2.misses == 0);
        bmq_route_cache_destroy(c2);
    }
    // rebuild: ids renumbered, nothing of the old generation survives
    Packed nb;
    nb.add(key_of("t", "a/b", 100), 0);
    EXPECT(bmq_route_cache_rebuild(c, nb.bytes.data(), nb.off.data(), 1) == BMQ_OK);
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.entries == 0);
    EXPECT(cache_get(c, "t", "a/b", 3000, ids, ep, 16) && ids == std::vector<uint32_t>{0});
    EXPECT(cache_get(c, "u", "zz", 3000, ids, ep, 16) && ids.empty());
    EXPECT(bmq_route_cache_reset(c) == BMQ_OK && is_cached(c, "t", "#") == 0);
    bmq_route_cache_destroy(c);
}

// ---- 2b. the reference's own TenantRouteCacheTest scenarios (DWT/cache/TenantRouteCacheTest.java), on route ids ------------------------
static std::string group_key(const std::string& tenant, const std::string& filter, const std::string& group) { return bmq::encode_route_key(tenant, filter, 2, group); }
static void test_reference_cases() {
    const std::string TENANT = "tenantA", TOPIC = "sensor/temperature"; // :76-77
    auto apply = [](bmq_route_cache* c, const std::vector<std::pair<std::string, uint8_t>>& ops) {
        Packed p;
        for (auto& o : ops) p.add(o.first, o.second);
        EXPECT(bmq_route_cache_apply(c, p.bytes.data(), p.off.data(), p.op.data(), (uint32_t)p.op.size()) == BMQ_OK);
    };
    auto has = [](const std::vector<uint32_t>& ids, uint32_t id) { return std::find(ids.begin(), ids.end(), id) != ids.end(); };
    std::vector<uint32_t> ids;
    uint64_t ep = 0;
    { // shouldLoadAndIndexRoutesOnFirstAccess :110-125, shouldReuseCachedValueWithoutReload :127-143
        bmq_engine e;
        bmq_batcher b(&e);
        bmq_route_cache* c = nullptr;
        EXPECT(bmq_route_cache_create(&e, &b, nullptr, &c) == BMQ_OK);
        const std::string existing = key_of(TENANT, TOPIC, 1);
        apply(c, {{existing, 0}});
        EXPECT(is_cached(c, TENANT, TOPIC) == 0);
        EXPECT(cache_get(c, TENANT, TOPIC, 1, ids, ep, 16) && ids == std::vector<uint32_t>{e.model[existing]} && e.n_match == 1);
        EXPECT(is_cached(c, TENANT, TOPIC) == 1);
        EXPECT(cache_get(c, TENANT, TOPIC, 2, ids, ep, 16) && ids == std::vector<uint32_t>{e.model[existing]} && e.n_match == 1); // matcher called once
        // shouldApplyAddRoutesTask :187-214: a normal route on the topic and a group route on "sensor/#" show up in the cached topic
        const std::string new_normal = key_of(TENANT, TOPIC, 2), new_group = group_key(TENANT, "sensor/#", "groupA");
        apply(c, {{new_normal, 0}, {new_group, 0}});
        EXPECT(cache_get(c, TENANT, TOPIC, 3, ids, ep, 16) && ids.size() == 3 && has(ids, e.model[new_normal]) && has(ids, e.model[new_group]));
        // shouldApplyRemoveRoutesTask :216-254: the normal route and the emptied group go, a group that only lost a member stays (its key
        // -- hence its id -- is untouched: the membership lives in the value)
        const std::string updatable = group_key(TENANT, "sensor/+", "groupUpdate");
        apply(c, {{updatable, 0}});
        EXPECT(cache_get(c, TENANT, TOPIC, 4, ids, ep, 16) && ids.size() == 4);
        const uint32_t id_updatable = e.model[updatable], id_existing = e.model[existing];
        apply(c, {{existing, 1}, {new_group, 1}});
        EXPECT(cache_get(c, TENANT, TOPIC, 5, ids, ep, 16) && ids.size() == 2 && !has(ids, id_existing) && has(ids, id_updatable) && has(ids, e.model[new_normal]));
        bmq_route_cache_destroy(c);
    }
    { // shouldQueueTasksUntilLoadCompletes :256-292: an AddRoutes task arrives while the load of the topic is still running
        bmq_engine e;
        bmq_batcher b(&e);
        bmq_route_cache* c = nullptr;
        EXPECT(bmq_route_cache_create(&e, &b, nullptr, &c) == BMQ_OK);
        const std::string existing = key_of(TENANT, TOPIC, 1), new_normal = key_of(TENANT, TOPIC, 2);
        apply(c, {{existing, 0}});
        e.hold_loads = true;
        std::vector<uint32_t> first;
        std::thread loader([&] {
            uint64_t ep2 = 0;
            EXPECT(cache_get(c, TENANT, TOPIC, 1, first, ep2, 16));
        });
        while (e.loads_waiting.load() == 0) std::this_thread::sleep_for(std::chrono::microseconds(50)); // loadStarted
        apply(c, {{new_normal, 0}});                                                                     // refresh(AddRoutesTask) meanwhile
        e.hold_loads = false;                                                                            // allowLoad
        loader.join();
        EXPECT(first == std::vector<uint32_t>{e.model[existing]}); // the future completes with what the load saw
        EXPECT(cache_get(c, TENANT, TOPIC, 2, ids, ep, 16) && ids.size() == 2 && has(ids, e.model[new_normal])); // ... and the next get sees the addition
        bmq_route_cache_stats st{};
        bmq_route_cache_stats_get(c, &st);
        EXPECT(st.stale_loads == 1); // the overtaken load was not cached
        bmq_route_cache_destroy(c);
    }
    { // shouldBoundZeroRouteTopicsByMaxWeight :359-417: 100 topics without routes, max weight 10 -> at most 12 cached / indexed
        bmq_engine e;
        bmq_batcher b(&e);
        bmq_route_cache_config cfg{};
        cfg.struct_size = sizeof(cfg);
        cfg.max_routes_per_tenant = 10;
        cfg.shards_per_tenant = 1;
        bmq_route_cache* c = nullptr;
        EXPECT(bmq_route_cache_create(&e, &b, &cfg, &c) == BMQ_OK);
        for (int i = 0; i < 100; i++) EXPECT(cache_get(c, TENANT, "sensor/t" + std::to_string(i), 1, ids, ep, 16) && ids.empty());
        bmq_route_cache_stats st{};
        bmq_route_cache_stats_get(c, &st);
        EXPECT(st.entries <= 12 && st.entries >= 1 && st.cached_routes <= 12 && st.evictions >= 88);
        int indexed = 0;
        for (int i = 0; i < 100; i++) indexed += is_cached(c, TENANT, "sensor/t" + std::to_string(i));
        EXPECT(indexed == (int)st.entries && is_cached(c, TENANT, "#") == 1); // the index is bounded along with the cache
        bmq_route_cache_destroy(c);
    }
    { // shouldCleanupIndexOnExpiryAndExplicitInvalidation :419-470 (EXPIRY = 1 minute :78)
        bmq_engine e;
        bmq_batcher b(&e);
        bmq_route_cache* c = nullptr;
        EXPECT(bmq_route_cache_create(&e, &b, nullptr, &c) == BMQ_OK);
        apply(c, {{key_of(TENANT, TOPIC, 1), 0}});
        EXPECT(cache_get(c, TENANT, TOPIC, 0, ids, ep, 16) && ids.size() == 1 && is_cached(c, TENANT, TOPIC) == 1);
        uint64_t dropped = 0;
        EXPECT(bmq_route_cache_expire(c, 60001, &dropped) == BMQ_OK && dropped == 1 && is_cached(c, TENANT, TOPIC) == 0); // ticker.advance(EXPIRY + 1 ms); cleanUp()
        EXPECT(cache_get(c, TENANT, TOPIC, 60002, ids, ep, 16) && is_cached(c, TENANT, TOPIC) == 1 && e.n_match == 2);    // loaded again
        EXPECT(bmq_route_cache_reset(c) == BMQ_OK && is_cached(c, TENANT, TOPIC) == 0);                                   // explicit invalidation
        bmq_route_cache_destroy(c);
    }
}

// ---- 2c. fan-out caps through the cache: getMatch = IMatchedRoutes.routes() (TenantRouteCache.java:299-301) ----------------------------
// matchAll over the model in KV order with MatchedRoutes' rules (MatchedRoutes.java:87-141): the reference's result for one (tenant, topic)
struct Throttle {
    int type;
    uint32_t id;
    int max;
    bool operator==(const Throttle& o) const { return type == o.type && id == o.id && max == o.max; }
};
static std::vector<uint32_t> matched_routes(const std::map<std::string, uint32_t>& model, const std::string& tenant, const std::string& topic, int max_pf, int max_gf,
                                            std::vector<Throttle>* events) {
    std::vector<uint32_t> routes;
    int persistent = 0, groups = 0;
    const auto tl = bmq::cache::split(topic, '/');
    for (auto& kv : model) { // std::map: unsigned byte order of the keys == the KV iterator's order
        bmq::RouteKeyParts kp;
        if (!bmq::decode_route_key(kv.first, kp) || kp.tenant != tenant) continue;
        if (!bmq::cache::filter_matches(bmq::cache::split(kp.esc_filter, '\0'), tl)) continue;
        if (kp.flag == 1) { // addNormalMatching :87-109
            const bool is_persistent = kp.receiver.size() >= 2 && kp.receiver[0] == '1' && kp.receiver[1] == '\0'; // subBrokerId == 1
            if (is_persistent) {
                if (persistent < max_pf) persistent++;
                else {
                    if (events) events->push_back({0, kv.second, max_pf});
                    continue;
                }
            }
            routes.push_back(kv.second);
        } else { // putGroupMatching :119-141
            if (groups + 1 <= max_gf) {
                groups++;
                routes.push_back(kv.second);
            } else if (events) events->push_back({1, kv.second, max_gf});
        }
    }
    std::sort(routes.begin(), routes.end());
    return routes;
}
static std::string persistent_key(const std::string& tenant, const std::string& filter, int rid) {
    return bmq::encode_route_key(tenant, filter, 1, std::string("1\0", 2) + "inbox" + std::to_string(rid) + std::string("\0d", 2));
}
struct EventLog {
    std::mutex mu;
    std::vector<std::pair<std::string, Throttle>> got; // "tenant|topic" -> event
    static void sink(void* user, const uint8_t* tenant, uint32_t tl, const uint8_t* topic, uint32_t pl, int32_t type, uint32_t id, int32_t max) {
        EventLog* l = (EventLog*)user;
        std::lock_guard<std::mutex> g(l->mu);
        l->got.push_back({std::string((const char*)tenant, tl) + "|" + std::string((const char*)topic, pl), Throttle{type, id, max}});
    }
    std::vector<Throttle> take(const std::string& tenant, const std::string& topic) {
        std::lock_guard<std::mutex> g(mu);
        std::vector<Throttle> out;
        for (auto& p : got)
            if (p.first == tenant + "|" + topic) out.push_back(p.second);
        got.clear();
        return out;
    }
};
static void test_caps() {
    bmq_engine e;
    bmq_batcher b(&e);
    bmq_route_cache_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.default_max_persistent_fanout = 3;
    cfg.default_max_group_fanout = 2;
    bmq_route_cache* c = nullptr;
    EXPECT(bmq_route_cache_create(&e, &b, &cfg, &c) == BMQ_OK);
    EventLog log;
    EXPECT(bmq_route_cache_set_event_sink(c, EventLog::sink, &log) == BMQ_OK);
    auto apply = [&](const std::vector<std::pair<std::string, uint8_t>>& ops) {
        Packed p;
        for (auto& o : ops) p.add(o.first, o.second);
        EXPECT(bmq_route_cache_apply(c, p.bytes.data(), p.off.data(), p.op.data(), (uint32_t)p.op.size()) == BMQ_OK);
    };
    // 5 persistent + 2 transient inboxes and 4 shared-subscription groups match s/t; an unrelated topic matches 1 persistent route
    std::vector<std::pair<std::string, uint8_t>> ops;
    for (int i = 0; i < 5; i++) ops.push_back({persistent_key("T", i % 2 ? "s/+" : "s/t", i), 0});
    for (int i = 0; i < 2; i++) ops.push_back({key_of("T", "s/#", 100 + i), 0});
    for (int i = 0; i < 4; i++) ops.push_back({group_key("T", i % 2 ? "s/t" : "+/t", "g" + std::to_string(i)), 0});
    ops.push_back({persistent_key("T", "x", 9), 0});
    apply(ops);
    std::vector<uint32_t> ids;
    uint64_t ep = 0;
    auto check = [&](const std::string& topic, int pf, int gf, bool expect_load, int line) {
        const uint64_t before = e.n_match;
        std::vector<Throttle> want_ev;
        const auto want = matched_routes(e.model, "T", topic, pf, gf, &want_ev);
        const bool ok = cache_get(c, "T", topic, 10, ids, ep, 2); // small first buffer: the NOSPACE round trip must not double the events
        const auto got_ev = log.take("T", topic);
        const bool loaded = e.n_match != before;
        if (!ok || ids != want || loaded != expect_load || (loaded ? !(got_ev == want_ev) : !got_ev.empty())) {
            fprintf(stderr, "cache_fuzz: caps check failed (line %d): topic %s caps %d/%d got %zu ids want %zu, loaded %d want %d, events %zu want %zu\n", line,
                    topic.c_str(), pf, gf, ids.size(), want.size(), (int)loaded, (int)expect_load, got_ev.size(), loaded ? want_ev.size() : 0);
            g_fail++;
        }
    };
    // NOSPACE on a miss loads twice (the row was not cached the first time only if it was stale): events of both loads are compared below
    // against ONE load, so the first call uses a buffer that fits
    auto check_fit = [&](const std::string& topic, int pf, int gf, bool expect_load, int line) {
        const uint64_t before = e.n_match;
        std::vector<Throttle> want_ev;
        const auto want = matched_routes(e.model, "T", topic, pf, gf, &want_ev);
        const bool ok = cache_get(c, "T", topic, 10, ids, ep, 64);
        const auto got_ev = log.take("T", topic);
        const bool loaded = e.n_match != before;
        if (!ok || ids != want || loaded != expect_load || (loaded ? !(got_ev == want_ev) : !got_ev.empty())) {
            fprintf(stderr, "cache_fuzz: caps check failed (line %d): topic %s caps %d/%d got %zu ids want %zu, loaded %d want %d, events %zu want %zu\n", line,
                    topic.c_str(), pf, gf, ids.size(), want.size(), (int)loaded, (int)expect_load, got_ev.size(), loaded ? want_ev.size() : 0);
            g_fail++;
        }
    };
    (void)check;
    check_fit("s/t", 3, 2, true, __LINE__);   // miss: 3 of 5 persistent, 2 of 4 groups, both transient ones; 2 + 2 events
    EXPECT(ids.size() == 3 + 2 + 2);
    check_fit("s/t", 3, 2, false, __LINE__);  // hit: the capped row, no events
    check_fit("x", 3, 2, true, __LINE__);     // a row no cap binds on: not even classified
    bmq_route_cache_stats st{};
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.cached_routes == 7 + 1); // the weigher counts the capped row (TenantRouteCache.java:108)
    // refresh(AddRoutes): a persistent route whose key sorts in front of the admitted ones, and a group behind the admitted ones
    apply({{persistent_key("T", "+/t", 50), 0}, {group_key("T", "s/t", "zz"), 0}});
    check_fit("s/t", 3, 2, true, __LINE__);   // dropped by the mutation, re-matched: caps in key order again, events again
    check_fit("s/t", 3, 2, false, __LINE__);
    // refresh(RemoveRoutes) of an admitted persistent route: the next one in key order moves up
    apply({{persistent_key("T", "+/t", 50), 1}});
    check_fit("s/t", 3, 2, true, __LINE__);
    // MatchedRoutes.adjust: raise the persistent cap -- the cached row sits at the old cap, so it must be re-matched
    EXPECT(bmq_route_cache_set_caps(c, (const uint8_t*)"T", 1, 4, 2) == BMQ_OK);
    check_fit("s/t", 4, 2, true, __LINE__);
    check_fit("s/t", 4, 2, false, __LINE__);
    check_fit("x", 4, 2, false, __LINE__);    // 1 persistent route, nowhere near a cap: the entry adopts the new caps without a load
    // lower the group cap below what is cached: re-matched (the reference clamps arbitrary groups; a reload clamps in key order)
    EXPECT(bmq_route_cache_set_caps(c, (const uint8_t*)"T", 1, 4, 1) == BMQ_OK);
    check_fit("s/t", 4, 1, true, __LINE__);
    // raise both far beyond the row: one more load (the row was AT the old caps), then hits
    EXPECT(bmq_route_cache_set_caps(c, (const uint8_t*)"T", 1, 1000, 1000) == BMQ_OK);
    check_fit("s/t", 1000, 1000, true, __LINE__);
    EXPECT(ids.size() == 5 + 2 + 5);
    check_fit("s/t", 1000, 1000, false, __LINE__);
    // lower again: the row (12 ids) was never classified under 1000 / 1000, so how many of them are persistent is unknown -> re-matched
    EXPECT(bmq_route_cache_set_caps(c, (const uint8_t*)"T", 1, 5, 5) == BMQ_OK);
    check_fit("s/t", 5, 5, true, __LINE__);
    // raise by one: the row holds 5 persistent routes == the old cap, a 6th might have been thrown away -> re-matched
    EXPECT(bmq_route_cache_set_caps(c, (const uint8_t*)"T", 1, 6, 7) == BMQ_OK);
    check_fit("s/t", 6, 7, true, __LINE__);
    // lower, but not below what the row holds (5 persistent, 5 groups, counted by the last load): the entry just carries the new caps
    EXPECT(bmq_route_cache_set_caps(c, (const uint8_t*)"T", 1, 5, 5) == BMQ_OK);
    check_fit("s/t", 5, 5, false, __LINE__);
    bmq_route_cache_tenant_stats ts{};
    EXPECT(bmq_route_cache_tenant_stats_get(c, (const uint8_t*)"T", 1, &ts) == BMQ_OK && ts.max_persistent_fanout == 5 && ts.max_group_fanout == 5 &&
           ts.entries == 2 && ts.hits >= 5 && ts.misses >= 7 && ts.cached_routes == 12 + 1);
    EXPECT(bmq_route_cache_tenant_stats_get(c, (const uint8_t*)"nobody", 6, &ts) == BMQ_E_STATE);
    // the same through get_batch (cached path and direct path) and get_async
    EXPECT(bmq_route_cache_set_caps(c, (const uint8_t*)"T", 1, 2, 1) == BMQ_OK);
    {
        const uint32_t tenant_off[2] = {0, 1};
        const std::string bytes = std::string("s/txs/t") + std::string(16, '\0');
        const uint32_t off[4] = {0, 3, 4, 7}, tt[3] = {0, 0, 0};
        uint32_t row[4], out[64];
        uint64_t need = 0;
        EXPECT(bmq_route_cache_get_batch(c, (const uint8_t*)"T", tenant_off, 1, tt, (const uint8_t*)bytes.data(), off, 3, 20, row, out, 64, &need, nullptr) == BMQ_OK);
        std::vector<Throttle> want_ev;
        const auto want = matched_routes(e.model, "T", "s/t", 2, 1, &want_ev);
        EXPECT(std::vector<uint32_t>(out + row[0], out + row[1]) == want && std::vector<uint32_t>(out + row[2], out + row[3]) == want);
        EXPECT(std::vector<uint32_t>(out + row[1], out + row[2]) == matched_routes(e.model, "T", "x", 2, 1, nullptr));
        EXPECT(log.take("T", "s/t") == want_ev); // identical misses of one request are ONE load: one set of events
        bmq_route_cache_config big = cfg;
        big.direct_batch_topics = 2;
        bmq_route_cache* c2 = nullptr;
        EXPECT(bmq_route_cache_create(&e, &b, &big, &c2) == BMQ_OK);
        EventLog log2;
        bmq_route_cache_set_event_sink(c2, EventLog::sink, &log2);
        EXPECT(bmq_route_cache_get_batch(c2, (const uint8_t*)"T", tenant_off, 1, tt, (const uint8_t*)bytes.data(), off, 3, 20, row, out, 64, &need, nullptr) == BMQ_OK);
        std::vector<Throttle> ev32;
        const auto want32 = matched_routes(e.model, "T", "s/t", 3, 2, &ev32); // c2 has its own (default) caps
        EXPECT(need == 2 * want32.size() + 1 && std::vector<uint32_t>(out + row[0], out + row[1]) == want32 &&
               std::vector<uint32_t>(out + row[2], out + row[3]) == want32 && row[2] - row[1] == 1);
        auto ev2 = log2.take("T", "s/t");
        std::vector<Throttle> twice = ev32;
        twice.insert(twice.end(), ev32.begin(), ev32.end()); // the direct path matches every row it is given: both copies report
        EXPECT(ev2 == twice);
        bmq_route_cache_destroy(c2);
    }
    {
        struct Got {
            std::vector<uint32_t> ids;
            std::atomic<int> done{0};
        } got;
        apply({{key_of("T", "s/t", 777), 0}}); // drop s/t so that the future is a miss
        auto cb = +[](void* user, int status, const uint32_t* ids, uint32_t n, uint64_t) {
            Got* g = (Got*)user;
            EXPECT(status == BMQ_OK);
            g->ids.assign(ids, ids + n);
            g->done = 1;
        };
        EXPECT(bmq_route_cache_get_async(c, (const uint8_t*)"T", 1, (const uint8_t*)"s/t", 3, 30, cb, &got) == BMQ_OK);
        while (!got.done) std::this_thread::sleep_for(std::chrono::microseconds(50));
        std::vector<Throttle> want_ev;
        EXPECT(got.ids == matched_routes(e.model, "T", "s/t", 2, 1, &want_ev) && log.take("T", "s/t") == want_ev);
        got.done = 0;
        EXPECT(bmq_route_cache_get_async(c, (const uint8_t*)"T", 1, (const uint8_t*)"s/t", 3, 31, cb, &got) == BMQ_OK && got.done == 1); // hit: inline
        EXPECT(got.ids == matched_routes(e.model, "T", "s/t", 2, 1, nullptr) && log.take("T", "s/t").empty());
    }
    bmq_route_cache_destroy(c);
}

// ---- 2d. tenant lifecycle (SubscriptionCache.java:79-107) and per-tenant meters (TenantRouteCache.java:141-147) -----------------------------
static long rss_kb() {
    long pages = 0, rss = 0;
    if (FILE* f = fopen("/proc/self/statm", "r")) {
        if (fscanf(f, "%ld %ld", &pages, &rss) != 2) rss = 0;
        fclose(f);
    }
    return rss * 4;
}
static void test_lifecycle() {
    bmq_engine e;
    bmq_batcher b(&e);
    bmq_route_cache_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.expiry_ms = 1000; // tenant_idle_ms = 0 -> 2 x expiry
    bmq_route_cache* c = nullptr;
    EXPECT(bmq_route_cache_create(&e, &b, &cfg, &c) == BMQ_OK);
    Packed p;
    p.add(key_of("keep", "a/#", 1), 0);
    EXPECT(bmq_route_cache_apply(c, p.bytes.data(), p.off.data(), p.op.data(), 1) == BMQ_OK);
    std::vector<uint32_t> ids;
    uint64_t ep = 0, dropped = 0;
    bmq_route_cache_stats st{};
    bmq_route_cache_tenant_stats ts{};
    EXPECT(cache_get(c, "keep", "a/b", 0, ids, ep, 8) && cache_get(c, "gone", "a/b", 0, ids, ep, 8));
    EXPECT(bmq_route_cache_set_caps(c, (const uint8_t*)"gone", 4, 7, 9) == BMQ_OK);
    // entries expire after 1 s idle, the tenant's cache after 2 s without a get; isCached / refresh do not keep a tenant alive
    EXPECT(cache_get(c, "keep", "a/b", 900, ids, ep, 8) && cache_get(c, "keep", "a/b", 1700, ids, ep, 8));
    EXPECT(is_cached(c, "gone", "#") == 1);
    EXPECT(bmq_route_cache_expire(c, 1999, &dropped) == BMQ_OK);
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.tenants == 2 && st.tenants_expired == 0 && dropped == 1); // gone's entry expired, its (empty) cache is still there
    EXPECT(bmq_route_cache_tenant_stats_get(c, (const uint8_t*)"gone", 4, &ts) == BMQ_OK && ts.entries == 0 && ts.misses == 1 && ts.max_persistent_fanout == 7);
    EXPECT(bmq_route_cache_expire(c, 2000, &dropped) == BMQ_OK);
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.tenants == 1 && st.tenants_expired == 1 && st.misses == 2 && st.hits == 2); // the counters of the destroyed cache still count
    EXPECT(bmq_route_cache_tenant_stats_get(c, (const uint8_t*)"gone", 4, &ts) == BMQ_E_STATE); // ... its meters are gone (stopCounting, :304-311)
    EXPECT(bmq_route_cache_tenant_stats_get(c, (const uint8_t*)"keep", 4, &ts) == BMQ_OK && ts.hits == 2 && ts.entries == 1);
    EXPECT(cache_get(c, "gone", "a/b", 2100, ids, ep, 8)); // comes back with its first loaded row -- and with the caps set for it
    EXPECT(bmq_route_cache_tenant_stats_get(c, (const uint8_t*)"gone", 4, &ts) == BMQ_OK && ts.max_persistent_fanout == 7 && ts.max_group_fanout == 9 && ts.misses == 1);
    // "10 k tenants come and go": waves of tenants, each swept two idle periods later -- the table does not grow, memory is returned
    EXPECT(bmq_route_cache_expire(c, 9000, &dropped) == BMQ_OK && dropped == 2); // keep and gone go first
    long rss_after_first = 0;
    for (int wave = 0; wave < 6; wave++) {
        const uint64_t now = 10000 + (uint64_t)wave * 5000;
        for (int i = 0; i < 10000; i++) {
            const std::string tn = "w" + std::to_string(wave) + "-" + std::to_string(i);
            EXPECT(cache_get(c, tn, "a/b", now, ids, ep, 8));
        }
        bmq_route_cache_stats_get(c, &st);
        EXPECT(st.tenants == 10000 && st.entries == 10000);
        EXPECT(bmq_route_cache_expire(c, now + 2500, &dropped) == BMQ_OK);
        bmq_route_cache_stats_get(c, &st);
        EXPECT(st.tenants == 0 && st.entries == 0 && dropped == 10000);
        if (wave == 1) rss_after_first = rss_kb();
    }
    const long rss_end = rss_kb();
    printf("  lifecycle: 6 waves of 10000 tenants, RSS after wave 2: %ld KB, after wave 6: %ld KB\n", rss_after_first, rss_end);
#if !defined(__SANITIZE_ADDRESS__) // (ASan parks freed memory in its quarantine: RSS says nothing there)
    EXPECT(rss_end < rss_after_first + rss_after_first / 4 + 8192); // flat: the allocator may keep some, the table must not
#endif
    bmq_route_cache_stats_get(c, &st);
    EXPECT(st.tenants_expired == 60000 + 3 && st.misses == 60000 + 3); // keep, gone, gone again + the waves
    bmq_route_cache_destroy(c);
}

// ---- 3. getters against a mutator ---------------------------------------------------------------------------------------------------
static void test_concurrent(uint64_t seed, int n_threads, int ms) {
    bmq_engine e;
    e.match_delay_us = 40;
    bmq_batcher b(&e);
    bmq_route_cache_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.max_routes_per_tenant = 400;
    cfg.shards_per_tenant = 4;
    cfg.mutation_log_entries = 64; // the log is cut all the time: loads older than it must be refused, not trusted
    bmq_route_cache* c = nullptr;
    EXPECT(bmq_route_cache_create(&e, &b, &cfg, &c) == BMQ_OK);
    const std::vector<std::string> tenants = {"t", "u", "tenant-three"};
    const std::vector<std::string> alpha = {"a", "b", "c", "", "$s"};
    auto topic_of = [&](std::mt19937_64& r) {
        std::string t;
        for (size_t d = 1 + r() % 3, k = 0; k < d; k++) t += (k ? "/" : "") + alpha[r() % alpha.size()];
        return t;
    };
    auto filter_of = [&](std::mt19937_64& r) {
        std::string f;
        const size_t d = 1 + r() % 3;
        for (size_t k = 0; k < d; k++) {
            const int x = (int)(r() % 8);
            f += (k ? "/" : "") + (x == 0 ? std::string("+") : (x == 1 && k + 1 == d ? std::string("#") : alpha[r() % alpha.size()]));
        }
        return f;
    };
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> n_get{0}, n_apply{0};
    std::vector<std::thread> th;
    for (int w = 0; w < n_threads; w++)
        th.emplace_back([&, w]() {
            std::mt19937_64 r(seed * 977 + (uint64_t)w);
            std::vector<uint32_t> ids;
            uint64_t ep = 0;
            while (!stop) {
                const std::string& tn = tenants[r() % tenants.size()];
                const std::string tp = topic_of(r);
                if (!cache_get(c, tn, tp, 1000, ids, ep)) {
                    EXPECT(!"get failed");
                    break;
                }
                std::map<std::string, uint32_t> snap;
                {
                    std::lock_guard<std::mutex> g(e.mu);
                    EXPECT(ep < e.history.size());
                    snap = e.history[ep < e.history.size() ? ep : 0];
                }
                EXPECT(ids == brute(snap, tn, tp)); // whatever came back is the truth of the epoch it names
                n_get++;
            }
        });
    // two more getters use the future-shaped call: hits are called back inline, misses from the stand-in's dispatcher thread
    struct AsyncCtx {
        bmq_engine* e;
        std::string tenant, topic;
        std::atomic<uint64_t>* n_cb;
    };
    auto on_async = +[](void* user, int status, const uint32_t* ids, uint32_t n, uint64_t epoch) {
        std::unique_ptr<AsyncCtx> a((AsyncCtx*)user);
        EXPECT(status == BMQ_OK);
        std::map<std::string, uint32_t> snap;
        {
            std::lock_guard<std::mutex> g(a->e->mu);
            EXPECT(epoch < a->e->history.size());
            snap = a->e->history[epoch < a->e->history.size() ? epoch : 0];
        }
        EXPECT(std::vector<uint32_t>(ids, ids + n) == brute(snap, a->tenant, a->topic));
        a->n_cb->fetch_add(1);
    };
    std::atomic<uint64_t> n_async{0}, n_async_cb{0};
    for (int w = 0; w < 2; w++)
        th.emplace_back([&, w]() {
            std::mt19937_64 r(seed * 55 + (uint64_t)w);
            while (!stop) {
                auto a = new AsyncCtx{&e, tenants[r() % tenants.size()], topic_of(r), &n_async_cb};
                const int rc = bmq_route_cache_get_async(c, (const uint8_t*)a->tenant.data(), (uint32_t)a->tenant.size(), (const uint8_t*)a->topic.data(),
                                                         (uint32_t)a->topic.size(), 1000, on_async, a);
                EXPECT(rc == BMQ_OK);
                n_async++;
                if (n_async.load() - n_async_cb.load() > 2000) std::this_thread::sleep_for(std::chrono::microseconds(100)); // bounded backlog
            }
        });
    th.emplace_back([&]() { // and one asks for whole batches
        std::mt19937_64 r(seed * 91);
        std::string tn_bytes;
        std::vector<uint32_t> tenant_off{0};
        for (auto& t : tenants) {
            tn_bytes += t;
            tenant_off.push_back((uint32_t)tn_bytes.size());
        }
        while (!stop) {
            const uint32_t n = 1 + (uint32_t)(r() % 24);
            std::string bytes;
            std::vector<uint32_t> off{0}, tt(n);
            std::vector<std::string> tps(n);
            for (uint32_t i = 0; i < n; i++) {
                tps[i] = topic_of(r);
                tt[i] = (uint32_t)(r() % tenants.size());
                bytes += tps[i];
                off.push_back((uint32_t)bytes.size());
            }
            bytes.append(16, '\0');
            std::vector<uint32_t> row(n + 1), out(256);
            uint64_t need = 0;
            int rc;
            while ((rc = bmq_route_cache_get_batch(c, (const uint8_t*)tn_bytes.data(), tenant_off.data(), (uint32_t)tenants.size(), tt.data(),
                                                   (const uint8_t*)bytes.data(), off.data(), n, 1000, row.data(), out.data(), out.size(), &need, nullptr)) ==
                   BMQ_E_NOSPACE)
                out.resize(need + 16);
            EXPECT(rc == BMQ_OK);
            // rows of one call may come from different epochs (hits) -- each must be the truth of SOME epoch up to now
            std::vector<std::map<std::string, uint32_t>> hist;
            {
                std::lock_guard<std::mutex> g(e.mu);
                hist = e.history;
            }
            for (uint32_t i = 0; i < n && rc == BMQ_OK; i++) {
                const std::vector<uint32_t> got(out.begin() + row[i], out.begin() + row[i + 1]);
                bool ok = false;
                for (size_t ep = hist.size(); ep-- > 1 && !ok;) ok = got == brute(hist[ep], tenants[tt[i]], tps[i]);
                EXPECT(ok);
            }
            n_get += n;
        }
    });
    std::thread mut([&]() {
        std::mt19937_64 r(seed * 31 + 5);
        std::vector<std::string> live;
        while (!stop) {
            Packed p;
            const size_t n = 1 + r() % 6;
            for (size_t i = 0; i < n; i++) {
                if (!live.empty() && r() % 2) {
                    const size_t k = r() % live.size();
                    p.add(live[k], 1);
                    live.erase(live.begin() + (long)k);
                } else {
                    live.push_back(key_of(tenants[r() % tenants.size()], filter_of(r), (int)(r() % 50)));
                    p.add(live.back(), 0);
                }
            }
            EXPECT(bmq_route_cache_apply(c, p.bytes.data(), p.off.data(), p.op.data(), (uint32_t)p.op.size()) == BMQ_OK);
            n_apply++;
            std::this_thread::sleep_for(std::chrono::microseconds(150));
        }
    });
    std::atomic<uint64_t> n_sweeps{0};
    std::thread sweeper([&]() { // every tenant's cache is destroyed over and over while the getters hold pointers into the table
        while (!stop) {
            uint64_t dropped = 0;
            EXPECT(bmq_route_cache_expire(c, 1000 + 2 * 60000, &dropped) == BMQ_OK); // the getters' clock stands at 1000: everything is idle
            n_sweeps++;
            std::this_thread::sleep_for(std::chrono::milliseconds(3));
        }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(ms));
    stop = true;
    for (auto& t : th) t.join();
    mut.join();
    sweeper.join();
    for (int spin = 0; spin < 20000 && n_async_cb.load() < n_async.load(); spin++) std::this_thread::sleep_for(std::chrono::microseconds(200));
    EXPECT(n_async_cb.load() == n_async.load() && n_async.load() > 0); // every future completed
    // settled: what the cache serves now is the truth of the FINAL model -- a load overtaken by a mutation was not cached
    e.match_delay_us = 0;
    std::mt19937_64 r(seed);
    std::vector<uint32_t> ids;
    uint64_t ep = 0;
    uint64_t served = 0;
    for (auto& tn : tenants)
        for (int q = 0; q < 400; q++) {
            const std::string tp = topic_of(r);
            const uint64_t before = e.n_match;
            EXPECT(cache_get(c, tn, tp, 1000, ids, ep));
            served += e.n_match == before;
            EXPECT(ids == brute(e.model, tn, tp));
        }
    bmq_route_cache_stats st{};
    bmq_route_cache_stats_get(c, &st);
    printf("  concurrent: %llu async gets, ", (unsigned long long)n_async.load());
    printf("%llu gets, %llu applies, hits %llu misses %llu invalidations %llu stale loads refused %llu evictions %llu; %llu of 1200 final "
           "probes served from the cache\n",
           (unsigned long long)n_get.load(), (unsigned long long)n_apply.load(), (unsigned long long)st.hits, (unsigned long long)st.misses,
           (unsigned long long)st.invalidations, (unsigned long long)st.stale_loads, (unsigned long long)st.evictions, (unsigned long long)served);
    EXPECT(st.hits > 0 && st.invalidations > 0 && st.tenants_expired > 0 && n_sweeps.load() > 0);
    bmq_route_cache_destroy(c);
}

// ---- hit-path throughput (not a test): cache_fuzz perf <threads> <calls per thread> ------------------------------------------------
static void perf(int n_threads, int calls) {
    bmq_engine e;
    bmq_batcher b(&e);
    bmq_route_cache_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.max_routes_per_tenant = 1ull << 40;
    bmq_route_cache* c = nullptr;
    bmq_route_cache_create(&e, &b, &cfg, &c);
    const int n_tenants = 1000, n_topics = 100000;
    std::mt19937_64 r(7);
    std::vector<double> cdf(n_tenants);
    double acc = 0;
    for (int i = 0; i < n_tenants; i++) cdf[i] = (acc += 1.0 / (i + 1));
    std::vector<std::pair<std::string, std::string>> q(n_topics);
    for (auto& p : q) {
        const double u = (r() >> 11) * (1.0 / 9007199254740992.0) * acc;
        const int t = (int)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
        p = {"tenant" + std::to_string(1000000 + t), "l0_" + std::to_string(r() % 8) + "/l1_" + std::to_string(r() % 64) + "/l2_" + std::to_string(r() % 512) +
                                                          "/l3_" + std::to_string(r() % 64)};
    }
    std::vector<uint32_t> ids;
    uint64_t ep;
    for (auto& p : q) cache_get(c, p.first, p.second, 1, ids, ep, 64); // load
    std::atomic<uint64_t> sum{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int w = 0; w < n_threads; w++)
        th.emplace_back([&, w]() {
            std::mt19937_64 rr(w);
            uint32_t out[64], n = 0;
            uint64_t epoch = 0, local = 0;
            for (int i = 0; i < calls; i++) {
                auto& p = q[rr() % q.size()];
                bmq_route_cache_get(c, (const uint8_t*)p.first.data(), (uint32_t)p.first.size(), (const uint8_t*)p.second.data(), (uint32_t)p.second.size(), 2,
                                    out, 64, &n, &epoch);
                local += n;
            }
            sum += local;
        });
    for (auto& t : th) t.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bmq_route_cache_stats st{};
    bmq_route_cache_stats_get(c, &st);
    printf("perf: %d threads x %d hits in %.3f s = %.2f M gets/s (entries %llu, hits %llu, misses %llu)\n", n_threads, calls, sec,
           n_threads * (double)calls / sec / 1e6, (unsigned long long)st.entries, (unsigned long long)st.hits, (unsigned long long)st.misses);
    bmq_route_cache_destroy(c);
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "perf") {
        perf(argc > 2 ? atoi(argv[2]) : 8, argc > 3 ? atoi(argv[3]) : 1000000);
        return 0;
    }
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int threads = argc > 2 ? atoi(argv[2]) : 6;
    const int ms = argc > 3 ? atoi(argv[3]) : 1500;
    test_topic_index(seed);
    test_behaviour();
    test_reference_cases();
    test_caps();
    test_lifecycle();
    test_concurrent(seed, threads, ms);
    if (g_fail) {
        fprintf(stderr, "cache_fuzz FAILED: %d\n", g_fail);
        return 1;
    }
    printf("cache_fuzz ok: seed %llu\n", (unsigned long long)seed);
    return 0;
}
