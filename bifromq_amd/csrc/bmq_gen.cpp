// bmq_gen.cpp -- deterministic synthetic workloads of SURVEY.md 8d / BASELINE.md section 3 (bench + test tooling,
// not part of the match path).  PRNG = splitmix64.  Built into libbmq_gen.so.
//
//   vocabulary per level l: V = [8, 64, 512, 4096, 4096, 4096, 256, 64], token "l{l}_{rank}", rank ~ Zipf(1.1)
//   depth uniform in [3, 8]; 1 % get a "$sys" first level; 0.5 % contain one empty level
//   filters (mixed): 70 % literal, 15 % one '+', 5 % two '+', 8 % trailing '#', 2 % '+' ... '#'
//   routes: one normal route per filter (receiverUrl "0\0inbox{i}\0d{i%64}"), 1 % of filters get 32 receivers,
//           0.5 % are $share groups; a repeated filter string gets a fresh receiver (distinct key, same filter)
//   publishes: 90 % instantiate a stored filter (wildcards filled from the vocabulary), 10 % fresh random;
//              tenant of a publish ~ Zipf(1.0) over tenants
//   retain direction: retained topics are literal topics as above; query filters 50 % one '+', 30 % trailing '#'
//           (preceded by >= 2 non-'#' levels), 20 % both.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <chrono>
#include <atomic>

#include "bmq_codec.h"
#include "bmq_layout.h"

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

struct Zipf {
    std::vector<double> cdf;
    Zipf(uint32_t n, double s) : cdf(n) {
        double acc = 0;
        for (uint32_t i = 0; i < n; i++) {
            acc += 1.0 / std::pow((double)(i + 1), s);
            cdf[i] = acc;
        }
        for (auto& c : cdf) c /= acc;
    }
    uint32_t draw(Rng& r) const {
        const double u = r.uniform();
        return (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
    }
};

const uint32_t VOCAB[8] = {8, 64, 512, 4096, 4096, 4096, 256, 64};

struct Vocab {
    std::vector<Zipf> z;
    Vocab() {
        for (int l = 0; l < 8; l++) z.emplace_back(VOCAB[l], 1.1);
    }
    std::string token(int level, Rng& r) const {
        const int l = level < 8 ? level : 7;
        return "l" + std::to_string(l) + "_" + std::to_string(z[l].draw(r));
    }
};

using Levels = std::vector<std::string>;

Levels random_topic_levels(const Vocab& v, Rng& r) {
    const int depth = 3 + (int)r.below(6);
    Levels lv(depth);
    for (int i = 0; i < depth; i++) lv[i] = v.token(i, r);
    if (r.below(100) == 0) lv[0] = "$sys";
    if (r.below(200) == 0) lv[r.below((uint32_t)depth)] = "";
    return lv;
}

// mode 0: literal only; 1: dist mix; 2: retain query mix
Levels random_filter_levels(const Vocab& v, Rng& r, int mode) {
    Levels lv = random_topic_levels(v, r);
    if (mode == 0) return lv;
    const uint32_t p = r.below(100);
    auto plus = [&]() { lv[r.below((uint32_t)lv.size())] = "+"; };
    if (mode == 1) {
        if (p < 70) {
        } else if (p < 85) plus();
        else if (p < 90) { plus(); plus(); }
        else if (p < 98) lv.back() = "#";
        else { lv.back() = "#"; lv[r.below((uint32_t)lv.size() - 1)] = "+"; }
    } else {
        if (p < 50) plus();
        else if (p < 80) lv.back() = "#"; // depth >= 3: at least 2 levels before '#'
        else { lv.back() = "#"; lv[r.below((uint32_t)lv.size() - 1)] = "+"; }
        if (lv.size() >= 2 && lv[0] == "+" && lv[1] == "#") lv[0] = v.token(0, r);
        if (lv.size() == 3 && lv.back() == "#" && (lv[0] == "+" && lv[1] == "+")) lv[0] = v.token(0, r);
    }
    return lv;
}

std::string join(const Levels& lv) {
    std::string s;
    for (size_t i = 0; i < lv.size(); i++) {
        if (i) s.push_back('/');
        s += lv[i];
    }
    return s;
}

struct Packed {
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off{0};
    void push(const std::string& s) {
        bytes.insert(bytes.end(), s.begin(), s.end());
        off.push_back((uint32_t)bytes.size());
    }
    void pad() {
        const size_t n = bytes.size();
        bytes.resize(((n + 15) & ~(size_t)15) + 16, 0);
        (void)n;
    }
};

struct Gen {
    uint64_t seed;
    uint32_t tenant_base = 0;            // global index of this generator's first tenant (contiguous shard), or
    std::vector<uint32_t> tenant_ids;    // ... the explicit global tenant indices of a hash shard (ascending)
    uint32_t n_tenants, per_tenant;
    int mode;
    Vocab vocab;
    Packed tenants;                      // tenant ids "tenant%06u"
    Packed keys;                         // all route keys, sorted
    std::vector<uint32_t> tenant_first;  // first key rank per tenant (n_tenants + 1)
    Packed out_topics;                   // last generated batch
    std::vector<uint32_t> out_tenant;
};

std::string tenant_name(uint32_t t) {
    char b[32];
    snprintf(b, sizeof b, "tenant%06u", t);
    return b;
}

void gen_tenant_keys(const Gen& g, uint32_t t_local, std::vector<std::string>& out) {
    const uint32_t t = g.tenant_ids.empty() ? g.tenant_base + t_local : g.tenant_ids[t_local];
    Rng r(g.seed * 0x100000001B3ull + 0x517CC1B727220A95ull * (t + 1));
    const std::string tn = tenant_name(t);
    out.clear();
    out.reserve(g.per_tenant + 32);
    uint32_t rid = 0;
    while (out.size() < g.per_tenant) {
        const std::string f = join(random_filter_levels(g.vocab, r, g.mode));
        const uint32_t p = r.below(1000);
        if (p < 5) { // 0.5 %: shared subscription group
            const std::string grp = "g" + std::to_string(r.below(64));
            out.push_back(bmq::encode_route_key(tn, f, r.below(2) ? 2 : 3, grp));
        } else {
            const uint32_t n_recv = p < 15 ? 32u : 1u; // 1 %: 32 receivers on one filter
            for (uint32_t k = 0; k < n_recv && out.size() < g.per_tenant; k++, rid++) {
                std::string recv = "0";
                recv.push_back('\0');
                recv += "inbox" + std::to_string(rid);
                recv.push_back('\0');
                recv += "d" + std::to_string(rid % 64);
                out.push_back(bmq::encode_route_key(tn, f, 1, recv));
            }
        }
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
}

} // namespace

extern "C" {

static void* gen_build(Gen* g);

// tenants tenant_base .. tenant_base + n_tenants - 1
void* bmqgen_create(uint64_t seed, uint32_t tenant_base, uint32_t n_tenants, uint32_t routes_per_tenant, int mode) {
    Gen* g = new Gen();
    g->seed = seed;
    g->tenant_base = tenant_base;
    g->n_tenants = n_tenants;
    g->per_tenant = routes_per_tenant;
    g->mode = mode;
    return gen_build(g);
}

// an explicit, ascending list of global tenant indices (the tenants hash(tenantId) mod N assigns to one rank)
void* bmqgen_create_list(uint64_t seed, const uint32_t* tenant_ids, uint32_t n_tenants, uint32_t routes_per_tenant, int mode) {
    Gen* g = new Gen();
    g->seed = seed;
    g->tenant_ids.assign(tenant_ids, tenant_ids + n_tenants);
    g->n_tenants = n_tenants;
    g->per_tenant = routes_per_tenant;
    g->mode = mode;
    return gen_build(g);
}

static void* gen_build(Gen* g) {
    const uint32_t n_tenants = g->n_tenants;
    for (uint32_t t = 0; t < n_tenants; t++) g->tenants.push(tenant_name(g->tenant_ids.empty() ? g->tenant_base + t : g->tenant_ids[t]));
    g->tenants.pad();
    std::vector<std::vector<std::string>> per(n_tenants);
    unsigned hw = std::thread::hardware_concurrency();
    if (!hw) hw = 1;
    const unsigned nth = std::min<unsigned>(hw, n_tenants ? n_tenants : 1);
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nth; w++)
        th.emplace_back([&, w] {
            for (uint32_t t = w; t < n_tenants; t += nth) gen_tenant_keys(*g, t, per[t]);
        });
    for (auto& x : th) x.join();
    g->tenant_first.assign(n_tenants + 1, 0);
    uint64_t total_bytes = 0;
    for (uint32_t t = 0; t < n_tenants; t++)
        for (auto& k : per[t]) total_bytes += k.size();
    if (total_bytes >= 0xFFFFFFF0ull) { // uint32 offsets of the ABI
        delete g;
        return nullptr;
    }
    g->keys.bytes.reserve(total_bytes + 32);
    for (uint32_t t = 0; t < n_tenants; t++) { // fixed-width tenant ids: tenant order == key order
        g->tenant_first[t] = (uint32_t)(g->keys.off.size() - 1);
        for (auto& k : per[t]) g->keys.push(k);
        std::vector<std::string>().swap(per[t]);
    }
    g->tenant_first[n_tenants] = (uint32_t)(g->keys.off.size() - 1);
    return g;
}

void bmqgen_destroy(void* h) { delete (Gen*)h; }

uint32_t bmqgen_n_keys(void* h) { return (uint32_t)((Gen*)h)->keys.off.size() - 1; }
const uint8_t* bmqgen_key_bytes(void* h) { return ((Gen*)h)->keys.bytes.data(); }
const uint32_t* bmqgen_key_off(void* h) { return ((Gen*)h)->keys.off.data(); }
uint32_t bmqgen_n_tenants(void* h) { return ((Gen*)h)->n_tenants; }
const uint8_t* bmqgen_tenant_bytes(void* h) { return ((Gen*)h)->tenants.bytes.data(); }
const uint32_t* bmqgen_tenant_off(void* h) { return ((Gen*)h)->tenants.off.data(); }
const uint32_t* bmqgen_tenant_first(void* h) { return ((Gen*)h)->tenant_first.data(); }

// Generates a publish batch (kept inside the generator until the next call).  tenant_lo/hi restrict the tenants
// publishes are drawn for (a rank's shard); hit_permille = share of topics instantiated from a stored filter.
// grouped != 0: the batch is ordered by tenant (stable), the shape of a BatchDistRequest -- one DistPack per tenant
// holding that tenant's topics (DW/DistWorkerCoProc.java:522-539).
uint32_t bmqgen_topics(void* h, uint64_t seed, uint32_t n_topics, uint32_t tenant_lo, uint32_t tenant_hi, uint32_t hit_permille,
                       int grouped) {
    Gen& g = *(Gen*)h;
    Rng r(seed ^ 0xD1B54A32D192ED03ull);
    if (tenant_hi > g.n_tenants) tenant_hi = g.n_tenants;
    if (tenant_lo >= tenant_hi) return 0;
    Zipf zt(tenant_hi - tenant_lo, 1.0);
    g.out_topics = Packed();
    g.out_tenant.clear();
    g.out_tenant.reserve(n_topics);
    g.out_topics.bytes.reserve((size_t)n_topics * 44);
    for (uint32_t i = 0; i < n_topics; i++) {
        const uint32_t t = tenant_lo + zt.draw(r);
        const uint32_t k0 = g.tenant_first[t], k1 = g.tenant_first[t + 1];
        Levels lv;
        if (k1 > k0 && r.below(1000) < hit_permille) {
            const uint32_t k = k0 + r.below(k1 - k0);
            bmq::RouteKeyParts kp;
            const std::string_view key((const char*)g.keys.bytes.data() + g.keys.off[k], g.keys.off[k + 1] - g.keys.off[k]);
            bmq::decode_route_key(key, kp);
            size_t s = 0;
            const std::string_view f = kp.esc_filter;
            for (size_t j = 0; j <= f.size(); j++)
                if (j == f.size() || f[j] == '\0') {
                    lv.emplace_back(f.substr(s, j - s));
                    s = j + 1;
                }
            for (size_t j = 0; j < lv.size(); j++)
                if (lv[j] == "+") lv[j] = g.vocab.token((int)j, r);
            if (lv.back() == "#") {
                lv.pop_back();
                const uint32_t extra = r.below(3);
                for (uint32_t e = 0; e < extra; e++) lv.push_back(g.vocab.token((int)lv.size(), r));
                if (lv.empty()) lv.push_back(g.vocab.token(0, r));
            }
        } else {
            lv = random_topic_levels(g.vocab, r);
        }
        g.out_topics.push(join(lv));
        g.out_tenant.push_back(t);
    }
    if (grouped) {
        std::vector<uint32_t> order(n_topics);
        for (uint32_t i = 0; i < n_topics; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return g.out_tenant[a] < g.out_tenant[b]; });
        Packed np;
        np.bytes.reserve(g.out_topics.bytes.size() + 32);
        std::vector<uint32_t> nt(n_topics);
        for (uint32_t i = 0; i < n_topics; i++) {
            const uint32_t j = order[i];
            np.bytes.insert(np.bytes.end(), g.out_topics.bytes.begin() + g.out_topics.off[j],
                            g.out_topics.bytes.begin() + g.out_topics.off[j + 1]);
            np.off.push_back((uint32_t)np.bytes.size());
            nt[i] = g.out_tenant[j];
        }
        g.out_topics = std::move(np);
        g.out_tenant.swap(nt);
    }
    g.out_topics.pad();
    return n_topics;
}
const uint8_t* bmqgen_topic_bytes(void* h) { return ((Gen*)h)->out_topics.bytes.data(); }
const uint32_t* bmqgen_topic_off(void* h) { return ((Gen*)h)->out_topics.off.data(); }
const uint32_t* bmqgen_topic_tenant(void* h) { return ((Gen*)h)->out_tenant.data(); }

// Retain direction: n retained literal topics (unique per tenant) or n query filters (mode 2) into the same out buffers.
uint32_t bmqgen_retain(void* h, uint64_t seed, uint32_t n, int filters) {
    Gen& g = *(Gen*)h;
    Rng r(seed ^ 0xA24BAED4963EE407ull);
    g.out_topics = Packed();
    g.out_tenant.clear();
    Zipf zt(g.n_tenants ? g.n_tenants : 1, 1.0);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t t = g.n_tenants > 1 ? zt.draw(r) : 0;
        g.out_topics.push(join(filters ? random_filter_levels(g.vocab, r, 2) : random_topic_levels(g.vocab, r)));
        g.out_tenant.push_back(t);
    }
    g.out_topics.pad();
    return n;
}

// ---- load driver for the batching front (include/bmq.h, bmq_batcher_match_all): n_threads threads issue ONE-topic calls, the
// production call pattern of TenantRouteCache (DW/cache/TenantRouteCache.java:180-193).  `fn` is the address of
// bmq_batcher_match_all (passed in so that this library does not link against libbmq.so).  Per topic: the number of ids
// and an order-sensitive hash of the row, for the caller to compare with a direct bmq_match_batch over the same topics.
typedef int (*match_all_fn)(void*, const uint8_t*, uint32_t, const uint8_t*, const uint32_t*, uint32_t, uint32_t*, uint32_t*,
                            uint64_t, uint64_t*, uint64_t*);
uint64_t bmqgen_row_hash(const uint32_t* ids, uint64_t n) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (uint64_t i = 0; i < n; i++) h = (h ^ ids[i]) * 0x100000001B3ull;
    return h;
}
int bmqgen_drive_singletons(void* fn, void* batcher, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                            const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                            uint32_t n_threads, uint32_t* out_count, uint64_t* out_hash, double* out_seconds) {
    if (!fn || !batcher || !n_threads) return -1;
    const match_all_fn match = (match_all_fn)fn;
    std::atomic<uint32_t> next{0};
    std::atomic<int> err{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (uint32_t w = 0; w < n_threads; w++)
        th.emplace_back([&] {
            std::vector<uint32_t> ids(1024);
            for (;;) {
                const uint32_t i = next.fetch_add(1);
                if (i >= n_topics || err.load()) break;
                const uint32_t ti = topic_tenant[i];
                if (ti >= n_tenants) {
                    err = -1;
                    break;
                }
                const uint32_t off[2] = {0, topic_off[i + 1] - topic_off[i]};
                uint32_t row_ptr[2];
                uint64_t need = 0, epoch = 0;
                int rc = match(batcher, tenants + tenant_off[ti], tenant_off[ti + 1] - tenant_off[ti], topics + topic_off[i], off, 1,
                               row_ptr, ids.data(), ids.size(), &need, &epoch);
                if (rc == -3) { // BMQ_E_NOSPACE: grow and ask again
                    ids.resize(need + 64);
                    rc = match(batcher, tenants + tenant_off[ti], tenant_off[ti + 1] - tenant_off[ti], topics + topic_off[i], off, 1,
                               row_ptr, ids.data(), ids.size(), &need, &epoch);
                }
                if (rc) {
                    err = rc;
                    break;
                }
                out_count[i] = (uint32_t)need;
                out_hash[i] = bmqgen_row_hash(ids.data(), need);
            }
        });
    for (auto& t : th) t.join();
    if (out_seconds) *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return err.load();
}

// ---- the same through the route cache (bmq_route_cache_get): n_threads threads, every topic once, `passes` times over the batch (the
// second pass of a batch finds what the first one loaded)
typedef int (*cache_get_fn)(void*, const uint8_t*, uint32_t, const uint8_t*, uint32_t, uint64_t, uint32_t*, uint32_t, uint32_t*, uint64_t*);
int bmqgen_drive_cache(void* fn, void* cache, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                       const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, uint32_t n_threads, uint32_t passes, uint32_t* out_count,
                       uint64_t* out_hash, double* out_seconds) {
    if (!fn || !cache || !n_threads) return -1;
    const cache_get_fn get = (cache_get_fn)fn;
    std::atomic<int> err{0};
    for (uint32_t pass = 0; pass < passes && !err.load(); pass++) {
        std::atomic<uint32_t> next{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (uint32_t w = 0; w < n_threads; w++)
            th.emplace_back([&] {
                std::vector<uint32_t> ids(1024);
                for (;;) {
                    const uint32_t i = next.fetch_add(1);
                    if (i >= n_topics || err.load()) break;
                    const uint32_t ti = topic_tenant[i];
                    if (ti >= n_tenants) {
                        err = -1;
                        break;
                    }
                    uint32_t n = 0;
                    uint64_t epoch = 0;
                    int rc;
                    for (;;) {
                        rc = get(cache, tenants + tenant_off[ti], tenant_off[ti + 1] - tenant_off[ti], topics + topic_off[i], topic_off[i + 1] - topic_off[i],
                                 1000, ids.data(), (uint32_t)ids.size(), &n, &epoch);
                        if (rc != -3) break; // BMQ_E_NOSPACE: grow and ask again
                        ids.resize((size_t)n + 64);
                    }
                    if (rc) {
                        err = rc;
                        break;
                    }
                    out_count[i] = n;
                    out_hash[i] = bmqgen_row_hash(ids.data(), n);
                }
            });
        for (auto& t : th) t.join();
        if (out_seconds) out_seconds[pass] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return err.load();
}

// ---- the same for the asynchronous side (bmq_batcher_submit): n_threads threads submit every topic once, as fast as the batcher
// takes them; the callback records count + row hash; returns when every callback has run.
typedef void (*batcher_cb)(void*, int, const uint32_t*, uint32_t, uint64_t);
typedef int (*submit_fn)(void*, const uint8_t*, uint32_t, const uint8_t*, uint32_t, batcher_cb, void*);
struct AsyncCtx {
    uint32_t* out_count;
    uint64_t* out_hash;
    std::atomic<uint32_t> done{0};
    std::atomic<int> err{0};
};
struct AsyncReq {
    AsyncCtx* ctx;
    uint32_t i;
};
static void drive_cb(void* user, int status, const uint32_t* ids, uint32_t n, uint64_t) {
    AsyncReq* r = (AsyncReq*)user;
    if (status) r->ctx->err = status;
    else {
        r->ctx->out_count[r->i] = n;
        r->ctx->out_hash[r->i] = bmqgen_row_hash(ids, n);
    }
    r->ctx->done.fetch_add(1);
}
int bmqgen_drive_async(void* fn, void* batcher, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                       const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                       uint32_t n_threads, uint32_t* out_count, uint64_t* out_hash, double* out_seconds) {
    if (!fn || !batcher || !n_threads) return -1;
    const submit_fn submit = (submit_fn)fn;
    AsyncCtx ctx;
    ctx.out_count = out_count;
    ctx.out_hash = out_hash;
    std::vector<AsyncReq> reqs(n_topics);
    std::atomic<uint32_t> submitted{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (uint32_t w = 0; w < n_threads; w++)
        th.emplace_back([&, w] {
            for (uint32_t i = w; i < n_topics && !ctx.err.load(); i += n_threads) {
                const uint32_t ti = topic_tenant[i];
                reqs[i] = AsyncReq{&ctx, i};
                const int rc = ti < n_tenants ? submit(batcher, tenants + tenant_off[ti], tenant_off[ti + 1] - tenant_off[ti],
                                                        topics + topic_off[i], topic_off[i + 1] - topic_off[i], drive_cb, &reqs[i])
                                              : -1;
                if (rc) {
                    ctx.err = rc;
                    break;
                }
                submitted.fetch_add(1);
            }
        });
    for (auto& t : th) t.join();
    while (ctx.done.load() < submitted.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (out_seconds) *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return ctx.err.load();
}

// ---- and through the route cache's future-shaped call (bmq_route_cache_get_async): n_threads threads issue every topic once per pass
// without ever blocking on the GPU; hits complete inline, misses from the batcher's dispatcher thread
typedef int (*cache_get_async_fn)(void*, const uint8_t*, uint32_t, const uint8_t*, uint32_t, uint64_t, batcher_cb, void*);
int bmqgen_drive_cache_async(void* fn, void* cache, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                             const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, uint32_t n_threads, uint32_t passes,
                             uint32_t* out_count, uint64_t* out_hash, double* out_seconds) {
    if (!fn || !cache || !n_threads) return -1;
    const cache_get_async_fn get = (cache_get_async_fn)fn;
    std::vector<AsyncReq> reqs(n_topics);
    for (uint32_t pass = 0; pass < passes; pass++) {
        AsyncCtx ctx;
        ctx.out_count = out_count;
        ctx.out_hash = out_hash;
        std::atomic<uint32_t> submitted{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (uint32_t w = 0; w < n_threads; w++)
            th.emplace_back([&, w] {
                for (uint32_t i = w; i < n_topics && !ctx.err.load(); i += n_threads) {
                    const uint32_t ti = topic_tenant[i];
                    reqs[i] = AsyncReq{&ctx, i};
                    const int rc = ti < n_tenants ? get(cache, tenants + tenant_off[ti], tenant_off[ti + 1] - tenant_off[ti], topics + topic_off[i],
                                                        topic_off[i + 1] - topic_off[i], 1000, drive_cb, &reqs[i])
                                                  : -1;
                    if (rc) {
                        ctx.err = rc;
                        break;
                    }
                    submitted.fetch_add(1);
                }
            });
        for (auto& t : th) t.join();
        while (ctx.done.load() < submitted.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (out_seconds) out_seconds[pass] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (ctx.err.load()) return ctx.err.load();
    }
    return 0;
}

} // extern "C"
