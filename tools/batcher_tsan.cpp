// batcher_tsan.cpp -- test tool, not product: the concurrency of the batching front (bifromq_amd/csrc/bmq_batcher.inc) under
// ThreadSanitizer, without a GPU.  The engine is replaced by a stand-in whose "match" is a pure function of (tenant, topic), so
// every caller can check that it received exactly its own rows, whatever batch they travelled in:
//   * many threads in bmq_batcher_match_all (single topics and small sets, some with too small output buffers),
//   * threads in bmq_batcher_submit with callbacks, back-pressure (tiny max_batch_topics),
//   * an "apply" thread bumping the epoch under the engine lock, as bmq_routes_apply does,
//   * bmq_batcher_destroy while submitted requests are still waiting (they must all be called back).
// Build + run: make -C bifromq_amd/csrc tsan   (tests/test_host.py runs it)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "../include/bmq.h"

struct bmq_engine { // the members the batching front touches
    std::recursive_mutex api;
    uint64_t epoch = 1;
    int device = 0;
};

static uint32_t fake_count(std::string_view tenant, std::string_view topic) { return (uint32_t)((tenant.size() * 7 + topic.size() * 3) % 6); }
static uint32_t fake_id(std::string_view tenant, std::string_view topic, uint32_t k) {
    uint32_t h = 2166136261u;
    for (char c : tenant) h = (h ^ (uint8_t)c) * 16777619u;
    for (char c : topic) h = (h ^ (uint8_t)c) * 16777619u;
    return h % 1000003u + k;
}
static std::atomic<uint64_t> g_batches{0};
static std::atomic<int> g_launch_us{30}; // what a launch costs: slept away in the sanitizer runs, busy-waited in perf mode
static std::atomic<bool> g_busy_wait{false};

extern "C" int bmq_match_batch(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants,
                               const uint32_t* topic_tenant, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                               uint32_t* out_row_ptr, uint32_t* out_route_ids, uint64_t out_capacity, uint64_t* out_needed) {
    std::unique_lock<std::recursive_mutex> lk(e->api);
    g_batches++;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_topics; i++) {
        if (topic_tenant[i] >= n_tenants) return BMQ_E_INVAL;
        const std::string_view tn((const char*)tenants + tenant_off[topic_tenant[i]], tenant_off[topic_tenant[i] + 1] - tenant_off[topic_tenant[i]]);
        const std::string_view tp((const char*)topics + topic_off[i], topic_off[i + 1] - topic_off[i]);
        out_row_ptr[i] = (uint32_t)total;
        total += fake_count(tn, tp);
    }
    out_row_ptr[n_topics] = (uint32_t)total;
    *out_needed = total;
    if (total > out_capacity) return BMQ_E_NOSPACE;
    if (g_busy_wait.load()) { // perf mode: the GPU takes this long, the calling thread waits for it (hipStreamSynchronize)
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(g_launch_us.load());
        while (std::chrono::steady_clock::now() < until) std::this_thread::yield();
    } else std::this_thread::sleep_for(std::chrono::microseconds(g_launch_us.load())); // a launch takes a while: requests pile up meanwhile
    for (uint32_t i = 0; i < n_topics; i++) {
        const std::string_view tn((const char*)tenants + tenant_off[topic_tenant[i]], tenant_off[topic_tenant[i] + 1] - tenant_off[topic_tenant[i]]);
        const std::string_view tp((const char*)topics + topic_off[i], topic_off[i + 1] - topic_off[i]);
        for (uint32_t k = 0; k < out_row_ptr[i + 1] - out_row_ptr[i]; k++) out_route_ids[out_row_ptr[i] + k] = fake_id(tn, tp, k);
    }
    return BMQ_OK;
}

// bmq_match_submit_dev / bmq_match_wait_dev of the stand-in: three tickets over the caller's buffers; a launch is ready launch_us after its
// submit, except that the "GPU" works on one launch at a time for 40% of that (the kernels; the rest overlaps with the neighbours)
struct FakeTicket {
    bool used = false;
    const uint8_t *tenants, *topics;
    const uint32_t *tenant_off, *topic_tenant, *topic_off;
    uint32_t n_tenants, n_topics;
    uint32_t *row, *ids;
    uint64_t cap;
    std::chrono::steady_clock::time_point ready;
};
static FakeTicket g_tk[BMQ_MAX_TICKETS];
static std::mutex g_tk_mu;
static std::chrono::steady_clock::time_point g_gpu_free;
static std::atomic<int> g_ticket_refusals{0}; // test knob: the next n submits are refused (BMQ_E_STATE), as when other users hold every ticket

extern "C" void* bmq_host_alloc(size_t n) { return malloc(n); }
extern "C" void bmq_host_free(void* p) { free(p); }
extern "C" int bmq_match_submit_dev(bmq_engine* e, const uint8_t* tenants, const uint32_t* tenant_off, uint32_t n_tenants, const uint32_t* topic_tenant,
                                    const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics, uint32_t* row, uint32_t* ids, uint64_t cap,
                                    uint64_t* total, int* out_ticket) {
    std::unique_lock<std::recursive_mutex> lk(e->api);
    if (((uintptr_t)topics & 15) || !total) return BMQ_E_INVAL;
    if (g_ticket_refusals.load() > 0 && g_ticket_refusals.fetch_sub(1) > 0) return BMQ_E_STATE;
    std::lock_guard<std::mutex> g(g_tk_mu);
    for (int k = 0; k < BMQ_MAX_TICKETS; k++) {
        FakeTicket& t = g_tk[k];
        if (t.used) continue;
        t.used = true;
        t.tenants = tenants, t.tenant_off = tenant_off, t.n_tenants = n_tenants, t.topic_tenant = topic_tenant, t.topics = topics, t.topic_off = topic_off;
        t.n_topics = n_topics, t.row = row, t.ids = ids, t.cap = cap;
        const auto now = std::chrono::steady_clock::now();
        const auto L = std::chrono::microseconds(g_launch_us.load());
        t.ready = std::max(now + L * 6 / 10, g_gpu_free) + L * 4 / 10;
        g_gpu_free = t.ready;
        *out_ticket = k;
        return BMQ_OK;
    }
    return BMQ_E_STATE;
}
extern "C" int bmq_match_wait_dev(bmq_engine*, int ticket, uint64_t* out_total) {
    FakeTicket* t;
    {
        std::lock_guard<std::mutex> g(g_tk_mu);
        t = &g_tk[ticket];
        if (!t->used) return BMQ_E_STATE;
    }
    if (g_busy_wait.load())
        while (std::chrono::steady_clock::now() < t->ready) std::this_thread::yield();
    else std::this_thread::sleep_until(t->ready);
    g_batches++;
    uint64_t total = 0;
    int rc = BMQ_OK;
    auto tenant_of = [&](uint32_t i) { return std::string_view((const char*)t->tenants + t->tenant_off[t->topic_tenant[i]], t->tenant_off[t->topic_tenant[i] + 1] - t->tenant_off[t->topic_tenant[i]]); };
    auto topic_of = [&](uint32_t i) { return std::string_view((const char*)t->topics + t->topic_off[i], t->topic_off[i + 1] - t->topic_off[i]); };
    for (uint32_t i = 0; i < t->n_topics && rc == BMQ_OK; i++) {
        if (t->topic_tenant[i] >= t->n_tenants) rc = BMQ_E_INVAL;
        else {
            t->row[i] = (uint32_t)total;
            total += fake_count(tenant_of(i), topic_of(i));
        }
    }
    if (rc == BMQ_OK) {
        t->row[t->n_topics] = (uint32_t)total;
        if (out_total) *out_total = total;
        if (total > t->cap) rc = BMQ_E_NOSPACE;
    }
    for (uint32_t i = 0; i < t->n_topics && rc == BMQ_OK; i++)
        for (uint32_t k = 0; k < t->row[i + 1] - t->row[i]; k++) t->ids[t->row[i] + k] = fake_id(tenant_of(i), topic_of(i), k);
    std::lock_guard<std::mutex> g(g_tk_mu);
    t->used = false;
    return rc;
}

#define BMQ_BATCHER_INITIAL_IDS 4
static const char* bmq_env(const char* name) { return getenv(name); } // (the library's: bmq_layout.h -- environment switches are experiment builds' only)
#include "../bifromq_amd/csrc/bmq_batcher.inc"

static std::atomic<int> g_fail{0};
#define EXPECT(c)                                                     \
    do {                                                              \
        if (!(c)) {                                                   \
            fprintf(stderr, "batcher_tsan: %s (line %d)\n", #c, __LINE__); \
            g_fail++;                                                 \
        }                                                             \
    } while (0)

struct CbCtx {
    std::string tenant, topic;
    std::atomic<int>* done;
};
static void on_done(void* user, int status, const uint32_t* ids, uint32_t n, uint64_t epoch) {
    CbCtx* c = (CbCtx*)user;
    EXPECT(status == BMQ_OK && epoch >= 1);
    EXPECT(n == fake_count(c->tenant, c->topic));
    for (uint32_t k = 0; k < n && status == BMQ_OK; k++) EXPECT(ids[k] == fake_id(c->tenant, c->topic, k));
    c->done->fetch_add(1);
}

// batcher_tsan perf <threads> <calls per thread> <launch us>: blocking single-topic calls per second over the stand-in (not a test)
static int perf(int n_threads, int calls, int launch_us) {
    bmq_engine eng;
    g_launch_us = launch_us;
    g_busy_wait = getenv("BMQ_PERF_SLEEP") == nullptr;
    bmq_batcher* b = nullptr;
    bmq_batcher_create(&eng, nullptr, &b);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int w = 0; w < n_threads; w++)
        th.emplace_back([&, w] {
            std::mt19937 rng(w);
            uint32_t row[2], ids[16];
            for (int it = 0; it < calls; it++) {
                const std::string tp = "a/" + std::to_string(rng() % 100000);
                const uint32_t off[2] = {0, (uint32_t)tp.size()};
                uint64_t need = 0, epoch = 0;
                bmq_batcher_match_all(b, (const uint8_t*)"tenant", 6, (const uint8_t*)tp.data(), off, 1, row, ids, 16, &need, &epoch);
            }
        });
    for (auto& t : th) t.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    bmq_batcher_stats st;
    bmq_batcher_stats_get(b, &st);
    printf("perf: %d threads x %d calls, launch %d us: %.0f calls/s, %llu launches, %.1f topics/launch, %.1f us per launch cycle\n", n_threads, calls, launch_us,
           n_threads * (double)calls / sec, (unsigned long long)st.n_batches, (double)st.n_topics / st.n_batches, sec * 1e6 / st.n_batches);
    bmq_batcher_destroy(b);
    return 0;
}

// batcher_tsan perf_async <threads> <calls per thread> <launch us>: bmq_batcher_submit from a few threads, callbacks counted
static void count_cb(void* user, int, const uint32_t*, uint32_t, uint64_t) { ((std::atomic<uint64_t>*)user)->fetch_add(1, std::memory_order_relaxed); }
static int perf_async(int n_threads, int calls, int launch_us) {
    bmq_engine eng;
    g_launch_us = launch_us;
    g_busy_wait = getenv("BMQ_PERF_SLEEP") == nullptr;
    bmq_batcher* b = nullptr;
    bmq_batcher_create(&eng, nullptr, &b);
    std::atomic<uint64_t> done{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int w = 0; w < n_threads; w++)
        th.emplace_back([&, w] {
            std::mt19937 rng(w);
            char tp[32];
            for (int it = 0; it < calls; it++) {
                const int n = snprintf(tp, sizeof tp, "l0_%u/l1_%u/l2_%u", (unsigned)(rng() % 8), (unsigned)(rng() % 64), (unsigned)(rng() % 4096));
                bmq_batcher_submit(b, (const uint8_t*)"tenant000017", 12, (const uint8_t*)tp, (uint32_t)n, count_cb, &done);
            }
        });
    for (auto& t : th) t.join();
    bmq_batcher_stats st;
    bmq_batcher_destroy(b); // drains
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("perf_async: %d threads x %d submits, launch %d us: %.0f calls/s (%llu callbacks)\n", n_threads, calls, launch_us, n_threads * (double)calls / sec,
           (unsigned long long)done.load());
    (void)st;
    return done.load() == (uint64_t)n_threads * calls ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "perf_async")
        return perf_async(argc > 2 ? atoi(argv[2]) : 4, argc > 3 ? atoi(argv[3]) : 200000, argc > 4 ? atoi(argv[4]) : 100);
    if (argc > 1 && std::string(argv[1]) == "perf") return perf(argc > 2 ? atoi(argv[2]) : 64, argc > 3 ? atoi(argv[3]) : 2000, argc > 4 ? atoi(argv[4]) : 100);
    bmq_engine eng;
    const char* tenants[] = {"t", "tenantB", "x-long-tenant"};
    for (int pass = 0; pass < 4; pass++) {
        // pass 3: every 5th joiner loses the CPU for 2 ms between its add on the join word and its slot write (the test hook of
        // bmq_batcher.inc) -- many generations pass meanwhile; it must find itself left out, check out and join a later launch.  With the
        // round-3 recycling rule (only the launch's requests counted) such a joiner woke up inside a later use of its generation.
        g_batcher_stall_every = pass == 3 ? 5 : 0;
        g_batcher_stall_us = 2000;
        bmq_batcher_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.max_batch_topics = pass == 0 || pass == 3 ? 0 : (pass == 1 ? 5 : 64);
        bmq_batcher* b = nullptr;
        EXPECT(bmq_batcher_create(&eng, &cfg, &b) == BMQ_OK && b);
        std::atomic<bool> stop{false};
        std::thread mutator([&] { // bmq_routes_apply: takes the engine lock, bumps the epoch
            while (!stop.load()) {
                {
                    std::unique_lock<std::recursive_mutex> lk(eng.api);
                    eng.epoch++;
                }
                if (pass == 2) g_ticket_refusals = 3; // other users of the engine hold every ticket now and then: the front falls back to bmq_match_batch
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        });
        std::vector<std::thread> th;
        for (int w = 0; w < 24; w++)
            th.emplace_back([&, w] {
                std::mt19937 rng(1000 * pass + w);
                for (int it = 0; it < 150; it++) {
                    const std::string tn = tenants[rng() % 3];
                    const uint32_t nt = 1 + (rng() % 8 == 0 ? rng() % 6 : 0);
                    std::string bytes;
                    std::vector<uint32_t> off{0};
                    std::vector<std::string> tps;
                    for (uint32_t i = 0; i < nt; i++) {
                        std::string tp = "a/" + std::to_string(rng() % 50) + std::string(rng() % 7, 'z');
                        bytes += tp;
                        off.push_back((uint32_t)bytes.size());
                        tps.push_back(tp);
                    }
                    std::vector<uint32_t> row(nt + 1), ids(rng() % 5 == 0 ? 1 : 64);
                    uint64_t need = 0, epoch = 0;
                    int rc = bmq_batcher_match_all(b, (const uint8_t*)tn.data(), (uint32_t)tn.size(), (const uint8_t*)bytes.data(), off.data(), nt, row.data(),
                                                   ids.data(), ids.size(), &need, &epoch);
                    uint64_t want = 0;
                    for (auto& tp : tps) want += fake_count(tn, tp);
                    EXPECT(need == want && epoch >= 1);
                    if (want > ids.size()) EXPECT(rc == BMQ_E_NOSPACE);
                    else {
                        EXPECT(rc == BMQ_OK);
                        for (uint32_t i = 0; i < nt && rc == BMQ_OK; i++) {
                            EXPECT(row[i + 1] - row[i] == fake_count(tn, tps[i]));
                            for (uint32_t k = 0; k < row[i + 1] - row[i]; k++) EXPECT(ids[row[i] + k] == fake_id(tn, tps[i], k));
                        }
                    }
                }
            });
        th.emplace_back([&] { // whole multi-tenant batches as launches of their own (bmq_batcher_match_batch), in turn with the collected ones
            std::mt19937 rng(4242 + pass);
            for (int it = 0; it < 60; it++) {
                const uint32_t n = 1 + rng() % 20;
                std::string tn_bytes = "ttenantBx-long-tenant";
                const uint32_t tenant_off[4] = {0, 1, 8, 21};
                std::string bytes;
                std::vector<uint32_t> off{0}, tt(n);
                std::vector<std::string> tps(n);
                for (uint32_t i = 0; i < n; i++) {
                    tps[i] = "b/" + std::to_string(rng() % 50);
                    tt[i] = rng() % 3;
                    bytes += tps[i];
                    off.push_back((uint32_t)bytes.size());
                }
                std::vector<uint32_t> row(n + 1), ids(200);
                uint64_t need = 0, epoch = 0;
                const int rc = bmq_batcher_match_batch(b, (const uint8_t*)tn_bytes.data(), tenant_off, 3, tt.data(), (const uint8_t*)bytes.data(), off.data(), n,
                                                       row.data(), ids.data(), ids.size(), &need, &epoch);
                EXPECT(rc == BMQ_OK && epoch >= 1);
                for (uint32_t i = 0; i < n && rc == BMQ_OK; i++) {
                    const std::string tn = tenants[tt[i]];
                    EXPECT(row[i + 1] - row[i] == fake_count(tn, tps[i]));
                    for (uint32_t k = 0; k < row[i + 1] - row[i]; k++) EXPECT(ids[row[i] + k] == fake_id(tn, tps[i], k));
                }
            }
        });
        std::atomic<int> done{0};
        std::vector<std::unique_ptr<CbCtx>> ctxs;
        std::mutex cm;
        std::atomic<int> submitted{0};
        for (int w = 0; w < 4; w++)
            th.emplace_back([&, w] {
                std::mt19937 rng(77 * pass + w);
                for (int it = 0; it < 400; it++) {
                    auto c = std::make_unique<CbCtx>();
                    c->tenant = tenants[rng() % 3];
                    c->topic = "s/" + std::to_string(rng() % 1000);
                    c->done = &done;
                    CbCtx* raw = c.get();
                    {
                        std::lock_guard<std::mutex> g(cm);
                        ctxs.push_back(std::move(c));
                    }
                    EXPECT(bmq_batcher_submit(b, (const uint8_t*)raw->tenant.data(), (uint32_t)raw->tenant.size(), (const uint8_t*)raw->topic.data(),
                                              (uint32_t)raw->topic.size(), on_done, raw) == BMQ_OK);
                    submitted++;
                }
            });
        for (auto& t : th) t.join();
        stop = true;
        mutator.join();
        bmq_batcher_stats st;
        EXPECT(bmq_batcher_stats_get(b, &st) == BMQ_OK && st.n_requests >= 24 * 150);
        if (cfg.max_batch_topics) EXPECT(st.max_batch_topics <= std::max<uint64_t>(cfg.max_batch_topics, 20)); // a request larger than the bound runs alone; bmq_batcher_match_batch launches are not collected at all
        bmq_batcher_destroy(b); // drains the asynchronous side: every submitted request has been called back when it returns
        EXPECT(done.load() == submitted.load() && submitted.load() == 1600);
    }
    EXPECT(g_batcher_left_out.load() > 20); // the stalled joiners of pass 3 really took the left-out path
    if (g_fail.load()) {
        fprintf(stderr, "batcher_tsan: %d failures\n", g_fail.load());
        return 1;
    }
    printf("batcher_tsan ok: %llu fake launches, %llu joiners left out and re-joined\n", (unsigned long long)g_batches.load(),
           (unsigned long long)g_batcher_left_out.load());
    return 0;
}
