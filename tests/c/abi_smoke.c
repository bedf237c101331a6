/* abi_smoke.c -- test: include/bmq.h is plain C (what a JNI / cgo binding compiles against) and the host-side entry points
 * work from C without any C++ or Python in between.  Built with gcc -std=c99 and linked against libbmq.so by
 * tests/test_host.py; runs without a GPU (host-only engine, device = -1). */
#include <stdio.h>
#include <string.h>

#include "bmq.h"

#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            fprintf(stderr, "abi_smoke: %s failed (line %d)\n", #cond, __LINE__); \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main(void) {
    /* codec: KVSchemaUtil.toNormalRouteKey("tenantA", "a/+/#", "0\0inbox1\0d3") */
    const uint8_t recv[] = {'0', 0, 'i', 'n', 'b', 'o', 'x', '1', 0, 'd', '3'};
    uint8_t key[256];
    const uint32_t kl = bmq_route_key_encode((const uint8_t*)"tenantA", 7, (const uint8_t*)"a/+/#", 5, 1, recv, sizeof recv, key, sizeof key);
    CHECK(kl > 0 && kl <= sizeof key);
    uint32_t spans[6];
    CHECK(bmq_route_key_decode(key, kl, spans) == 1);
    CHECK(spans[1] == 7 && memcmp(key + spans[0], "tenantA", 7) == 0);
    CHECK(spans[3] == 5 && memcmp(key + spans[2], "a\0+\0#", 5) == 0); /* escaped filter: levels joined by NUL */
    CHECK(spans[5] == sizeof recv && memcmp(key + spans[4], recv, sizeof recv) == 0);
    CHECK(bmq_java_string_hash((const uint8_t*)"hello", 5) == 99162322); /* "hello".hashCode() */

    /* host-only engine: build, inspect, and the loud refusal to match without a device */
    bmq_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.device = -1;
    bmq_engine* e = NULL;
    CHECK(bmq_engine_create(&cfg, &e) == BMQ_OK && e != NULL);
    const uint32_t off[2] = {0, kl};
    CHECK(bmq_rebuild(e, key, off, 1) == BMQ_OK);
    bmq_index_info info;
    CHECK(bmq_index_info_get(e, &info) == BMQ_OK && info.n_routes == 1 && info.n_tenants == 1);
    uint8_t back[256];
    uint32_t bl = 0;
    CHECK(bmq_route_key(e, 0, back, sizeof back, &bl) == BMQ_OK && bl == kl && memcmp(back, key, kl) == 0);
    uint32_t ids[4], n = 0;
    CHECK(bmq_index_find(e, (const uint8_t*)"tenantA", 7, (const uint8_t*)"a/+/#", 5, ids, 4, &n) == BMQ_OK && n == 1 && ids[0] == 0);
    const uint32_t toff[2] = {0, 7}, poff[2] = {0, 5}, tt[1] = {0};
    uint32_t row[2];
    uint64_t need = 0;
    CHECK(bmq_match_batch(e, (const uint8_t*)"tenantA", toff, 1, tt, (const uint8_t*)"a/b/c", poff, 1, row, ids, 4, &need) == BMQ_E_NODEVICE);
    bmq_batcher* b = NULL;
    CHECK(bmq_batcher_create(e, NULL, &b) == BMQ_E_NODEVICE && b == NULL);
    bmq_engine_destroy(e);
    printf("abi_smoke ok (%s)\n", bmq_version());
    return 0;
}
