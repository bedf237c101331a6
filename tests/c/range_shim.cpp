// range_shim.cpp -- test tool, not product: the range-pruning function of bmq_range_core.h (the code k_range_lookup runs on the
// GPU) compiled for the host, so that tests/test_range.py can compare it with the oracle's restatement of
// TenantRangeLookupCache.lookup without a GPU.
#include <cstddef>

#include "../../bifromq_amd/csrc/bmq_range_core.h"

extern "C" void range_lookup_host(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics, const uint32_t* topic_off, uint32_t n_topics,
                                  const uint8_t* kind, const uint8_t* first, const uint32_t* first_off, const uint8_t* last, const uint32_t* last_off,
                                  uint32_t n_cand, uint8_t* keep) {
    for (uint32_t i = 0; i < n_topics; i++)
        bmq::range_lookup_one(tenant, tenant_len, topics, topic_off[i], topic_off[i + 1], kind, first, first_off, last, last_off, n_cand,
                              keep + (size_t)i * n_cand);
}
