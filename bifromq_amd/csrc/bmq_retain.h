// bmq_retain.h -- retain direction (index of retained topics, queried by wildcard filters).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "bmq_layout.h"

namespace bmq {
struct RetainCounters {
    unsigned long long total_ids;
    uint32_t status, pad;
};
struct RetainBatchArgs {
    uint32_t n_filters;
};
struct RetainIndexHost {};
struct RetainDevice {};
} // namespace bmq
struct bmq_engine;
static int retain_finish(bmq_engine* e, uint64_t* out_total);
