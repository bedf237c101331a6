// bmq_retain_args.h -- what the host and the kernels of one retain-direction batch share: level kinds, status bits and the argument
// block of k_retain_walk / k_retain_walk_v1 / k_retain_walk_deep.  Plain structs: included by bmq_retain_kernels.h (device) and by the
// wave emulator's harness of tools/emu/ (host).
#pragma once
#include <cstdint>

#include "bmq_batch_args.h"
#include "bmq_retain.h"

namespace bmq {

constexpr uint32_t RT_PLUS = 0xFFFFFFFDu, RT_HASH = 0xFFFFFFFCu; // level kinds next to dictionary tokens
constexpr uint32_t ST_RETAIN_DEEP = 128u, ST_RETAIN_FRONT = 256u;
constexpr uint32_t RW_LV = 16;            // levels of a filter k_retain_walk handles (MQTT ingress rejects more: Setting.MaxTopicLevels); deeper: k_retain_walk_deep
constexpr uint32_t ST_RETAIN_LIST = 512u; // k_retain_walk: a frontier list outgrew the wave's arena (the batch is re-run with a larger one)

struct RetainArgs {
    RetainIndexView ix;
    const uint8_t* tenants;
    const uint32_t* tenant_off;
    uint32_t n_tenants;
    const uint32_t* filter_tenant;
    const uint8_t* filters;
    const uint32_t* filter_off;
    uint32_t n_filters;
    uint2* gscratch;       // per wave: 2 * gcap ranges
    uint32_t gcap;
    // filters of more than R_MAXL levels (MQTT ingress rejects more than 16: Setting.MaxTopicLevels): the walk lists them (their count
    // is Counters.slow_count), a second launch of the same walk with its per-level arrays in global memory answers them
    uint32_t* deep_list;   // [n_filters]
    uint32_t* deep_levels; // deep pass only: per wave 6 arrays of deep_maxl + 1 words
    uint32_t deep_maxl;
    // k_retain_walk (bmq_rwalk_kernel.h): per persistent wave and slot two frontier lists of rw_cap entries (what does not fit the LDS part)
    uint32_t* rw_arena;
    uint32_t rw_cap;
};

// Topics added since the bulk load (the overlay trie) are matched by a kernel of their own, next to k_retain_walk, into a range list of
// their own (ids above every bulk-loaded id: a row is its bulk-loaded ids, then these): what the kernels behind the walks read beside
// BatchArgs' list.  All null: no such list.
struct RetainOvList {
    const uint32_t* pair_off;
    const uint32_t* pair_cnt;
    const uint32_t* route_cnt;
    const MatchRange* pairs;
};

} // namespace bmq
