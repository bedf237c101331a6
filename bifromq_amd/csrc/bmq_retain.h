// bmq_retain.h -- retain direction: an index of RETAINED TOPICS queried by wildcard FILTERS.
//
// Replaces RetainTopicIndex (RS/index/RetainTopicIndex.java:35-144), a TopicLevelTrie keyed [tenantId, levels...]
// walked recursively with the RetainMatcher branch selector (:36-124; UTIL/index/TopicLevelTrie.java:190-249).
//
// Layout (host builds, HBM holds):
//   * topic id = rank of (tenant, level list) in byte order, so every trie subtree is ONE contiguous id range:
//     a trailing '#' is answered with (begin, count) ranges instead of a subtree traversal;
//   * nodes are numbered breadth first with children sorted by label bytes, so the children of a contiguous node range
//     are again ONE contiguous node range: a '+' level maps a range of nodes to a range of nodes in O(1) (two reads),
//     and the '$'-prefixed children a first-level wildcard must skip are one contiguous run;
//   * literal levels use a bucketised hash table (parent node, token) -> child, four 16-byte entries per 64-byte line.
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

#include "bmq_layout.h"

namespace bmq {

constexpr uint32_t RN_TERM = 0x80000000u; // RNode.child_count flag: a retained topic ends at this node

struct alignas(16) RNode {
    uint32_t child_begin; // first child (children are consecutive node ids)
    uint32_t child_count; // | RN_TERM
    uint32_t sub_begin;   // topic ids of the subtree: [sub_begin, sub_end); the node's own topic (if any) is sub_begin
    uint32_t sub_end;
};
static_assert(sizeof(RNode) == 16, "RNode must be 16 bytes");

struct alignas(16) REdge { // hash entry: (parent node, token) -> child node; parent == NONE: empty
    uint32_t parent, token, child, pad;
};

struct alignas(32) RTenantSlot { // tenant directory entry (token == 0: empty)
    uint32_t token;
    uint32_t root;                 // node id of the tenant's root
    uint32_t sys_node_lo, sys_node_hi; // the root's '$'-prefixed children: node ids [lo, hi)
    uint32_t sys_id_lo, sys_id_hi;     // ... and the topic ids below them: [lo, hi)
    uint32_t pad[2];
};
static_assert(sizeof(RTenantSlot) == 32, "RTenantSlot must be 32 bytes");

struct RetainIndexView {
    const RNode* nodes;
    const REdge* edges;
    uint32_t edge_bucket_mask; // (entries / 4) - 1
    const RTenantSlot* tenants;
    uint32_t tenant_mask;
    const DictSlot* dict;
    uint32_t dict_group_mask;
    const uint8_t* pool;
};

struct RetainIndexHost {
    // sorted, de-duplicated (tenant, topic) set; topic id = position
    std::vector<uint8_t> bytes;      // tenant '\0' topic, concatenated
    std::vector<uint64_t> off{0};
    std::vector<uint32_t> tenant_len;
    std::vector<RNode> nodes;
    std::vector<REdge> edges;
    std::vector<RTenantSlot> tenants;
    std::vector<DictSlot> dict;
    std::vector<uint8_t> pool;
    uint64_t n_topics = 0, n_tenants = 0;
    std::string error;

    size_t size() const { return off.size() - 1; }
    std::string_view tenant_of(size_t i) const { return std::string_view((const char*)bytes.data() + off[i], tenant_len[i]); }
    std::string_view topic_of(size_t i) const {
        return std::string_view((const char*)bytes.data() + off[i] + tenant_len[i] + 1, (size_t)(off[i + 1] - off[i]) - tenant_len[i] - 1);
    }
    // replace the topic set; entries given as parallel (tenant, topic) string views
    void assign(std::vector<std::pair<std::string, std::string>>&& items);
    std::vector<std::pair<std::string, std::string>> items() const;
    bool build();
};

BMQ_HD uint32_t redge_bucket(uint32_t parent, uint32_t token, uint32_t mask) {
    uint32_t h = (parent ^ rotl32(token, 16)) * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h & mask;
}

} // namespace bmq
