// bmq_retain.h -- retain direction: an index of RETAINED TOPICS queried by wildcard FILTERS.
//
// Replaces RetainTopicIndex (RS/index/RetainTopicIndex.java:35-144), a TopicLevelTrie keyed [tenantId, levels...]
// walked recursively with the RetainMatcher branch selector (:36-124; UTIL/index/TopicLevelTrie.java:190-249).
//
// Layout of a BULK LOAD (host builds, HBM holds; immutable until the next one -- add / remove in between mutate the overlay of
// bmq_retain_core.h on the device), per tenant:
//   * topic id = id_base(tenant) + rank of the topic's level list in byte order, so every trie subtree is ONE contiguous
//     id range: a trailing '#' is answered with (begin, count) ranges instead of a subtree traversal;
//   * a tenant's nodes are numbered breadth first (local ids, root = 0) with children sorted by label bytes, so the
//     children of a contiguous node range are again ONE contiguous node range: a '+' level maps a range of nodes to a
//     range of nodes in O(1) (two reads), and the '$'-prefixed children a first-level wildcard must skip are one run;
//   * literal levels use the tenant's bucketised edge hash (parent node, token) -> child, four 16-byte entries per
//     64-byte line.
#pragma once
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "bmq_dict.h"
#include "bmq_layout.h"

namespace bmq {

constexpr uint32_t RN_TERM = 0x80000000u; // RNode.child_count flag: a retained topic ends at this node

struct alignas(16) RNode {
    uint32_t child_begin; // first child (children are consecutive tenant-local node ids)
    uint32_t child_count; // | RN_TERM
    uint32_t sub_begin;   // tenant-local topic ranks of the subtree: [sub_begin, sub_end); the node's own topic is sub_begin
    uint32_t sub_end;
};
static_assert(sizeof(RNode) == 16, "RNode must be 16 bytes");

constexpr uint32_t RE_OVERFLOW = 0x80000000u; // REdge.child flag, FIRST entry of a bucket only: an edge whose home bucket this is lies in a later bucket
struct alignas(16) REdge { // hash entry: (parent node, token) -> child node (tenant-local ids); parent == NONE: empty
    uint32_t parent, token;
    uint32_t child; // | RE_OVERFLOW in a bucket's first entry: a look-up that misses in its home bucket goes on only then
    uint32_t child_topic; // the child's sub_begin | RN_TERM if a retained topic ends at the child: a filter's last literal level needs no node read
};

// Grandchild postings (round 5).  A literal level behind a '+' asks, for every child of ONE node g, for its child labelled t: hundreds of
// look-ups that mostly miss (56 % of all node visits of the survey's retain workload).  For a node g with at least RPOST_MIN children the
// index holds the answer ready: the edges (parent, t, child) of all grandchildren of g, ordered by (t, parent) -- `posts` --, and a hash
// (g, t) -> their slice.  parent order is frontier order, so the slice IS the next frontier, and an entry carries what a filter's last
// level needs (child_topic).
constexpr uint32_t RPOST_MIN = 32;
struct alignas(16) RGp { // hash entry: (grandparent node, token) -> posts[begin, begin + count) of the tenant; gp == NONE: empty
    uint32_t gp, token, begin;
    uint32_t count; // | RE_OVERFLOW in a bucket's first entry (see REdge.child)
};
BMQ_HD uint32_t rgp_bucket(uint32_t gp, uint32_t token, uint32_t mask) {
    uint32_t h = (gp ^ rotl32(token, 16)) * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h & mask;
}

struct alignas(64) RTenantSlot { // tenant directory entry (token == 0: empty)
    uint32_t token;
    uint32_t node_base;                // first node of the tenant in the global node array (its root)
    uint32_t edge_base;                // first entry of the tenant's edge region
    uint32_t edge_bucket_mask;         // (entries of the region / 4) - 1
    uint32_t id_base;                  // global id of the tenant's first topic
    uint32_t sys_node_lo, sys_node_hi; // the root's '$'-prefixed children: local node ids [lo, hi)
    uint32_t sys_id_lo, sys_id_hi;     // ... and the local topic ranks below them: [lo, hi)
    uint32_t post_base;                // first entry of the tenant's grandchild postings
    uint32_t gp_base, gp_bucket_mask;  // its (grandparent, token) hash: first entry, (entries / 4) - 1
    uint32_t pad[4];
};
static_assert(sizeof(RTenantSlot) == 64, "RTenantSlot must be 64 bytes");

constexpr uint32_t ON_SYS = 0x80000000u; // ONode.str_len flag: the label starts with '$'
struct alignas(32) ONode {               // one overlay node = one level of a topic added since the last bulk load
    uint32_t parent;                     // overlay node of the parent level (the root, node 0, has parent NONE; its children are tenants)
    uint32_t h1, h2;                     // LevelHash of the label
    uint32_t str_off;                    // the label's bytes in the overlay string pool
    uint32_t str_len;                    // | ON_SYS
    uint32_t first_child, next_sibling;  // child list (NONE ends it): what '+' and '#' enumerate
    uint32_t topic_id;                   // id of the topic that ends here (NONE: none ever did); whether it is retained NOW says the dead bit
};
static_assert(sizeof(ONode) == 32, "ONode must be 32 bytes");

// What the match kernels read on top of the bulk-loaded index.
struct RetainDynView {
    const ONode* onodes;
    const uint32_t* oedges; // node index, 0 = empty (the root is nobody's child)
    uint32_t oedge_mask;
    const uint8_t* opool;
    const unsigned long long* dead_bits; // bit id: the id is not retained now (removed, or never handed out)
    const uint32_t* dead_rank;           // dead ids in front of word w
    uint32_t base_n;                     // ids below are bulk-loaded ranks, ids from here on belong to overlay topics
    uint32_t use_dead;                   // some bulk-loaded id is dead: matched ranges are counted / expanded through the bitmap
    uint32_t ov_live;                    // overlay topics handed out so far (0: the overlay walk is skipped)
};

struct RetainIndexView {
    const RNode* nodes;
    const REdge* edges;
    const REdge* posts; // grandchild postings (see RGp)
    const RGp* gps;
    const RTenantSlot* tenants;
    uint32_t tenant_mask;
    const DictSlot* dict;
    uint32_t dict_group_mask;
    const uint8_t* pool;
    const unsigned long long* expire_at; // per topic id: the millisecond the retained message expires at (~0: never; 0: removed)
    RetainDynView dyn;                   // what changed since the bulk load (bmq_retain_core.h)
};

constexpr uint64_t RETAIN_NEVER = ~0ull;
// expireAt of RS/RetainStoreCoProc.java:298-304: physical part of the HLC timestamp (base-hlc HLC.java:145-151: hlc >>> 16, in ms)
// plus the expiry interval in seconds
BMQ_HD uint64_t retain_expire_at(uint64_t timestamp_hlc, uint32_t expiry_seconds) { return (timestamp_hlc >> 16) + (uint64_t)expiry_seconds * 1000ull; }

struct RTenantState {
    std::string name;
    std::vector<std::string> topics; // sorted (level-list byte order), unique
    std::vector<uint64_t> ts;        // per topic: HLC timestamp given to IRetainTopicIndex.add (0 if none)
    std::vector<uint32_t> expiry;    // per topic: expirySeconds (0xFFFFFFFF with ts 0: never expires)
    uint32_t token = 0;
    std::vector<RNode> nodes;        // breadth-first, local ids
    std::vector<REdge> edges;        // the tenant's edge region (size = 4 * buckets, power of two)
    std::vector<REdge> posts;        // grandchild postings, ordered by (grandparent, token, parent)
    std::vector<RGp> gps;            // (grandparent, token) -> slice of posts (size = 4 * buckets, power of two)
    uint32_t node_base = 0, node_cap = 0, edge_base = 0, edge_cap = 0, id_base = 0;
    uint32_t post_base = 0, post_cap = 0, gp_base = 0, gp_cap = 0;
    uint32_t sys_node_lo = 0, sys_node_hi = 0, sys_id_lo = 0, sys_id_hi = 0;
};

struct RetainIndexHost {
    // ---- image of the device arrays ----
    std::vector<RNode> nodes;
    std::vector<REdge> edges;
    std::vector<REdge> posts;
    std::vector<RGp> gps;
    std::vector<RTenantSlot> tenants;
    std::vector<DictSlot> dict;
    std::vector<uint8_t> pool;
    std::vector<uint64_t> expire_at; // by topic id
    // ---- what changed since the last upload ----
    bool full_upload = true, dict_changed = true;
    std::vector<RTenantState*> dirty;
    // ---- bookkeeping ----
    std::map<std::string, std::unique_ptr<RTenantState>> by_name; // byte order of tenant ids
    std::vector<RTenantState*> order;                             // by id_base
    HostDict dict_h;
    std::deque<std::string> strings;
    uint32_t node_free = 0, edge_free = 0, post_free = 0, gp_free = 0;
    uint64_t n_topics = 0;
    std::string error;

    struct Item {
        std::string tenant, topic;
        uint64_t ts = 0;
        uint32_t expiry = 0xFFFFFFFFu;
        bool has_ts = false;
    };
    bool rebuild(std::vector<Item>&& items); // (add / remove between bulk loads: bmq_retain_core.h, on the device)
    bool topic(uint32_t id, std::string_view& tenant, std::string_view& topic, uint64_t* ts = nullptr, uint32_t* expiry = nullptr) const;
    uint64_t n_tenants() const { return by_name.size(); }

private:
    bool refresh(std::vector<RTenantState*>& touched);
};

// directory hash of a tenant's dictionary token (retain direction: tenants are dictionary entries like levels)
BMQ_HD uint32_t tenant_hash(uint32_t token) {
    uint32_t x = token * 0x9E3779B1u;
    return x ^ (x >> 15);
}
// Home bucket of the edge (parent, token): hash(token) + parent.  LINEAR in the parent on purpose: a literal level behind a '+' looks the
// same token up under every node of a node RANGE -- consecutive parents --, and their home buckets are then consecutive 64-byte lines
// (one stream per 64 lanes instead of 64 random lines).  Tokens spread the edges of one parent; parents of one token never collide.
BMQ_HD uint32_t redge_bucket(uint32_t parent, uint32_t token, uint32_t mask) {
    uint32_t h = token * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return (h + parent) & mask;
}

} // namespace bmq
