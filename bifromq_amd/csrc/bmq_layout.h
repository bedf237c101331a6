// bmq_layout.h -- HBM data layout shared by the host builder and the gfx950 kernels.
//
// Dist direction index ("filter trie", the inverse of the reference's per-call topic trie,
// TRIE/TopicTrieNode.java:37-163).  Every trie node is ONE 32-byte slot; a node's id IS its slot index.
//   * all nodes of a tenant live in the tenant's private region of the slot table (TenantSlot);
//   * inside a region, the edge (parent slot, token) hashes to a home BUCKET = two slots = one aligned 64-byte line;
//     insertion fills the first free slot from the home bucket onwards (load factor <= 1/2), so a child lookup reads
//     ONE line in ~92 % of the cases and gets the child's complete header with it.  Measured on MI355X the walk is
//     bound by the rate of random 64-byte line fetches (~54 G lines/s from HBM, ~200 G/s from L2,
//     tools/ubench_lines.hip), so lines per visited node is the figure of merit;
//   * '+' children are reached by slot index: the walk reads the bucket line that contains the slot, so both kinds of
//     work item cost the same single line; tenant roots travel in the directory entry (TenantSlot) and cost nothing;
//   * '#' children are never nodes: the routes of "<path>/#" hang off the parent (hash_*).
#pragma once
#include <stdint.h>

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define BMQ_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define BMQ_HD inline
#endif

namespace bmq {

constexpr uint32_t NONE = 0xFFFFFFFFu;        // empty slot / no child
constexpr uint32_t ROOT_PARENT = 0xFFFFFFFEu; // parent key of tenant roots
constexpr uint32_t TOK_UNKNOWN = 0;           // level string not in the dictionary
constexpr uint32_t TOK_PLUS = 1;              // the '+' edge
constexpr uint32_t TOK_FIRST = 2;             // first dictionary token
constexpr int FAST_LEVELS = 16;               // topics with more levels take the slow path
                                              // (Setting.MaxTopicLevels default, Setting.java:45)
constexpr uint32_t RANGE_INDIRECT = 0x80000000u; // count flag: begin indexes route_pos[] instead of being the first id

struct alignas(32) TrieSlot {
    uint32_t parent;      // region-relative slot of the parent (NONE = empty slot, ROOT_PARENT = tenant root)
    uint32_t token;       // dictionary token of the edge label (TOK_PLUS for '+')
    uint32_t own_begin;   // routes whose filter ends at this node: ids rank_base + own_begin .. +count-1, or
    uint32_t own_count;   //   route_pos[rp_base + own_begin ..] when (own_count & RANGE_INDIRECT)
    uint32_t hash_begin;  // routes of "<this path>/#", same encoding
    uint32_t hash_count;
    uint32_t plus_child;  // region-relative slot of the '+' child or NONE
    uint32_t lit_bloom;   // 32-bit Bloom mask over the literal children's tokens; 0 = no literal child
};
static_assert(sizeof(TrieSlot) == 32, "TrieSlot must be 32 bytes");

// Tenant directory entry: the tenant's region of the slot table.  Bucket k of the region = slots base+2k, base+2k+1.
// Slot indices stored INSIDE a region (TrieSlot.parent, plus_child, the root) are relative to `base`, and route ids
// inside a region are relative to `rank_base`: a region can be rebuilt, moved or re-based without touching the others.
struct alignas(64) TenantSlot {
    uint32_t token;     // dictionary token of the tenant id; 0 = empty directory slot
    uint32_t root;      // slot of the tenant's root node, relative to base
    uint32_t base;      // first slot of the region (even)
    uint32_t buckets;   // number of 2-slot buckets (>= 1)
    uint32_t rank_base; // global route id of the tenant's first route (ids are ranks in KV key order)
    uint32_t rp_base;   // first entry of the tenant in route_pos[]
    // copy of the root slot's payload: every topic of the tenant starts at the root, so the walk takes it from the
    // directory entry it reads anyway instead of spending a line fetch and a round on it
    uint32_t root_hash_begin, root_hash_count; // routes of the filter "#"
    uint32_t root_plus_child, root_lit_bloom;
    uint32_t pad[6];
};
static_assert(sizeof(TenantSlot) == 64, "TenantSlot must be 64 bytes");
constexpr TenantSlot EMPTY_TENANT{0, 0, 0, 1, 0, 0, 0, 0, NONE, 0, {0, 0, 0, 0, 0, 0}};

// Level dictionary: level string -> token, exact (bytes verified).  Strings <= 16 bytes live inline.  Open addressing
// over groups of DICT_GROUP slots (one 64-byte line) at load factor <= 1/4: a lookup reads its home group in one go.
constexpr uint32_t DICT_GROUP = 2;
struct alignas(32) DictSlot {
    uint32_t tag;       // second hash, forced non-zero; 0 = empty slot
    uint32_t token;
    uint32_t len;
    uint32_t pool_off;  // offset of the full string in the pool (only read when len > 16)
    uint32_t inl[4];    // first 16 bytes, little-endian packed, zero padded
};
static_assert(sizeof(DictSlot) == 32, "DictSlot must be 32 bytes");

// Level hash over the level's bytes taken as little-endian 32-bit words (the last word zero padded; an empty
// trailing chunk is not hashed).  Only shifts/adds per word -- 32-bit multiplies are quarter rate on CDNA and the walk
// kernel is instruction-issue bound -- and one multiplicative mix per level.  Same code on host and device.  The hash
// only picks the dictionary slot and a tag: equality is always decided on the bytes.
struct LevelHash {
    uint32_t h1, h2;
};
BMQ_HD uint32_t rotl32(uint32_t x, uint32_t r) { return (x << r) | (x >> (32u - r)); }
BMQ_HD LevelHash level_hash_init() { return {0x811C9DC5u, 0x9747B28Cu}; }
BMQ_HD void level_hash_word(LevelHash& h, uint32_t w) {
    h.h1 = rotl32(h.h1, 5) ^ w;
    h.h1 += h.h1 << 3;
    h.h2 = rotl32(h.h2, 11) + w;
    h.h2 ^= h.h2 >> 7;
}
BMQ_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
BMQ_HD uint32_t level_hash_slot(const LevelHash& h, uint32_t len) { return mix32(h.h1 ^ (len * 0x85EBCA6Bu)); }
BMQ_HD uint32_t level_hash_tag(const LevelHash& h) {
    uint32_t x = h.h2 ^ (h.h1 >> 3);
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 13;
    return x | 1u;
}

// home bucket of edge (parent slot, token) in a region of `buckets` buckets (fastrange: any size)
BMQ_HD uint32_t edge_bucket(uint32_t parent, uint32_t token, uint32_t buckets) {
    uint32_t h = (parent ^ rotl32(token, 16)) * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return (uint32_t)(((uint64_t)h * buckets) >> 32);
}
BMQ_HD uint32_t bloom_bit(uint32_t token) { return (token ^ (token >> 5) ^ (token >> 11)) & 31u; }
BMQ_HD uint32_t tenant_hash(uint32_t token) {
    uint32_t x = token * 0x9E3779B1u;
    return x ^ (x >> 15);
}

// Everything a kernel needs to read the dist index.
struct DistIndexView {
    const TrieSlot* trie;
    const TenantSlot* tenants;
    uint32_t tenant_mask;      // tenant directory slots - 1
    const DictSlot* dict;
    uint32_t dict_group_mask;  // (dictionary slots / DICT_GROUP) - 1
    const uint8_t* pool;       // level strings longer than 16 bytes
    const uint32_t* route_pos; // ids of the (rare) nodes whose route ids are not one contiguous rank range
};

// A matched range of one filter node: ids begin .. begin+count-1, or route_pos[begin ..] if count has RANGE_INDIRECT.
struct MatchRange {
    uint32_t begin, count;
};

} // namespace bmq
