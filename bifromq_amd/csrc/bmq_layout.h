// bmq_layout.h -- HBM data layout shared by the host builder and the gfx950 kernels.
//
// Dist direction index ("filter trie", the inverse of the reference's per-call topic trie,
// TRIE/TopicTrieNode.java:37-163): every node is ONE 32-byte slot of an open-addressing table keyed by
// (parent slot, edge token).  A node's id IS its slot index, so a literal child lookup is a single
// probe that returns the child's complete header; '+' children and tenant roots are reached by slot
// index without probing.  '#' children are never nodes: their routes hang off the parent (hash_*).
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
constexpr int FAST_LEVELS = 16;               // topics with more levels go to the slow path
                                              // (Setting.MaxTopicLevels default, Setting.java:45)

struct alignas(32) TrieSlot {
    uint32_t parent;      // slot index of the parent (NONE = empty slot, ROOT_PARENT = tenant root)
    uint32_t token;       // dictionary token of the edge label (TOK_PLUS for '+')
    uint32_t own_begin;   // routes whose filter ends at this node: route_pos[own_begin .. +own_count)
    uint32_t own_count;
    uint32_t hash_begin;  // routes of "<this path>/#"
    uint32_t hash_count;
    uint32_t plus_child;  // slot index of the '+' child or NONE
    uint32_t lit_bloom;   // 32-bit Bloom mask over the literal children's tokens; 0 = no literal child
};
static_assert(sizeof(TrieSlot) == 32, "TrieSlot must be 32 bytes");

// Level dictionary: level string -> token, exact (bytes verified).  Strings <= 16 bytes live inline.
struct alignas(32) DictSlot {
    uint32_t tag;       // second hash, forced non-zero; 0 = empty slot
    uint32_t token;
    uint32_t len;
    uint32_t pool_off;  // offset of the full string in the pool (only read when len > 16)
    uint32_t inl[4];    // first 16 bytes, little-endian packed, zero padded
};
static_assert(sizeof(DictSlot) == 32, "DictSlot must be 32 bytes");

// Tenant directory: tenant token -> the tenant's root node and its private REGION of the slot table.  All nodes of a
// tenant live in [base, base + size): a wave that works on one tenant's publishes touches only that region, which
// for typical tenants (10^4 routes ~ 1.5 MiB) stays resident in the XCD's 4 MiB L2.
struct alignas(16) TenantSlot {
    uint32_t token; // dictionary token of the tenant id; 0 = empty slot
    uint32_t root;  // slot index of the tenant's root node
    uint32_t base;  // first slot of the region
    uint32_t size;  // slots in the region (any value >= 1, not a power of two)
};
static_assert(sizeof(TenantSlot) == 16, "TenantSlot must be 16 bytes");

// Incremental level hash: two 32-bit lanes (slot index, tag).  Same code on host and device.
struct LevelHash {
    uint32_t h1, h2;
};
BMQ_HD LevelHash level_hash_init() { return {0x811C9DC5u, 0x9747B28Cu}; }
BMQ_HD void level_hash_step(LevelHash& h, uint32_t byte) {
    h.h1 = (h.h1 ^ byte) * 0x01000193u;
    h.h2 = (h.h2 + byte + 1u) * 0x9E3779B1u;
    h.h2 ^= h.h2 >> 15;
}
BMQ_HD uint32_t level_hash_slot(const LevelHash& h, uint32_t len) {
    uint32_t x = h.h1 ^ (len * 0x85EBCA6Bu);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    return x;
}
BMQ_HD uint32_t level_hash_tag(const LevelHash& h) { return h.h2 | 1u; }

BMQ_HD uint32_t edge_hash(uint32_t parent, uint32_t token) {
    uint32_t x = parent * 0x9E3779B1u + token * 0x85EBCA77u;
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
BMQ_HD uint32_t bloom_bit(uint32_t token) { return (token * 0x9E3779B1u) >> 27; }
// home slot of edge (parent, token) inside a region of `size` slots (fastrange: no power-of-two needed)
BMQ_HD uint32_t edge_home(uint32_t parent, uint32_t token, uint32_t size) {
    return (uint32_t)(((uint64_t)edge_hash(parent, token) * size) >> 32);
}
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
    uint32_t dict_mask;
    const uint8_t* pool;       // level strings longer than 16 bytes
    const uint32_t* route_pos; // route ids grouped per node, groups ordered by first id
};

// A matched range: routes route_pos[begin .. begin+count) belong to one filter node.
struct MatchRange {
    uint32_t begin, count;
};

} // namespace bmq
