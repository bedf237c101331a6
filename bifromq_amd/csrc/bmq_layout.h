// bmq_layout.h -- HBM data layout shared by the index builder (bmq_build_core.h: the same code runs as gfx950 kernels and,
// for the sanitizer fuzzers / host-only engines, as plain C++) and the match kernels (bmq_dist_kernels.h).
//
// Dist direction index ("filter trie", the inverse of the reference's per-call topic trie,
// TRIE/TopicTrieNode.java:37-163).  The index LIVES in HBM and is MUTATED there: bmq_rebuild / bmq_routes_apply upload
// route keys and run builder kernels; nothing is built on host cores.
//   * every trie node is ONE 32-byte slot; all nodes of a tenant live in the tenant's private region of the slot table;
//   * a node has a tenant-local NODE ID (the tenant root is id 0, others are handed out by an atomic counter in the
//     tenant's directory entry).  The edge (parent node id, token) hashes to a home BUCKET = two slots = one aligned
//     64-byte line; insertion claims the first free slot from the home bucket onwards with one 64-bit CAS on the edge
//     key (load factor <= 1/2), so a child lookup reads ONE line in ~92 % of the cases and gets the child's complete
//     header with it.  Node ids are not slot positions: a region is grown by re-inserting its slots into a larger
//     region in parallel (k_rehash_region), no id changes;
//   * '+' children are edges with token TOK_PLUS; bit 31 of the parent's Bloom word says whether one exists.  Layout v3 (round 6): the
//     '+' child of node X is placed in the OTHER slot of the line X's own slot lies in whenever that slot is free at the time (else at
//     the hashed home of (X, TOK_PLUS), like a literal edge): 44 % of the nodes a publish topic discovers on the survey's workload are
//     '+' children, and a walk that has fetched X's line then has X/+ with it -- no second line, no second round.  The '+' child of a
//     tenant ROOT sits at its hashed home and the directory entry remembers where (TenantSlot.root_plus);
//   * '#' children are never nodes: the routes of "<path>/#" hang off the parent (hash_*);
//   * the tenant root is not a slot: its payload lives in the directory entry the walk reads anyway.
// Route ids: after bmq_rebuild the id of a route is the rank of its KV key (keys arrive sorted from the KV iterator);
// routes added later by bmq_routes_apply get the next unused ids.  An id never changes and is never reused until the
// next bmq_rebuild; per id the key store holds (offset, length) of the key bytes and a hash of the key's tail.
#pragma once
#include <stdint.h>
#include <stdlib.h>

// Environment switches (profiling experiments: kernel geometry, debug clocks, phase timers ...) exist in -DBMQ_EXPERIMENTS=1 builds only
// (tools/build_variant.sh): in the library that ships bmq_config is the ONLY switch -- a broker's environment cannot change which kernel
// runs ("the shipped kernel is the measured kernel").
#ifndef BMQ_EXPERIMENTS
#define BMQ_EXPERIMENTS 0
#endif
inline const char* bmq_env(const char* name) { return BMQ_EXPERIMENTS ? getenv(name) : nullptr; }

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define BMQ_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define BMQ_HD inline
#endif

namespace bmq {

constexpr uint32_t NONE = 0xFFFFFFFFu;        // empty slot / no child / node id not published yet
constexpr uint32_t TOK_UNKNOWN = 0;           // level string not in the dictionary
constexpr uint32_t TOK_PLUS = 1;              // the '+' edge
constexpr uint32_t TOK_FIRST = 2;             // first dictionary token
#ifndef BMQ_FAST_LEVELS
#define BMQ_FAST_LEVELS 16
#endif
constexpr int FAST_LEVELS = BMQ_FAST_LEVELS;  // topics with more levels take the slow path
                                              // (Setting.MaxTopicLevels default, Setting.java:45)
constexpr uint32_t RANGE_INDIRECT = 0x80000000u; // count flag: begin indexes route_pos[] instead of being the first id
constexpr uint32_t BLOOM_PLUS = 0x80000000u;     // lit_bloom bit 31: the node has a '+' child
constexpr uint64_t EDGE_EMPTY = 0x00000000FFFFFFFFull; // (parent = NONE, token = 0): free slot

struct alignas(32) TrieSlot {
    uint32_t parent;      // tenant-local node id of the parent (0 = tenant root); NONE = free slot
    uint32_t token;       // dictionary token of the edge label (TOK_PLUS for '+')
    uint32_t own_begin;   // routes whose filter ends at this node: ids own_begin .. +count-1, or
    uint32_t own_count;   //   route_pos[own_begin ..] when (own_count & RANGE_INDIRECT)
    uint32_t hash_begin;  // routes of "<this path>/#", same encoding
    uint32_t hash_count;
    uint32_t node;        // this node's tenant-local id (>= 1); NONE until the inserting thread has published it
    uint32_t lit_bloom;   // bits 0-30: Bloom mask over the literal children's tokens; bit 31: a '+' child exists
};
static_assert(sizeof(TrieSlot) == 32, "TrieSlot must be 32 bytes");

// Tenant directory entry = the tenant's region of the slot table + the payload of the tenant's ROOT node (every topic of the
// tenant starts there, so the walk takes it from the directory entry it reads anyway: no line fetch, no round).
// Bucket k of the region = slots base+2k, base+2k+1.  Open addressing by the 64-bit hash of the tenant id; the id's bytes
// (name_off/name_len into the tenant name pool) decide equality.
struct alignas(64) TenantSlot {
    uint32_t hash_lo, hash_hi; // 64-bit hash of the tenant id, forced non-zero; 0/0 = empty directory slot
    uint32_t name_off, name_len;
    uint32_t base;             // first slot of the region (even)
    uint32_t buckets;          // number of 2-slot buckets (>= 1)
    uint32_t n_nodes;          // node ids handed out so far (next id = n_nodes + 1)
    uint32_t n_routes;         // live routes of the tenant
    uint32_t root_hash_begin, root_hash_count; // routes of the filter "#"
    uint32_t root_lit_bloom;                   // Bloom word of the root (bit 31: a first-level '+' exists)
    uint32_t pending;          // builder scratch: upper bound of the nodes the batch being prepared may add
    uint32_t name12[3];        // the first 12 bytes of the tenant id (zero padded): ids up to 12 bytes are compared without the pool
    uint32_t root_plus;        // slot of the root's '+' child, relative to `base` (NONE: there is none): the walk reads it straight away
};
static_assert(sizeof(TenantSlot) == 64, "TenantSlot must be 64 bytes");

// Level dictionary: level string -> token, exact (bytes verified).  Strings <= 16 bytes live inline.  Open addressing
// over groups of DICT_GROUP slots (one 64-byte line) at load factor <= 1/4: a lookup reads its home group in one go.
// Insertion (builder kernels): claim a free slot by CAS on `tag`, fill it, publish `token` last (0 = not yet readable).
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

// home bucket of edge (parent node id, token) in a region of `buckets` buckets (fastrange: any size)
BMQ_HD uint32_t edge_bucket(uint32_t parent, uint32_t token, uint32_t buckets) {
    uint32_t h = (parent ^ rotl32(token, 16)) * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return (uint32_t)(((uint64_t)h * buckets) >> 32);
}
// Bloom bit of a literal child token: 0..30 (bit 31 is BLOOM_PLUS)
BMQ_HD uint32_t bloom_bit(uint32_t token) {
    const uint32_t b = (token ^ (token >> 5) ^ (token >> 11)) & 31u;
    return b - (b == 31u ? 1u : 0u); // 31 -> 30
}
// 64-bit hash of a tenant id (FNV-1a over the bytes, then a finaliser); never 0
BMQ_HD uint64_t tenant_hash_step(uint64_t h, uint32_t byte) { return (h ^ byte) * 0x100000001B3ull; }
constexpr uint64_t TENANT_HASH_INIT = 0xCBF29CE484222325ull;
BMQ_HD uint64_t tenant_hash_final(uint64_t h) {
    h ^= h >> 33;
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 29;
    return h ? h : 1ull;
}

// Everything a match kernel needs to read the dist index.
struct DistIndexView {
    const TrieSlot* trie;
    const TenantSlot* tenants;
    uint32_t tenant_mask;        // tenant directory slots - 1
    const uint8_t* tenant_names; // tenant id bytes (TenantSlot.name_off / name_len)
    const DictSlot* dict;
    uint32_t dict_group_mask;    // (dictionary slots / DICT_GROUP) - 1
    const uint8_t* pool;         // level strings longer than 16 bytes
    const uint32_t* route_pos;   // id lists of the nodes whose route ids are not one contiguous range
};

// A matched range of one filter node: ids begin .. begin+count-1, or route_pos[begin ..] if count has RANGE_INDIRECT.
struct MatchRange {
    uint32_t begin, count;
};

} // namespace bmq
