// bmq_build_core.h -- the index BUILDER of the dist direction: per-item functions that mutate the HBM-resident index
// (bmq_layout.h) in place.  Every function here is BMQ_HD: bmq_engine.hip wraps them in gfx950 kernels (one lane per
// route key / per filter node); bmq_exec_host.h runs the very same functions on host threads for host-only engines
// (build / inspect, never match) and for the sanitizer fuzzers (tools/host_fuzz.cpp) -- there is no second builder.
//
// What it replaces in the reference: nothing is "built" there -- DistWorkerCoProc keeps routes in the KV store and
// TenantRouteMatcher scans them per call (DW/cache/TenantRouteMatcher.java:88-156).  The seams are
//   IKVRangeCoProc.reset(Boundary)   (DW/DistWorkerCoProc.java:283-291)        -> bulk load   (bulk_* + locate + group)
//   ISubscriptionCache.refresh(...)  (DW/DistWorkerCoProc.java:188-209)        -> incremental (prepare + locate + group)
// and the key layout parsed here is SCHEMA/KVSchemaUtil.java:91-130 / SCHEMA/cache/RouteDetailCache.java:53-117:
//   key = 0x00 | u16be(len tenant) | tenant | (level 0x00)* | 0x00 | bucket | flag | receiver | u16be(len receiver)
//
// Pipeline of one batch of n ops (put / delete of a route key), all on the device, in stream order with the match batches:
//   prepare   one lane per op: validate the key, find its tenant, bound the nodes it may add (bulk load: exact, from the
//             common prefix with the previous key of the sorted scan)
//             -> host: create unknown tenants, grow regions / dictionary so that nothing below can run out of space
//   locate    one lane per op: walk / extend the tenant's trie level by level (dictionary + edge inserts are lock-free
//             CAS claims), giving the TARGET of the op = the filter node (and own / '#' list) its route belongs to
//   sort      ops by target, stable (device radix sort) -> ops on one filter are a contiguous GROUP in op order
//   group     one lane per group: apply the group's ops to the filter's id set in order (range <-> id list), no locks
//
// Concurrency rules used throughout: a lane never spins inside a branch (wave64 lanes run in lockstep: the lane it
// waits for may be masked off) -- "not ready yet" always means `continue` of the outermost probe loop; every loop is
// bounded and raises ERR_STUCK instead of hanging the GPU.
#pragma once
#include <sched.h>
#include <stdint.h>

#include "bmq_layout.h"

namespace bmq {

// ------------------------------------------------------------------------------------------------------------
// shared-memory accesses of the builder
// ------------------------------------------------------------------------------------------------------------
// MI355X has eight XCDs with private, mutually NON-coherent L2s, and an agent-scope acquire / release costs a cache-wide
// buffer_inv / buffer_wbl2 (1.7-6.5 us each, MI355X_MICROARCH.md "inter-workgroup visibility").  A builder lane does a dozen
// dependent probes per key, so fences per probe are out of the question (measured: locate of 100 k ops 1.9 ms with acquire
// loads).  Instead every access to memory that ANOTHER lane of the SAME kernel may write goes around the caches:
//   * claims and counters are relaxed agent-scope read-modify-write atomics (executed at the memory side, coherent by nature);
//   * payload written before a publication is stored with relaxed agent-scope atomic stores (`sc1`: write-through, the line
//     leaves the L2), then `s_waitcnt vmcnt(0)` drains them, then the flag is stored the same way (atom_publish);
//   * readers poll the flag and read the payload with relaxed agent-scope atomic loads (`sc1`), which is sufficient exactly
//     because the producer stored `sc1` (same document, "handoff-flag").
// Data produced by an EARLIER kernel (or read only by later kernels) is accessed with plain loads / stores: kernel boundaries
// are the acquire / release.  On the host (HostExec, sanitizer builds) the same functions use acquire / release atomics.
#if defined(__HIP_DEVICE_COMPILE__)
#define BMQ_MO_LOAD __ATOMIC_RELAXED
#define BMQ_MO_STORE __ATOMIC_RELAXED
#define BMQ_MO_RMW __ATOMIC_RELAXED
template <class T> BMQ_HD T atom_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> BMQ_HD void shared_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> BMQ_HD T shared_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> BMQ_HD void atom_publish(T* p, T v) { // everything this lane stored before is in memory before the flag is
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T> BMQ_HD T atom_cas(T* p, T expect, T desired) { // returns the value found
    __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;
}
template <class T> BMQ_HD T atom_add(T* p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> BMQ_HD T atom_or(T* p, T v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A look through the caches (an ordinary load) at something that never changes once it is PUBLISHED -- a dictionary slot with its token, a trie
// slot with its node id.  A hit is final whatever the age of the line; a miss says nothing (the line may be older than the entry) and sends the
// caller to the coherent loads above.  (Round 6: every lane of a 100 k-op batch used to read the dictionary slots of the few level strings
// all keys share -- "churn", the eight first-level tokens -- around the caches: hundreds of thousands of requests for a handful of lines,
// one L2 channel each, and k_b_locate took 114 us however its passes were arranged.)
template <class T> BMQ_HD T peek_load(const T* p) { return *reinterpret_cast<const volatile T*>(p); }
BMQ_HD void spin_pause(uint32_t) {}
#else
template <class T> BMQ_HD T atom_load(const T* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
template <class T> BMQ_HD void shared_store(T* p, T v) { *p = v; } // ordered by the release store of the flag that follows
template <class T> BMQ_HD T shared_load(const T* p) { return *p; } // ordered by the acquire load of the flag before
template <class T> BMQ_HD void atom_publish(T* p, T v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
template <class T> BMQ_HD T atom_cas(T* p, T expect, T desired) { // returns the value found
    __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    return expect;
}
template <class T> BMQ_HD T atom_add(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
template <class T> BMQ_HD T atom_or(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_ACQ_REL); }
template <class T> BMQ_HD T peek_load(const T* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); } // (the host has one kind of load)
// A host thread that waits for another thread's publication gives the processor away after a few looks: the publisher may have lost it
// between its claim and its publication (fuzzer threads on a busy box: the spin bound -- iterations, not time -- ran out: ERR_STUCK
// without anything being stuck, one run in ten with the CPU test tier running beside it).  On the GPU the publishing lane is running by construction.
BMQ_HD void spin_pause(uint32_t spins) {
    if (spins > 64) sched_yield();
}
#endif

// ------------------------------------------------------------------------------------------------------------
// state
// ------------------------------------------------------------------------------------------------------------
enum : uint32_t {
    ERR_BAD_KEY = 1u,     // malformed route key or op code (nothing was changed: raised by prepare)
    ERR_UNSORTED = 2u,    // bulk load: keys not strictly ascending (host sorts + de-duplicates and retries)
    ERR_STUCK = 4u,       // a bounded probe loop ran out (damaged table): the batch is abandoned
    ERR_DICT_FULL = 8u,   // dictionary / string pool out of room despite prepare's bound (cannot happen; checked anyway)
    ERR_REGION_FULL = 16u // tenant region out of room despite prepare's bound (cannot happen; checked anyway)
};

// Persistent counters of the index + per-batch results of the builder kernels (device memory; the host reads it back
// after prepare and after group).
// Same-address atomics serialise in the L2 atomic unit (~15 ns each under contention: 23 M node creations of a bulk load bumping
// ONE counter cost 350 ms), so nothing hot counts into a single word: nodes are counted per tenant (TenantSlot.n_nodes), batch
// results into N_CTR_LANES copies indexed by the work item, summed by the host.
constexpr uint32_t N_CTR_LANES = 64;
struct BuildCounters {
    // persistent
    unsigned long long rp_used;    // words handed out in route_pos (word 0 is never used)
    unsigned long long rp_garbage; // words of abandoned id lists
    uint32_t n_tokens;             // dictionary tokens handed out (next = TOK_FIRST + n_tokens)
    uint32_t dpool_used;           // bytes used in the dictionary string pool
    // bmq_routes_apply's stages run back to back with ONE read-back at the end: the first stage that finds something the host has to
    // deal with (an error, an unknown tenant, a region / the dictionary / the id-list pool to grow) closes the gate -- gate = its number:
    // 1 prepare, 2 locate, 3 group --, the stages behind it return at once and the host takes over from that stage (they are idempotent).
    // Not part of the per-batch block below: it has to survive the zeroing between the stages.
    uint32_t gate;
    uint32_t pad_gate;
    // per batch (zeroed by the host before prepare)
    uint32_t err;
    uint32_t n_unknown;            // ops whose tenant is not in the directory (listed in unknown_list)
    uint32_t n_grow;               // tenants whose region must grow first (listed in grow_list)
    uint32_t n_deferred;           // groups that found the id-list pool full (re-run after the host grew it)
    unsigned long long rp_need;    // id-list words the deferred groups asked for
    uint32_t n_bulk_tenants;       // bulk load: number of tenants (runs of equal tenant id in the sorted scan)
    uint32_t pad0;
    uint32_t n_dups[N_CTR_LANES];    // puts of keys that were already there
    uint32_t n_removed[N_CTR_LANES]; // deletes that removed a route
    uint32_t n_added[N_CTR_LANES];   // puts that added a route
};

// Everything the builder kernels touch.  All pointers are device pointers (host pointers under HostExec).
struct DistIndexMut {
    TrieSlot* trie;
    TenantSlot* tenants;
    uint32_t tenant_mask;
    const uint8_t* tenant_names;
    DictSlot* dict;
    uint32_t dict_group_mask;
    uint8_t* dpool;
    uint32_t dpool_cap;
    uint32_t* route_pos;
    unsigned long long rp_cap;
    // key store: per route id
    unsigned long long* kref; // offset into kpool | (key length << 40); 0 = no such route (never existed / deleted)
    uint32_t* khash;          // hash of the key's tail (bucket, flag, receiver, receiver length): membership tests
    uint32_t id_cap;
    const uint8_t* kpool;     // key bytes; readable 16 bytes past the end
    BuildCounters* bc;
    // the fan-out grouping's per-id cache (bmq_fanout_core.h: dgroup[id] = group slot of the route), or null: a deleted route is
    // marked there by the very lane that deletes it, so the grouping learns about it with the one gather it does anyway
    uint32_t* fo_dgroup;
    uint32_t fo_cap;
};
constexpr uint32_t FO_DEAD_ID = 0xFFFFFFFEu; // fo_dgroup[id]: the route has been deleted (ids are not reused before the next rebuild)
constexpr unsigned KREF_LEN_SHIFT = 40;
constexpr unsigned long long KREF_OFF_MASK = (1ull << KREF_LEN_SHIFT) - 1;

// One batch of ops.  Key i = kpool[key_base + key_off[i] .. key_base + key_off[i+1]).
struct OpBatch {
    unsigned long long key_base;
    const uint32_t* key_off; // [n + 1]
    const uint8_t* op;       // [n] 0 = put, 1 = delete; null = all puts
    uint32_t n;
    uint32_t id_base;        // a put's id = id_base + (number of puts before it in the batch) -- see put_rank
    const uint32_t* put_rank;// [n] exclusive count of puts before op i; null (bulk load) = i
    uint32_t bulk;           // 1 = bulk load into an empty index: keys strictly ascending, all puts, no membership tests
    uint32_t sized;          // 1 = the tenants' regions already hold whatever the batch adds (DistIndex::reserve_like): no growth bound is taken
    // scratch, [n] each unless noted
    uint32_t* dir_slot;      // prepare: directory slot of the op's tenant, NONE = unknown
    uint32_t* nn;            // bulk prepare: nodes the key adds (exact); then reused as tenant index after the scan
    uint32_t* flag;          // bulk prepare: 1 = first key of a tenant
    unsigned long long* target;  // locate: target of the op (TARGET_NONE = nothing to do)
    uint32_t* order;         // op indices sorted by target (stable)
    unsigned long long* sorted_target;
    uint8_t* group_done;     // [n] per sorted position: group head already applied (re-runs after pool growth)
    uint32_t* unknown_list;  // [n] op indices with an unknown tenant
    uint32_t* grow_list;     // [2 * (tenant_mask + 1)] (directory slot, nodes needed) pairs
    // bulk load tenant table, [n_tenants] each (filled by bulk_tenants_one, read back by the host)
    uint32_t* bt_first;      // index of the tenant's first key
    uint32_t* bt_nodes;      // exact number of trie nodes of the tenant
    uint32_t* bt_keys;       // number of keys
    const uint32_t* bt_dir;  // host -> device: directory slot of tenant t
};
// target = kind << 62 | slot: kind 0 own list of trie slot `slot` (absolute index), 1 its '#' list, 2 the '#' list of the
// tenant root (slot = directory slot)
constexpr uint32_t PENDING_FORCE = 0x80000000u; // TenantSlot.pending flag: grow the region whatever the count says
constexpr unsigned long long TARGET_NONE = ~0ull;
constexpr unsigned TARGET_KIND_SHIFT = 40;
BMQ_HD unsigned long long make_target(uint32_t kind, unsigned long long slot) { return ((unsigned long long)kind << TARGET_KIND_SHIFT) | slot; }

// ------------------------------------------------------------------------------------------------------------
// bytes
// ------------------------------------------------------------------------------------------------------------
// 4 bytes at byte offset p (any alignment), little endian.  The buffer is readable 16 bytes past its last byte.
BMQ_HD uint32_t bytes_word_at(const uint8_t* base, unsigned long long p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t* a = reinterpret_cast<const uint32_t*>(base + (p & ~3ull));
    return __builtin_amdgcn_alignbyte(a[1], a[0], (uint32_t)(p & 3ull));
#else
    uint32_t w;
    __builtin_memcpy(&w, base + p, 4);
    return w;
#endif
}
BMQ_HD uint32_t be16_at(const uint8_t* base, unsigned long long p) { return ((uint32_t)base[p] << 8) | base[p + 1]; }

// One level starting at pos: bytes up to the next separator byte (SEP4 = the separator in all four byte lanes) or `end`.
// Returns hash, first 16 bytes, length; `pos` ends up behind the level (on the separator or at end).
template <uint32_t SEP4>
BMQ_HD void scan_level_bytes(const uint8_t* base, unsigned long long& pos, unsigned long long end, LevelHash& h, uint32_t inl[4],
                             uint32_t& len) {
    h = level_hash_init();
    inl[0] = inl[1] = inl[2] = inl[3] = 0;
    len = 0;
    for (;;) {
        const unsigned long long remaining = end - pos;
        if (remaining == 0) break;
        const uint32_t w = bytes_word_at(base, pos);
        uint32_t nb = 4;
        const uint32_t x = w ^ SEP4;
        const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u; // exact for the lowest hit, which is all that is used
        if (z) nb = (uint32_t)__builtin_ctz(z) >> 3;
        if (nb > remaining) nb = (uint32_t)remaining;
        if (nb) {
            const uint32_t wm = nb == 4 ? w : (w & ((1u << (8u * nb)) - 1u));
            level_hash_word(h, wm);
            if (len < 16) inl[len >> 2] = wm;
            len += nb;
            pos += nb;
        }
        if (nb < 4) break;
    }
}

// A route key taken apart (offsets into kpool).  RouteDetailCache.java:53-109 parses from both ends in the same way.
struct KeyView {
    unsigned long long start, end;           // whole key
    unsigned long long tenant, tenant_end;
    unsigned long long esc, esc_end;         // escaped filter: levels separated by NUL (no trailing NUL)
    unsigned long long tail;                 // bucket byte: [tail, end) = bucket, flag, receiver, u16be(len receiver)
    uint32_t flag;
};
BMQ_HD bool key_parse(const uint8_t* kp, unsigned long long start, unsigned long long end, KeyView& k) {
    const unsigned long long len = end - start;
    if (len < 3 + 2 + 2 + 2 || kp[start] != 0) return false;
    const unsigned long long tlen = be16_at(kp, start + 1), rlen = be16_at(kp, end - 2);
    if (len < 3 + tlen + 4 + rlen + 2) return false;
    k.start = start;
    k.end = end;
    k.tenant = start + 3;
    k.tenant_end = k.tenant + tlen;
    k.esc = k.tenant_end;
    const unsigned long long recv = end - 2 - rlen;
    k.esc_end = recv - 4;
    k.tail = recv - 2;
    if (kp[k.esc_end] != 0 || kp[k.esc_end + 1] != 0) return false;
    k.flag = kp[recv - 1];
    return k.flag >= 1 && k.flag <= 3;
}
BMQ_HD uint64_t tenant_hash_bytes(const uint8_t* b, unsigned long long beg, unsigned long long end) {
    uint64_t h = TENANT_HASH_INIT;
    for (unsigned long long i = beg; i < end; i++) h = tenant_hash_step(h, b[i]);
    return tenant_hash_final(h);
}
// 16 bytes at byte offset p (any alignment) as four little-endian words, the bytes from n on (n <= 16) read as zero.  Five aligned words,
// requested together: ONE memory latency instead of sixteen (a byte loop waits for every byte; the builder's lanes run at 1-2 waves per SIMD,
// nothing hides a latency there).  Key-pool only: the pool is readable 32 bytes past its last key (DistIndex::ensure_keys).
BMQ_HD void bytes_chunk16(const uint8_t* base, unsigned long long p, uint32_t n, uint32_t w[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t* a = reinterpret_cast<const uint32_t*>(base + (p & ~3ull));
    const uint32_t sh = (uint32_t)(p & 3ull);
    const uint32_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4];
    w[0] = __builtin_amdgcn_alignbyte(a1, a0, sh);
    w[1] = __builtin_amdgcn_alignbyte(a2, a1, sh);
    w[2] = __builtin_amdgcn_alignbyte(a3, a2, sh);
    w[3] = __builtin_amdgcn_alignbyte(a4, a3, sh);
#else
    __builtin_memcpy(w, base + p, 16);
#endif
    for (uint32_t i = 0; i < 4; i++) {
        const uint32_t have = n > 4 * i ? n - 4 * i : 0u; // valid bytes of word i
        if (have < 4) w[i] &= have ? (1u << (8 * have)) - 1u : 0u;
    }
}
// hash of a key's tail (bucket, flag, receiver, receiver length): the builder's membership tests compare it before they compare bytes
BMQ_HD uint32_t tail_hash(const uint8_t* kp, unsigned long long beg, unsigned long long end) {
    uint32_t h = 0x811C9DC5u ^ (uint32_t)(end - beg);
    for (unsigned long long p = beg; p < end; p += 16) {
        uint32_t w[4];
        bytes_chunk16(kp, p, (uint32_t)(end - p < 16 ? end - p : 16), w);
        h = (h ^ w[0]) * 0x01000193u;
        h = (rotl32(h, 13) ^ w[1]) * 0x01000193u;
        h = (rotl32(h, 13) ^ w[2]) * 0x01000193u;
        h = (rotl32(h, 13) ^ w[3]) * 0x01000193u;
    }
    return mix32(h) | 1u;
}
// kp[ao, ao + n) == kp[bo, bo + n), 16 bytes per step (key pool only, see bytes_chunk16)
BMQ_HD bool pool_bytes_equal(const uint8_t* kp, unsigned long long ao, unsigned long long bo, unsigned long long n) {
    for (unsigned long long i = 0; i < n; i += 16) {
        uint32_t x[4], y[4];
        const uint32_t m = (uint32_t)(n - i < 16 ? n - i : 16);
        bytes_chunk16(kp, ao + i, m, x);
        bytes_chunk16(kp, bo + i, m, y);
        if (((x[0] ^ y[0]) | (x[1] ^ y[1]) | (x[2] ^ y[2]) | (x[3] ^ y[3])) != 0) return false;
    }
    return true;
}
BMQ_HD bool bytes_equal(const uint8_t* a, unsigned long long ao, const uint8_t* b, unsigned long long bo, unsigned long long n) {
    for (unsigned long long i = 0; i < n; i++)
        if (a[ao + i] != b[bo + i]) return false;
    return true;
}
// <0, 0, >0: unsigned byte order, a proper prefix sorts first (KV order)
BMQ_HD int bytes_compare(const uint8_t* p, unsigned long long a, unsigned long long an, unsigned long long b, unsigned long long bn) {
    const unsigned long long m = an < bn ? an : bn;
    for (unsigned long long i = 0; i < m; i++) {
        const int d = (int)p[a + i] - (int)p[b + i];
        if (d) return d;
    }
    return an < bn ? -1 : (an > bn ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------------------
// tenant directory
// ------------------------------------------------------------------------------------------------------------
// directory slot of the tenant whose id is kp[beg, end), or NONE
BMQ_HD uint32_t tenant_find(const TenantSlot* dir, uint32_t mask, const uint8_t* names, const uint8_t* kp, unsigned long long beg,
                            unsigned long long end) {
    const uint64_t h = tenant_hash_bytes(kp, beg, end);
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32), len = (uint32_t)(end - beg);
    uint32_t d = (lo ^ hi) & mask;
    for (uint32_t probes = 0; probes <= mask; probes++) {
        const TenantSlot& t = dir[d];
        if (t.hash_lo == 0 && t.hash_hi == 0) return NONE;
        if (t.hash_lo == lo && t.hash_hi == hi && t.name_len == len && bytes_equal(names, t.name_off, kp, beg, len)) return d;
        d = (d + 1) & mask;
    }
    return NONE;
}

// ------------------------------------------------------------------------------------------------------------
// dictionary: level string -> token, find or insert
// ------------------------------------------------------------------------------------------------------------
// kp[start, start+len) is the level; h / inl as produced by scan_level_bytes.  insert = false: TOK_UNKNOWN when absent.
BMQ_HD uint32_t dict_intern(const DistIndexMut& ix, const LevelHash& h, uint32_t len, const uint32_t inl[4], const uint8_t* kp,
                            unsigned long long start, bool insert) {
    const uint32_t tag = level_hash_tag(h);
    uint32_t g = level_hash_slot(h, len) & ix.dict_group_mask;
    // through the caches first: the home group's slots, if they hold this string with its token published (peek_load)
    for (uint32_t jj = 0; jj < DICT_GROUP; jj++) {
        const DictSlot* s = ix.dict + DICT_GROUP * (size_t)g + jj;
        if (peek_load(&s->tag) != tag) continue;
        const uint32_t tok = peek_load(&s->token);
        if (tok == 0 || peek_load(&s->len) != len || peek_load(&s->inl[0]) != inl[0] || peek_load(&s->inl[1]) != inl[1] || peek_load(&s->inl[2]) != inl[2] ||
            peek_load(&s->inl[3]) != inl[3])
            continue;
        bool eq = true;
        if (len > 16) {
            const uint32_t po = peek_load(&s->pool_off);
            for (uint32_t i = 16; i < len && eq; i++) eq = peek_load(ix.dpool + po + i) == kp[start + i];
        }
        if (eq) return tok;
    }
    uint32_t j = 0, probes = 0, spins = 0;
    // an insert gives up early (the host grows the table and re-runs locate) instead of crawling through a nearly full table
    const uint32_t max_probes = insert && ix.dict_group_mask > 128u ? 128u : ix.dict_group_mask;
    for (;;) {
        if (probes > max_probes || spins > (1u << 22)) {
            atom_or(&ix.bc->err, probes > max_probes ? (uint32_t)ERR_DICT_FULL : (uint32_t)ERR_STUCK);
            return TOK_UNKNOWN;
        }
        DictSlot* s = ix.dict + DICT_GROUP * (size_t)g + j;
        uint32_t t = atom_load(&s->tag);
        if (t == 0) {
            if (!insert) return TOK_UNKNOWN;
            t = atom_cas(&s->tag, 0u, tag);
            if (t == 0) { // claimed: fill, then publish the token
                uint32_t off = 0;
                if (len > 16) {
                    off = atom_add(&ix.bc->dpool_used, (len + 3u) & ~3u);
                    if ((unsigned long long)off + len > ix.dpool_cap) {
                        atom_or(&ix.bc->err, (uint32_t)ERR_DICT_FULL);
                        off = 0; // keeps every reader inside the pool; the batch is abandoned by the host
                    } else
                        for (uint32_t i = 0; i < len; i++) shared_store(ix.dpool + off + i, kp[start + i]);
                }
                shared_store(&s->len, len);
                shared_store(&s->pool_off, off);
                shared_store(&s->inl[0], inl[0]);
                shared_store(&s->inl[1], inl[1]);
                shared_store(&s->inl[2], inl[2]);
                shared_store(&s->inl[3], inl[3]);
                const uint32_t tok = TOK_FIRST + atom_add(&ix.bc->n_tokens, 1u);
                atom_publish(&s->token, tok);
                return tok;
            }
        }
        if (t == tag) {
            const uint32_t tok = atom_load(&s->token);
            if (tok == 0) { // its inserter has not published yet: look again (never spin inside a branch)
                spin_pause(spins++);
                continue;
            }
            bool eq = shared_load(&s->len) == len && shared_load(&s->inl[0]) == inl[0] && shared_load(&s->inl[1]) == inl[1] &&
                      shared_load(&s->inl[2]) == inl[2] && shared_load(&s->inl[3]) == inl[3];
            if (eq && len > 16) {
                const uint32_t po = shared_load(&s->pool_off);
                for (uint32_t i = 16; i < len && eq; i++) eq = shared_load(ix.dpool + po + i) == kp[start + i];
            }
            if (eq) return tok;
        }
        if (++j == DICT_GROUP) { // next group; a lookup may stop at a group with a free slot only when it is not inserting
            j = 0;
            g = (g + 1) & ix.dict_group_mask;
            probes++;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// trie: child of (parent node id, token) in a tenant's region, find or insert
// ------------------------------------------------------------------------------------------------------------
// Returns the child's node id (NONE: absent and !insert, or failure) and its absolute slot index in *slot_abs.
// `created` tells the caller to record the child in the parent's Bloom word.
// parent_slot: absolute slot index of the parent's own slot (PARENT_IS_ROOT for the tenant root, which has none).
// Layout v3: the '+' child of a node X is looked for in the OTHER slot of X's line first, then along the probe sequence of its hashed home
// (where a region growth leaves a '+' child that did not sit beside its parent before, and where it is created when that other slot is
// taken); it is CREATED beside X whenever that slot is free.  Readers therefore never conclude anything from a free neighbour slot.
//   Two lanes creating X/+ at once agree: slots only go from free to taken, so either both see the neighbour slot taken by another edge
//   (both go to the hashed home, where the first-free protocol of literal edges applies) or the loser of the CAS on it finds the winner's key.
// ONE loop, one step per iteration (a probe, a claim, a look at a slot somebody else is filling): a lane that has to wait for another
// lane's publication goes round the SAME loop the publishing lane is in.  (A first version waited in an inner loop of its own: where the
// compiler put that loop in front of the claiming lane's "claim, publish" block of the same wave, the waiting lane span for ever -- a wave's
// lanes run a divergent branch one side after the other -- and the GPU build ended with ERR_STUCK; the host executor never showed it.)
constexpr unsigned long long PARENT_IS_ROOT = ~0ull;
BMQ_HD uint32_t trie_child(const DistIndexMut& ix, TenantSlot* ten, uint32_t base, uint32_t buckets, uint32_t parent, unsigned long long parent_slot,
                           uint32_t token, bool insert, unsigned long long& slot_abs, bool& created) {
    created = false;
    const uint64_t key = (uint64_t)parent | ((uint64_t)token << 32);
    const bool root_plus = token == TOK_PLUS && parent_slot == PARENT_IS_ROOT;
    const bool try_beside = token == TOK_PLUS && parent_slot != PARENT_IS_ROOT;
    const unsigned long long beside_abs = parent_slot ^ 1ull; // the other slot of the parent's line (try_beside only)
    // phase 0: the quick looks (the directory entry's reference to the root's '+' child / the slot beside the parent); 1: a pure lookup along
    // the hashed home's probe sequence (an insert that may go beside the parent: the child may have been left at its hashed home by a
    // region growth); 2: absent there -- claim the slot beside the parent; 3: find or insert along the hashed home's probe sequence
    uint32_t phase = 0, spins = 0, bk = 0, j = 0, probes = 0;
    TrieSlot* wait = nullptr; // a slot that holds `key`: its node id, once the claiming lane has published it
    unsigned long long wait_abs = 0;
    // through the caches first (peek_load): the slot beside the parent, then the two slots of the hashed home -- a slot that holds the edge
    // with its node id published is the answer, whatever the age of the line
    {
        auto peek = [&](unsigned long long abs) -> uint32_t {
            const TrieSlot* s = ix.trie + (size_t)abs;
            if (peek_load(reinterpret_cast<const uint64_t*>(s)) != key) return NONE;
            return peek_load(&s->node);
        };
        if (try_beside) {
            const uint32_t id = peek(beside_abs);
            if (id != NONE) {
                slot_abs = beside_abs;
                return id;
            }
        }
        if (!root_plus) {
            const unsigned long long home = (unsigned long long)base + 2ull * edge_bucket(parent, token, buckets);
            for (uint32_t jj = 0; jj < 2; jj++) {
                const uint32_t id = peek(home + jj);
                if (id != NONE) {
                    slot_abs = home + jj;
                    return id;
                }
            }
        }
    }
    for (;;) {
        if (spins > (1u << 22)) {
            atom_or(&ix.bc->err, (uint32_t)ERR_STUCK);
            return NONE;
        }
        if (wait) {
            const uint32_t id = atom_load(&wait->node);
            if (id == NONE) { // claimed by another lane a moment ago
                spin_pause(spins++);
                continue;
            }
            slot_abs = wait_abs;
            if (root_plus && atom_load(&ten->root_plus) == NONE) atom_publish(&ten->root_plus, (uint32_t)(wait_abs - base)); // (found before its creator said where)
            return id;
        }
        if (phase == 0) {
            if (root_plus) {
                const uint32_t rp = atom_load(&ten->root_plus);
                if (rp != NONE) {
                    wait = ix.trie + (size_t)base + rp, wait_abs = (unsigned long long)base + rp;
                    continue;
                }
            }
            if (try_beside && atom_load(reinterpret_cast<uint64_t*>(ix.trie + (size_t)beside_abs)) == key) {
                wait = ix.trie + (size_t)beside_abs, wait_abs = beside_abs;
                continue;
            }
            phase = (try_beside && insert) ? 1u : 3u;
            bk = edge_bucket(parent, token, buckets), j = 0, probes = 0;
            continue;
        }
        if (phase == 2) {
            TrieSlot* s = ix.trie + (size_t)beside_abs;
            const uint64_t k = atom_cas(reinterpret_cast<uint64_t*>(s), EDGE_EMPTY, key);
            if (k == EDGE_EMPTY) { // claimed (payload of a free slot is all zero, node = NONE): publish the node id
                const uint32_t id = atom_add(&ten->n_nodes, 1u) + 1u;
                atom_publish(&s->node, id);
                slot_abs = beside_abs;
                created = true;
                return id;
            }
            if (k == key) { // another lane put it there meanwhile
                wait = s, wait_abs = beside_abs;
                continue;
            }
            phase = 3; // taken by another edge: the hashed home
            bk = edge_bucket(parent, token, buckets), j = 0, probes = 0;
            continue;
        }
        // phases 1 and 3: one slot of the probe sequence
        const bool ins = insert && phase == 3;
        if (probes >= (ins && buckets > 256u ? 256u : buckets)) {
            if (phase == 1) {
                phase = 2;
                continue;
            }
            if (!ins) return NONE; // a lookup that wrapped around a full region: absent
            atom_or(&ten->pending, PENDING_FORCE); // the host moves the tenant into a larger region and re-runs locate
            atom_or(&ix.bc->err, (uint32_t)ERR_REGION_FULL);
            return NONE;
        }
        TrieSlot* s = ix.trie + (size_t)base + 2 * (size_t)bk + j;
        uint64_t* kp64 = reinterpret_cast<uint64_t*>(s);
        uint64_t k = atom_load(kp64);
        if (k == EDGE_EMPTY) { // first-free placement: an edge is never stored behind a free slot of its probe sequence
            if (phase == 1) {
                phase = 2;
                continue;
            }
            if (!ins) return NONE;
            k = atom_cas(kp64, EDGE_EMPTY, key);
            if (k == EDGE_EMPTY) {
                const uint32_t id = atom_add(&ten->n_nodes, 1u) + 1u;
                atom_publish(&s->node, id);
                slot_abs = (unsigned long long)base + 2ull * bk + j;
                if (root_plus) atom_publish(&ten->root_plus, 2u * bk + j);
                created = true;
                return id;
            }
        }
        if (k == key) {
            wait = s, wait_abs = (unsigned long long)base + 2ull * bk + j;
            continue;
        }
        if (++j == 2) {
            j = 0;
            bk = (bk + 1 == buckets) ? 0 : bk + 1;
            probes++;
        }
    }
}

// Levels of an escaped filter.  Calls f(level_start, len, hash, inl, is_last) for each level; f returns false to stop.
template <class F> BMQ_HD void for_each_level(const uint8_t* kp, unsigned long long esc, unsigned long long esc_end, F&& f) {
    unsigned long long pos = esc;
    for (;;) {
        LevelHash h;
        uint32_t inl[4], len;
        const unsigned long long start = pos;
        scan_level_bytes<0u>(kp, pos, esc_end, h, inl, len);
        const bool last = pos >= esc_end;
        if (!f(start, len, h, inl, last)) return;
        if (last) return;
        pos++; // the NUL
    }
}
BMQ_HD bool level_is_hash(const uint8_t* kp, unsigned long long start, uint32_t len, bool last) { return last && len == 1 && kp[start] == '#'; }
BMQ_HD bool level_is_plus(const uint8_t* kp, unsigned long long start, uint32_t len) { return len == 1 && kp[start] == '+'; }

// number of trie nodes on the key's path (levels, not counting a trailing '#') and the byte length of its levels
BMQ_HD void key_level_stats(const uint8_t* kp, const KeyView& k, uint32_t& node_levels, uint32_t& level_bytes) {
    uint32_t n = 0, b = 0;
    for_each_level(kp, k.esc, k.esc_end, [&](unsigned long long start, uint32_t len, const LevelHash&, const uint32_t*, bool last) {
        if (!level_is_hash(kp, start, len, last)) n++;
        b += (len + 3u) & ~3u;
        return true;
    });
    node_levels = n;
    level_bytes = b;
}

BMQ_HD bool batch_key(const DistIndexMut& ix, const OpBatch& ob, uint32_t i, KeyView& k) {
    return key_parse(ix.kpool, ob.key_base + ob.key_off[i], ob.key_base + ob.key_off[i + 1], k);
}

// ------------------------------------------------------------------------------------------------------------
// prepare (incremental batches): validate, find the tenant, bound the growth
// ------------------------------------------------------------------------------------------------------------
BMQ_HD void prepare_one(const DistIndexMut& ix, const OpBatch& ob, uint32_t i) {
    KeyView k;
    const bool is_put = !ob.op || ob.op[i] == 0;
    ob.target[i] = TARGET_NONE;
    if ((ob.op && ob.op[i] > 1) || !batch_key(ix, ob, i, k)) {
        atom_or(&ix.bc->err, (uint32_t)ERR_BAD_KEY);
        ob.dir_slot[i] = NONE;
        return;
    }
    const uint32_t d = tenant_find(ix.tenants, ix.tenant_mask, ix.tenant_names, ix.kpool, k.tenant, k.tenant_end);
    ob.dir_slot[i] = d;
    if (!is_put) return; // a delete adds nothing; with an unknown tenant it is a no-op
    uint32_t nl, lb;
    key_level_stats(ix.kpool, k, nl, lb);
    if (d == NONE) {
        const uint32_t p = atom_add(&ix.bc->n_unknown, 1u);
        ob.unknown_list[p] = i;
    } else if (!ob.sized) { // (a batch of a generation change: nl, the bound, would ask for room the exactly sized region does not need)
        atom_add(&ix.tenants[d].pending, nl);
    }
}
// one lane per directory slot: does the tenant's region hold what the batch may add at load factor <= 1/2?
BMQ_HD void prepare_check_one(const DistIndexMut& ix, const OpBatch& ob, uint32_t d) {
    TenantSlot& t = ix.tenants[d];
    if ((t.hash_lo | t.hash_hi) == 0 || t.pending == 0) return;
    unsigned long long need = (unsigned long long)t.n_nodes + (t.pending & ~PENDING_FORCE);
    if ((t.pending & PENDING_FORCE) && need <= t.buckets) need = (unsigned long long)t.buckets + 1; // an insert ran out of probes
    t.pending = 0;
    if (need > t.buckets) { // slots = 2 * buckets: nodes <= buckets keeps the load factor at 1/2
        const uint32_t p = atom_add(&ix.bc->n_grow, 1u);
        ob.grow_list[2 * p] = d;
        ob.grow_list[2 * p + 1] = need > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)need;
    }
}

// ------------------------------------------------------------------------------------------------------------
// bulk load: the sorted scan of a KV range (IKVRangeCoProc.reset).  Exact sizing from neighbouring keys.
// ------------------------------------------------------------------------------------------------------------
// flag[i] = key i starts a new tenant; nn[i] = trie nodes key i adds that key i-1 (same tenant) did not: the levels behind the
// longest common level prefix.  Also checks strict ascending order (== no duplicates).
BMQ_HD void bulk_prepare_one(const DistIndexMut& ix, const OpBatch& ob, uint32_t i) {
    KeyView k;
    ob.target[i] = TARGET_NONE;
    ob.flag[i] = 0;
    ob.nn[i] = 0;
    if (!batch_key(ix, ob, i, k)) {
        atom_or(&ix.bc->err, (uint32_t)ERR_BAD_KEY);
        return;
    }
    uint32_t nl, lb;
    key_level_stats(ix.kpool, k, nl, lb);
    bool new_tenant = i == 0;
    uint32_t shared = 0;
    if (i > 0) {
        KeyView p;
        if (!batch_key(ix, ob, i - 1, p)) return; // reported by lane i-1
        if (bytes_compare(ix.kpool, p.start, p.end - p.start, k.start, k.end - k.start) >= 0) atom_or(&ix.bc->err, (uint32_t)ERR_UNSORTED);
        new_tenant = (p.tenant_end - p.tenant) != (k.tenant_end - k.tenant) ||
                     !bytes_equal(ix.kpool, p.tenant, ix.kpool, k.tenant, k.tenant_end - k.tenant);
        if (!new_tenant) { // leading levels equal in both keys (a level counts only if both keys have a node for it)
            unsigned long long a = p.esc, b = k.esc;
            for (;;) {
                // compare one level: bytes up to NUL / end in both
                unsigned long long ia = a, ib = b;
                while (ia < p.esc_end && ib < k.esc_end && ix.kpool[ia] != 0 && ix.kpool[ib] != 0 && ix.kpool[ia] == ix.kpool[ib]) {
                    ia++;
                    ib++;
                }
                const bool enda = ia >= p.esc_end || ix.kpool[ia] == 0, endb = ib >= k.esc_end || ix.kpool[ib] == 0;
                if (!(enda && endb)) break; // the levels differ
                const bool lasta = ia >= p.esc_end, lastb = ib >= k.esc_end;
                // a trailing '#' is not a node in either key
                const bool hasha = level_is_hash(ix.kpool, a, (uint32_t)(ia - a), lasta), hashb = level_is_hash(ix.kpool, b, (uint32_t)(ib - b), lastb);
                if (hasha || hashb) break;
                shared++;
                if (lasta || lastb) break;
                a = ia + 1;
                b = ib + 1;
            }
        }
    }
    ob.flag[i] = new_tenant ? 1u : 0u;
    ob.nn[i] = nl - shared;
}
// after an inclusive scan of flag[] (tenant index + 1 per key): one lane per key notes its tenant; run heads record the run start
BMQ_HD void bulk_tenants_one(const DistIndexMut& ix, const OpBatch& ob, uint32_t i, const uint32_t* incl_scan) {
    const uint32_t t = incl_scan[i] - 1;
    if (i == 0 || incl_scan[i - 1] != incl_scan[i]) ob.bt_first[t] = i;
    if (i + 1 == ob.n) ix.bc->n_bulk_tenants = t + 1;
    ob.dir_slot[i] = t; // tenant index for now; locate maps it through bt_dir
}
// one lane per tenant: key count and exact node count of the run, from the inclusive scan of nn[] (no atomics: a run of 10 k keys
// bumping one counter cost 16 ms for 10 M keys)
BMQ_HD void bulk_counts_one(const OpBatch& ob, uint32_t t, uint32_t n_ten, const uint32_t* nn_incl) {
    const uint32_t first = ob.bt_first[t], next = t + 1 < n_ten ? ob.bt_first[t + 1] : ob.n;
    ob.bt_keys[t] = next - first;
    ob.bt_nodes[t] = nn_incl[next - 1] - (first ? nn_incl[first - 1] : 0u);
}

// ------------------------------------------------------------------------------------------------------------
// locate: the filter node of an op (created on the way for puts)
// ------------------------------------------------------------------------------------------------------------
// phase 0: the puts of the batch (filter nodes and dictionary tokens come into being), phase 1: its deletes (they find what exists --
// including what a put of the SAME batch in front of them created: one pass over both would let a delete's lane run ahead of the put's and
// drop the delete as "no such filter"; found by replaying merged mutation batches into the next generation, round 5), phase 2: every op
// (bulk loads and batches without deletes).
// Round 6 -- the same guarantee for half the price: phase 3 = EVERY op in one pass, a delete that does not find its filter is not dropped
// but marked TARGET_RETRY; phase 4 = the marked deletes once more, behind the kernel boundary that makes every put of the batch visible.
// A delete that found its filter in phase 3 found what phase 1 would have found (nodes are never removed); in the common batch -- unsubscribes
// of filters that exist -- phase 4 finds nothing to do (round 5: two full passes of ~60 us each over a 100 k-op batch).
constexpr unsigned long long TARGET_RETRY = ~0ull - 1;
BMQ_HD void locate_one(const DistIndexMut& ix, const OpBatch& ob, uint32_t i, uint32_t phase) {
    if (phase == 4) {
        if (ob.target[i] != TARGET_RETRY) return;
    } else if (phase < 2 && (!ob.op || ob.op[i] == 0) != (phase == 0)) return;
    ob.target[i] = TARGET_NONE;
    uint32_t d = ob.dir_slot[i];
    if (ob.bulk) d = ob.bt_dir[d];
    if (d == NONE) return; // unknown tenant: only deletes get here (puts made the host create the tenant first)
    KeyView k;
    if (!batch_key(ix, ob, i, k)) return;
    const bool is_put = !ob.op || ob.op[i] == 0;
    TenantSlot* ten = ix.tenants + d;
    const uint32_t base = ten->base, buckets = ten->buckets;
    uint32_t node = 0;                      // tenant root
    uint32_t* bloom = &ten->root_lit_bloom; // Bloom word of `node`
    unsigned long long slot_abs = 0;
    bool at_root = true, is_hash = false, ok = true;
    for_each_level(ix.kpool, k.esc, k.esc_end, [&](unsigned long long start, uint32_t len, const LevelHash& h, const uint32_t* inl, bool last) {
        if (level_is_hash(ix.kpool, start, len, last)) { // '#' is a wildcard only as the last level
            is_hash = true;
            return false;
        }
        uint32_t tok;
        if (level_is_plus(ix.kpool, start, len)) tok = TOK_PLUS;
        else {
            tok = dict_intern(ix, h, len, inl, ix.kpool, start, is_put);
            if (tok == TOK_UNKNOWN) { // delete of a filter with a level nobody ever used (or a failed insert: err is set)
                ok = false;
                return false;
            }
        }
        bool created;
        unsigned long long sa = 0;
        const uint32_t child = trie_child(ix, ten, base, buckets, node, at_root ? PARENT_IS_ROOT : slot_abs, tok, is_put, sa, created);
        if (child == NONE) {
            ok = false;
            return false;
        }
        if (created) {
            const uint32_t bit = tok == TOK_PLUS ? BLOOM_PLUS : (1u << bloom_bit(tok));
            if (!(atom_load(bloom) & bit)) atom_or(bloom, bit);
        }
        node = child;
        slot_abs = sa;
        bloom = &ix.trie[sa].lit_bloom;
        at_root = false;
        return true;
    });
    if (!ok) {
        if (phase == 3 && !is_put) ob.target[i] = TARGET_RETRY; // (a put of this batch may be creating the filter right now)
        return;
    }
    if (at_root && !is_hash) return; // cannot happen: a filter has at least one level
    // the filter "#" hangs off the tenant root (directory entry); every other filter off its node's slot
    ob.target[i] = at_root ? make_target(2, d) : make_target(is_hash ? 1 : 0, slot_abs);
    if (is_put) { // the key store entry of the new id (dropped again by the group step if the key turns out to be there)
        const uint32_t id = ob.id_base + (ob.put_rank ? ob.put_rank[i] : i);
        if (id < ix.id_cap) {
            ix.kref[id] = k.start | ((k.end - k.start) << KREF_LEN_SHIFT);
            ix.khash[id] = tail_hash(ix.kpool, k.tail, k.end);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// group: apply the ops of one filter's id set, in op order
// ------------------------------------------------------------------------------------------------------------
// The set is (begin, count): ids begin .. begin+count-1, or -- count & RANGE_INDIRECT -- the ascending id list
// route_pos[begin .. begin+count), whose capacity sits in route_pos[begin - 1].
struct IdSet {
    uint32_t begin, cf;
    BMQ_HD uint32_t count() const { return cf & ~RANGE_INDIRECT; }
    BMQ_HD bool indirect() const { return (cf & RANGE_INDIRECT) != 0; }
};
BMQ_HD uint32_t idset_at(const DistIndexMut& ix, const IdSet& s, uint32_t m) { return s.indirect() ? ix.route_pos[s.begin + m] : s.begin + m; }

// position of the member whose key tail equals key `k`'s, or NONE
// (Sixteen members per step: their ids and tail hashes are requested together -- one memory latency per sixteen members instead of two per
// member.  A group lane is alone on its chain of dependent reads, the builder's kernels run at 1-2 waves per SIMD, and a filter with a few
// hundred routes made its lane the one the whole k_b_group launch waited for.)
BMQ_HD uint32_t idset_find_key(const DistIndexMut& ix, const IdSet& s, const KeyView& k, uint32_t th) {
    const uint32_t n = s.count();
    const unsigned long long tl = k.end - k.tail;
    constexpr uint32_t STEP = 16;
    for (uint32_t m0 = 0; m0 < n; m0 += STEP) {
        uint32_t id[STEP], h[STEP];
#pragma unroll
        for (uint32_t j = 0; j < STEP; j++) id[j] = m0 + j < n ? idset_at(ix, s, m0 + j) : NONE;
#pragma unroll
        for (uint32_t j = 0; j < STEP; j++) h[j] = id[j] != NONE ? ix.khash[id[j]] : ~th;
#pragma unroll
        for (uint32_t j = 0; j < STEP; j++) {
            if (h[j] != th) continue;
            const unsigned long long r = ix.kref[id[j]];
            const unsigned long long off = r & KREF_OFF_MASK, len = r >> KREF_LEN_SHIFT;
            if (len < tl) continue;
            // same filter node => same tenant + levels; the keys are equal iff their tails are
            if (pool_bytes_equal(ix.kpool, off + len - tl, k.tail, tl)) return m0 + j;
        }
    }
    return NONE;
}
BMQ_HD uint32_t pow2_ceil(uint32_t v) {
    uint32_t p = 4;
    while (p < v) p <<= 1;
    return p;
}

// One lane per sorted position: heads (first op of a run of equal targets) apply their group.
// A group works on a PRIVATE copy of the set: local (begin, count) while the set is a plain range, and -- from the first op that
// needs an id list (non-contiguous put, removal from the middle, any change of an existing list) -- ONE freshly allocated list
// sized for everything the group can add.  Shared state (the node's begin/count, kref of removed ids) is written only at the end,
// so a group that finds the list pool full changes nothing, is counted in n_deferred and simply runs again after the host grew
// the pool.
BMQ_HD void group_one(const DistIndexMut& ix, const OpBatch& ob, uint32_t p) {
    const unsigned long long tg = ob.sorted_target[p];
    if (tg == TARGET_NONE) return;
    if (p > 0 && ob.sorted_target[p - 1] == tg) return; // not a head
    if (ob.group_done[p]) return;
    uint32_t e; // end of the group: upper bound of tg in the sorted array.  Groups are short (an op or two per filter): the next positions one by
                // one first -- a binary search over the whole batch is 17 dependent reads for a group of one
    {
        uint32_t lo = p + 1, hi = ob.n;
        for (uint32_t probe = 0; probe < 4 && lo < hi; probe++) {
            if (ob.sorted_target[lo] != tg) {
                hi = lo;
                break;
            }
            lo++;
        }
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (ob.sorted_target[mid] == tg) lo = mid + 1;
            else hi = mid;
        }
        e = lo;
    }
    const uint32_t kind = (uint32_t)(tg >> TARGET_KIND_SHIFT);
    const unsigned long long slot = tg & ((1ull << TARGET_KIND_SHIFT) - 1);
    uint32_t *pb, *pc;
    if (kind == 2) {
        pb = &ix.tenants[slot].root_hash_begin;
        pc = &ix.tenants[slot].root_hash_count;
    } else {
        TrieSlot& ts = ix.trie[slot];
        pb = kind ? &ts.hash_begin : &ts.own_begin;
        pc = kind ? &ts.hash_count : &ts.own_count;
    }
    const IdSet s0{*pb, *pc};
    IdSet s = s0;
    const uint32_t n_ops = e - p;
    if (ob.bulk) { // all puts into an empty set, ids ascending, no duplicates (strictly ascending keys)
        const uint32_t first = ob.id_base + ob.order[p], last = ob.id_base + ob.order[e - 1];
        if (last - first + 1 == n_ops) {
            s.begin = first;
            s.cf = n_ops;
        } else { // ids are not one contiguous range (SURVEY.md 8c quirk ii: keys of "x" interleave with keys of "x//...")
            const unsigned long long off = atom_add(&ix.bc->rp_used, (unsigned long long)n_ops + 1);
            if (off + n_ops + 1 > ix.rp_cap) {
                atom_add(&ix.bc->n_deferred, 1u);
                atom_add(&ix.bc->rp_need, (unsigned long long)n_ops + 1);
                return;
            }
            ix.route_pos[off] = n_ops;
            for (uint32_t q = 0; q < n_ops; q++) ix.route_pos[off + 1 + q] = ob.id_base + ob.order[p + q];
            s.begin = (uint32_t)(off + 1);
            s.cf = n_ops | RANGE_INDIRECT;
        }
        *pb = s.begin;
        *pc = s.cf;
        ob.group_done[p] = 1;
        return;
    }
    uint32_t n_put = 0;
    for (uint32_t q = p; q < e; q++) {
        n_put += (!ob.op || ob.op[ob.order[q]] == 0) ? 1u : 0u;
        ob.nn[ob.order[q]] = NONE; // scratch of this op: the id whose key store entry dies when the group commits
    }
    const uint32_t blk_cap = pow2_ceil(2 * (s0.count() + n_put));
    unsigned long long blk = 0; // private list block (header word at blk), allocated on first need
    bool failed = false;
    // make `s` a list inside the private block, holding its current members except position `skip`
    auto privatise = [&](uint32_t skip) -> bool {
        if (blk) return true;
        const unsigned long long off = atom_add(&ix.bc->rp_used, (unsigned long long)blk_cap + 1);
        if (off + blk_cap + 1 > ix.rp_cap) {
            atom_add(&ix.bc->n_deferred, 1u);
            atom_add(&ix.bc->rp_need, (unsigned long long)blk_cap + 1);
            failed = true;
            return false;
        }
        blk = off;
        const uint32_t n = s.count();
        uint32_t w = 0;
        for (uint32_t m = 0; m < n; m++)
            if (m != skip) ix.route_pos[off + 1 + w++] = idset_at(ix, s, m);
        ix.route_pos[off] = blk_cap;
        s.begin = (uint32_t)(off + 1);
        s.cf = w | RANGE_INDIRECT;
        return true;
    };
    uint32_t dups = 0, removed = 0, added = 0;
    for (uint32_t q = p; q < e && !failed; q++) {
        const uint32_t i = ob.order[q];
        KeyView k;
        if (!batch_key(ix, ob, i, k)) continue;
        const bool is_put = !ob.op || ob.op[i] == 0;
        const uint32_t th = tail_hash(ix.kpool, k.tail, k.end);
        const uint32_t m = idset_find_key(ix, s, k, th);
        const uint32_t n = s.count();
        if (is_put) {
            const uint32_t id = ob.id_base + (ob.put_rank ? ob.put_rank[i] : i);
            if (m != NONE) { // already there (a re-subscribe: same key, new incarnation in the value): the old id stays
                ob.nn[i] = id;
                dups++;
                continue;
            }
            if (blk == 0 && !s.indirect() && (n == 0 || id == s.begin + n)) { // still a plain range
                if (n == 0) s.begin = id;
                s.cf = n + 1;
            } else {
                if (!privatise(NONE)) break;
                ix.route_pos[s.begin + s.count()] = id; // ids only grow: the list stays ascending
                s.cf = (s.count() + 1) | RANGE_INDIRECT;
            }
            added++;
        } else if (m != NONE) {
            ob.nn[i] = idset_at(ix, s, m);
            if (blk == 0 && !s.indirect() && (m == 0 || m == n - 1)) { // an end of a plain range
                if (m == 0) s.begin++;
                s.cf = n - 1;
                if (s.cf == 0) s.begin = 0;
            } else if (blk == 0) {
                if (!privatise(m)) break;
            } else {
                for (uint32_t x = m; x + 1 < n; x++) ix.route_pos[s.begin + x] = ix.route_pos[s.begin + x + 1];
                s.cf = (n - 1) | RANGE_INDIRECT;
            }
            removed++;
        }
    }
    if (failed) return; // nothing shared was touched; the block request is on record (rp_need)
    if (s.indirect() && s.count() == 0) s = IdSet{0, 0}; // (its block becomes garbage below)
    // ---- commit ----
    unsigned long long garbage = 0;
    if (s0.indirect() && (blk != 0 || !s.indirect())) garbage += (unsigned long long)ix.route_pos[s0.begin - 1] + 1;
    if (blk != 0 && !s.indirect()) garbage += (unsigned long long)blk_cap + 1;
    for (uint32_t q = p; q < e; q++) {
        const uint32_t dead = ob.nn[ob.order[q]];
        if (dead != NONE && dead < ix.id_cap) {
            ix.kref[dead] = 0;
            if (ix.fo_dgroup && dead < ix.fo_cap) ix.fo_dgroup[dead] = FO_DEAD_ID;
        }
    }
    *pb = s.begin;
    *pc = s.cf;
    ob.group_done[p] = 1;
    if (garbage) atom_add(&ix.bc->rp_garbage, garbage);
    const uint32_t cl = p & (N_CTR_LANES - 1);
    if (dups) atom_add(&ix.bc->n_dups[cl], dups);
    if (removed) atom_add(&ix.bc->n_removed[cl], removed);
    if (added) atom_add(&ix.bc->n_added[cl], added);
    if (added != removed) {
        uint32_t d = ob.dir_slot[ob.order[p]];
        if (d != NONE) atom_add(&ix.tenants[d].n_routes, added - removed); // two's complement: may subtract
    }
}

// ------------------------------------------------------------------------------------------------------------
// region growth: re-insert one slot of the old region into the new one (node ids do not change)
// ------------------------------------------------------------------------------------------------------------
// Two passes over the old region (layout v3).  Pass 0: every literal edge, the root's '+' child and every '+' child that did NOT sit beside
// its parent, each to the first free slot from its hashed home on.  Pass 1: the '+' children that sat beside their parent (the other slot of
// their old line holds it): the parent -- placed by pass 0 -- is looked up in the new region and the child goes into the other slot of ITS
// line if that is free, else to its hashed home.  `d`: the tenant's directory slot (the root's '+' child reports its new place there).
BMQ_HD bool rehash_place(const DistIndexMut& ix, TrieSlot* d, const TrieSlot& src, uint64_t key) {
    if (atom_cas(reinterpret_cast<uint64_t*>(d), EDGE_EMPTY, key) != EDGE_EMPTY) return false;
    d->own_begin = src.own_begin;
    d->own_count = src.own_count;
    d->hash_begin = src.hash_begin;
    d->hash_count = src.hash_count;
    d->lit_bloom = src.lit_bloom;
    atom_publish(&d->node, src.node); // (pass 1 looks parents up while other lanes of pass 1 still place: a slot is readable once its id is)
    (void)ix;
    return true;
}
BMQ_HD void rehash_one(const DistIndexMut& ix, uint32_t old_base, uint32_t new_base, uint32_t new_buckets, uint32_t s, uint32_t pass, uint32_t d) {
    const TrieSlot src = ix.trie[(size_t)old_base + s];
    if (src.parent == NONE) return;
    const uint64_t key = (uint64_t)src.parent | ((uint64_t)src.token << 32);
    bool beside = false; // this is a '+' child and the other slot of its line holds its parent
    TrieSlot par{};
    if (src.token == TOK_PLUS && src.parent != 0) {
        par = ix.trie[(size_t)old_base + (s ^ 1u)];
        beside = par.parent != NONE && par.node == src.parent;
    }
    if ((pass == 1) != beside) return;
    if (beside) { // where did pass 0 put the parent?
        const uint64_t pkey = (uint64_t)par.parent | ((uint64_t)par.token << 32);
        uint32_t bk = edge_bucket(par.parent, par.token, new_buckets), j = 0;
        for (uint32_t probes = 0; probes < new_buckets;) {
            TrieSlot* q = ix.trie + (size_t)new_base + 2 * (size_t)bk + j;
            const uint64_t k = atom_load(reinterpret_cast<uint64_t*>(q));
            if (k == pkey) {
                if (rehash_place(ix, ix.trie + (size_t)new_base + ((2 * (size_t)bk + j) ^ 1u), src, key)) return;
                break; // the slot beside the parent is taken: hashed home
            }
            if (k == EDGE_EMPTY) break; // (cannot happen: pass 0 placed every parent)
            if (++j == 2) {
                j = 0;
                bk = (bk + 1 == new_buckets) ? 0 : bk + 1;
                probes++;
            }
        }
    }
    uint32_t bk = edge_bucket(src.parent, src.token, new_buckets), j = 0;
    for (uint32_t probes = 0; probes < new_buckets;) {
        if (rehash_place(ix, ix.trie + (size_t)new_base + 2 * (size_t)bk + j, src, key)) {
            if (src.token == TOK_PLUS && src.parent == 0) ix.tenants[d].root_plus = 2u * bk + j;
            return;
        }
        if (++j == 2) {
            j = 0;
            bk = (bk + 1 == new_buckets) ? 0 : bk + 1;
            probes++;
        }
    }
    atom_or(&ix.bc->err, (uint32_t)ERR_REGION_FULL);
}
constexpr TrieSlot FREE_SLOT{NONE, 0, 0, 0, 0, 0, NONE, 0};

// ------------------------------------------------------------------------------------------------------------
// read-only helpers (inspection: bmq_index_find, bmq_route_key)
// ------------------------------------------------------------------------------------------------------------
// exact lookup of (tenant, MQTT topic filter with '/' separators) -> the filter's id set; false if absent
BMQ_HD bool find_filter(const DistIndexMut& ix, const uint8_t* q, uint32_t tenant_len, uint32_t filter_len, IdSet& out) {
    out = IdSet{0, 0};
    const uint32_t d = tenant_find(ix.tenants, ix.tenant_mask, ix.tenant_names, q, 0, tenant_len);
    if (d == NONE) return false;
    TenantSlot* ten = ix.tenants + d;
    uint32_t node = 0;
    unsigned long long slot_abs = 0, pos = tenant_len;
    const unsigned long long end = (unsigned long long)tenant_len + filter_len;
    bool at_root = true;
    for (;;) {
        LevelHash h;
        uint32_t inl[4], len;
        const unsigned long long start = pos;
        scan_level_bytes<0x2F2F2F2Fu>(q, pos, end, h, inl, len);
        const bool last = pos >= end;
        if (level_is_hash(q, start, len, last)) {
            if (at_root) out = IdSet{ten->root_hash_begin, ten->root_hash_count};
            else out = IdSet{ix.trie[slot_abs].hash_begin, ix.trie[slot_abs].hash_count};
            return out.count() != 0;
        }
        const uint32_t tok = level_is_plus(q, start, len) ? TOK_PLUS : dict_intern(ix, h, len, inl, q, start, false);
        if (tok == TOK_UNKNOWN) return false;
        bool created;
        node = trie_child(ix, ten, ten->base, ten->buckets, node, at_root ? PARENT_IS_ROOT : slot_abs, tok, false, slot_abs, created);
        if (node == NONE) return false;
        at_root = false;
        if (last) break;
        pos++;
    }
    out = IdSet{ix.trie[slot_abs].own_begin, ix.trie[slot_abs].own_count};
    return out.count() != 0;
}

// ------------------------------------------------------------------------------------------------------------
// dictionary growth: re-insert one slot of the old table (tokens do not change)
// ------------------------------------------------------------------------------------------------------------
BMQ_HD void dict_rehash_one(const DictSlot* old, uint32_t i, const DistIndexMut& ix) {
    const DictSlot s = old[i];
    if (s.tag == 0) return;
    LevelHash h = level_hash_init(); // the slot hash is not stored: recompute it from the bytes
    for (uint32_t o = 0; o < s.len; o += 4) {
        uint32_t w;
        if (s.len <= 16) w = s.inl[o >> 2];
        else {
            w = 0;
            for (uint32_t k = 0; k < 4 && o + k < s.len; k++) w |= (uint32_t)ix.dpool[s.pool_off + o + k] << (8 * k);
        }
        level_hash_word(h, w);
    }
    uint32_t g = level_hash_slot(h, s.len) & ix.dict_group_mask, j = 0;
    for (uint32_t probes = 0; probes <= ix.dict_group_mask;) {
        DictSlot* d = ix.dict + DICT_GROUP * (size_t)g + j;
        if (atom_cas(&d->tag, 0u, s.tag) == 0u) {
            d->len = s.len;
            d->pool_off = s.pool_off;
            d->inl[0] = s.inl[0];
            d->inl[1] = s.inl[1];
            d->inl[2] = s.inl[2];
            d->inl[3] = s.inl[3];
            d->token = s.token;
            return;
        }
        if (++j == DICT_GROUP) {
            j = 0;
            g = (g + 1) & ix.dict_group_mask;
            probes++;
        }
    }
    atom_or(&ix.bc->err, (uint32_t)ERR_DICT_FULL);
}

// inspection: out[0] = number of ids of the filter, out[1 ..] = the first `cap` of them
BMQ_HD void find_copy(const DistIndexMut& ix, const uint8_t* q, uint32_t tenant_len, uint32_t filter_len, uint32_t* out, uint32_t cap) {
    IdSet s;
    if (!find_filter(ix, q, tenant_len, filter_len, s)) {
        out[0] = 0;
        return;
    }
    const uint32_t n = s.count();
    out[0] = n;
    for (uint32_t m = 0; m < n && m < cap; m++) out[1 + m] = idset_at(ix, s, m);
}
// id -> key store reference (0 for ids that are out of range or dead)
BMQ_HD void gather_ref_one(const DistIndexMut& ix, const uint32_t* ids, uint32_t i, uint32_t id_end, unsigned long long* out) {
    const uint32_t id = ids[i];
    out[i] = id < id_end && id < ix.id_cap ? ix.kref[id] : 0ull;
}
BMQ_HD void gather_bytes_one(const DistIndexMut& ix, const unsigned long long* refs, const uint64_t* offs, uint32_t i, uint8_t* out) {
    const unsigned long long r = refs[i];
    const unsigned long long off = r & KREF_OFF_MASK, len = r >> KREF_LEN_SHIFT;
    for (unsigned long long k = 0; k < len; k++) out[offs[i] + k] = ix.kpool[off + k];
}

} // namespace bmq
