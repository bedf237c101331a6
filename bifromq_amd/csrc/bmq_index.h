// bmq_index.h -- host-side owner of the dist index: the route keys (per tenant) and the image of what lives in HBM.
// Pure C++17, no HIP.  Everything is organised PER TENANT -- key set, trie region, route-id base -- so that a batch of
// subscribe/unsubscribe operations only rebuilds and re-uploads the regions of the tenants it touches
// (DW/DistWorkerCoProc.java:188-209 pushes Add/RemoveRoutesTask per tenant in the same way).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "bmq_dict.h"
#include "bmq_layout.h"

namespace bmq {

// ---- route-key codec: SCHEMA/KVSchemaUtil.java:91-130, SCHEMA/cache/RouteDetailCache.java:53-117 --------
struct RouteKeyParts {
    std::string_view tenant;
    std::string_view esc_filter; // levels joined by NUL (no trailing NUL)
    std::string_view receiver;   // receiverUrl (flag 1) or group name (flag 2/3)
    uint8_t bucket = 0;
    uint8_t flag = 0;
};
bool decode_route_key(std::string_view key, RouteKeyParts& out);
std::string encode_route_key(std::string_view tenant, std::string_view mqtt_filter_no_share, uint8_t flag,
                             std::string_view receiver);
int32_t java_string_hash(std::string_view utf8);

// ---- packed, sorted, de-duplicated key set (one per tenant) -------------------------------------------------
struct KeySet {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off{0};
    size_t size() const { return off.size() - 1; }
    std::string_view key(size_t i) const {
        return std::string_view((const char*)bytes.data() + off[i], (size_t)(off[i + 1] - off[i]));
    }
    void assign(std::vector<std::string_view>& keys);                       // any order; sorts + uniques
    // in order; op 0 put, 1 delete.  src (optional): for every rank of the result, the rank the key had before (>= 0) or
    // -(index into ops) - 1 for a key the batch put
    void apply(std::vector<std::pair<std::string_view, uint8_t>>& ops, std::vector<int64_t>* src = nullptr);
    int64_t find(std::string_view k) const;                                 // rank or -1
};

struct TenantState {
    std::string name;
    KeySet keys;
    uint32_t token = 0;      // dictionary token of the tenant id
    uint32_t base = 0;       // region: first slot in the global table
    uint32_t cap_slots = 0;  // slots reserved for the region (>= 2 * buckets)
    uint32_t buckets = 0;
    uint32_t root_rel = 0;   // root slot, relative to base
    uint32_t n_nodes = 0;
    uint32_t rank_base = 0;  // global id of the tenant's first route
    uint32_t rp_base = 0;    // first entry of the tenant in route_pos
    std::vector<uint32_t> indirect; // tenant-relative ids of nodes whose ids are not one contiguous range
    // EXPERIMENTAL, off unless BMQ_INCREMENTAL=1: the tenant's trie kept between applies (opaque, see bmq_index.cpp), so that a
    // batch of mutations costs O(ops * depth + nodes) on the host instead of a re-parse of all the tenant's keys
    std::shared_ptr<void> inc;
};

struct TenantOrder { // tenants in KV key order: u16be(len) then bytes (SCHEMA/KVSchemaUtil.java:91-94)
    bool operator()(const std::string& a, const std::string& b) const {
        if (a.size() != b.size()) return a.size() < b.size();
        return a < b;
    }
};

// Zero-on-demand slot array (calloc): growing the table must not cost a pass over gigabytes that no region owns yet.
struct SlotBuf {
    TrieSlot* p = nullptr;
    size_t n = 0;
    void (*on_release)(void* ctx, void* p) = nullptr; // called before the memory goes away (the engine un-pins it)
    void* release_ctx = nullptr;
    SlotBuf() = default;
    SlotBuf(const SlotBuf&) = delete;
    SlotBuf& operator=(const SlotBuf&) = delete;
    ~SlotBuf() { release(); }
    void release() {
        if (p && on_release) on_release(release_ctx, p);
        free(p);
        p = nullptr;
        n = 0;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    TrieSlot* data() { return p; }
    const TrieSlot* data() const { return p; }
    TrieSlot& operator[](size_t i) { return p[i]; }
    const TrieSlot& operator[](size_t i) const { return p[i]; }
    void clear() { release(); }
    bool grow(size_t m, size_t keep) { // contents of [0, keep) survive
        TrieSlot* q = (TrieSlot*)aligned_alloc_zero(m);
        if (!q) return false;
        if (keep) memcpy(q, p, keep * sizeof(TrieSlot));
        release();
        p = q;
        n = m;
        return true;
    }
    static void* aligned_alloc_zero(size_t m) { return calloc(m, sizeof(TrieSlot)); } // calloc of 32-byte items is >= 16-byte aligned; host side needs no more
};

struct DistIndexHost {
    // ---- image of the device arrays ----
    SlotBuf trie;                    // regions at their bases; slot indices inside a region are region-relative
    std::vector<TenantSlot> tenants; // directory
    std::vector<DictSlot> dict;
    std::vector<uint8_t> pool;
    std::vector<uint32_t> route_pos; // absolute ids
    // ---- what changed since the last upload ----
    bool full_upload = true;
    bool dict_changed = true;
    std::vector<std::pair<uint32_t, uint32_t>> dirty; // (base, slots) regions rewritten in place
    // ---- bookkeeping ----
    std::map<std::string, std::unique_ptr<TenantState>, TenantOrder> by_name;
    std::vector<TenantState*> order; // by rank_base
    HostDict dict_h;
    std::deque<std::string> strings; // owned level strings of the dictionary
    uint32_t next_free = 0;          // first unassigned slot of the table
    uint64_t n_routes = 0, n_nodes = 0;
    std::string error;

    bool rebuild(const uint8_t* keys, const uint32_t* key_off, uint32_t n);
    bool apply(const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n);
    std::string_view route_key(uint32_t id) const; // empty view if out of range
    // exact lookup (inspection only -- never used for matching): ids stored under (tenant, filter)
    std::vector<uint32_t> find_filter(std::string_view tenant, std::string_view mqtt_filter) const;
    uint64_t n_tenants() const { return by_name.size(); }
    uint64_t n_tokens() const { return dict_h.entries.size(); }

private:
    bool refresh(std::vector<TenantState*>& touched);
    uint32_t find_child(const TenantState& t, uint32_t parent_rel, uint32_t token) const;
};

} // namespace bmq
