// bmq_index.h -- host-side builder of the HBM-resident dist index (pure C++17, no HIP).
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

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

// ---- packed, sorted, de-duplicated key set ------------------------------------------------------------------
struct KeySet {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off{0};
    size_t size() const { return off.size() - 1; }
    std::string_view key(size_t i) const {
        return std::string_view((const char*)bytes.data() + off[i], (size_t)(off[i + 1] - off[i]));
    }
    // replace contents with the given keys (any order); sorts (parallel) + uniques
    void assign(const uint8_t* keys, const uint32_t* key_off, uint32_t n);
    // apply puts/deletes in order (op 0 put, 1 delete)
    void apply(const uint8_t* keys, const uint32_t* key_off, const uint8_t* op, uint32_t n);
    // rank of key or -1
    int64_t find(std::string_view k) const;
};

// ---- flattened dist index (host copy of what is uploaded to HBM) -----------------------------------------------
struct DistIndexHost {
    std::vector<TrieSlot> trie;
    std::vector<TenantSlot> tenants;
    std::vector<DictSlot> dict;
    std::vector<uint8_t> pool;
    std::vector<uint32_t> route_pos;
    uint64_t n_routes = 0, n_tenants = 0, n_nodes = 0, n_tokens = 0;
    std::string error;

    // Build from a sorted unique key set.  Returns false (error set) on a malformed key.
    bool build(const KeySet& ks);

    // host-side exact helpers (inspection only -- never used for matching)
    uint32_t find_token(std::string_view level) const;                 // TOK_UNKNOWN if absent
    const TenantSlot* find_tenant(uint32_t token) const;               // nullptr if absent
    uint32_t find_child(const TenantSlot& region, uint32_t parent_slot, uint32_t token) const; // slot or NONE
    uint32_t find_filter_node(std::string_view tenant, std::string_view mqtt_filter, bool& is_hash) const;
};

} // namespace bmq
