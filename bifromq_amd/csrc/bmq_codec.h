// bmq_codec.h -- route-key codec on the host: SCHEMA/KVSchemaUtil.java:91-130, SCHEMA/cache/RouteDetailCache.java:53-117.
// (The builder kernels parse the same layout on the device: bmq_build_core.h::key_parse.)
#pragma once
#include <cstdint>
#include <string>
#include <string_view>

namespace bmq {

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

// ---- retain store key schema: bifromq-retain/bifromq-retain-store-schema/src/main/java/org/apache/bifromq/retain/
// store/schema/KVSchemaUtil.java:44-73, LevelHash.java:30-50 ----
// one byte per level: FNV-1a 32 over the level's UTF-16 code units, lowest byte
uint8_t retain_level_hash_byte(std::string_view level_utf8);
// retainMessageKey(tenantId, topic) = 0x00 | u16be(len tenant) | tenant | u16be(#levels) | LevelHash(levels) | escape(topic)
std::string retain_message_key(std::string_view tenant, std::string_view topic);
// retainKeyPrefix(tenantId, levels, filterPrefix(parse(filter))): `levels` = level count of the filter (without a trailing '#');
// the prefix = the levels in front of the first wildcard.  Returned with the pieces MatchCallRangeRouter needs
// (bifromq-retain-server/.../scheduler/MatchCallRangeRouter.java:60-134).
struct RetainFilterRoute {
    std::string key_prefix;  // retainKeyPrefix(...), or retainMessageKey for a filter without wildcards
    std::string level_hash;  // LevelHash.hash(filterPrefix)
    uint16_t levels = 0;
    bool wildcard = false;    // the filter contains '+' or '#'
    bool multi = false;       // ... ends with '#': matches any number of further levels
};
RetainFilterRoute retain_filter_route(std::string_view tenant, std::string_view topic_filter);

} // namespace bmq
