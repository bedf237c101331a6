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

} // namespace bmq
