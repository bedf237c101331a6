// bmq_codec.cpp -- route-key codec on the host (see bmq_codec.h).
#include "bmq_codec.h"

#include <cstring>

namespace bmq {

// ------------------------------------------------------------------------------------------------------------
// codec
// ------------------------------------------------------------------------------------------------------------
// Java String.hashCode over the UTF-16 code units of a UTF-8 string (used for the bucket byte,
// SCHEMA/KVSchemaUtil.java:127-130).
int32_t java_string_hash(std::string_view s) {
    uint32_t h = 0;
    const size_t n = s.size();
    size_t i = 0;
    while (i < n) {
        const uint32_t c = (uint8_t)s[i];
        uint32_t cp;
        if (c < 0x80) {
            cp = c;
            i += 1;
        } else if ((c & 0xE0) == 0xC0 && i + 1 < n) {
            cp = ((c & 0x1F) << 6) | ((uint8_t)s[i + 1] & 0x3F);
            i += 2;
        } else if ((c & 0xF0) == 0xE0 && i + 2 < n) {
            cp = ((c & 0x0F) << 12) | (((uint8_t)s[i + 1] & 0x3F) << 6) | ((uint8_t)s[i + 2] & 0x3F);
            i += 3;
        } else if ((c & 0xF8) == 0xF0 && i + 3 < n) {
            cp = ((c & 0x07) << 18) | (((uint8_t)s[i + 1] & 0x3F) << 12) | (((uint8_t)s[i + 2] & 0x3F) << 6) |
                 ((uint8_t)s[i + 3] & 0x3F);
            i += 4;
        } else {
            cp = 0xFFFD;
            i += 1;
        }
        if (cp >= 0x10000) { // surrogate pair
            cp -= 0x10000;
            h = 31u * h + (0xD800u + (cp >> 10));
            h = 31u * h + (0xDC00u + (cp & 0x3FF));
        } else {
            h = 31u * h + cp;
        }
    }
    return (int32_t)h;
}

static inline void put_u16be(std::string& s, size_t v) {
    s.push_back((char)((v >> 8) & 0xFF));
    s.push_back((char)(v & 0xFF));
}

// key = 0x00 | u16be(len tenant) | tenant | (level 0x00)* | 0x00 | bucket | flag | receiver | u16be(len receiver)
std::string encode_route_key(std::string_view tenant, std::string_view filter, uint8_t flag,
                             std::string_view receiver) {
    std::string k;
    k.reserve(tenant.size() + filter.size() + receiver.size() + 10);
    k.push_back('\0');
    put_u16be(k, tenant.size());
    k.append(tenant);
    for (char c : filter) k.push_back(c == '/' ? '\0' : c); // TopicUtil.escape: '/' -> NUL
    k.push_back('\0');                                       // terminates the last level
    k.push_back('\0');                                       // end of filter
    const uint32_t h = (uint32_t)java_string_hash(receiver);
    k.push_back((char)((h ^ (h >> 16)) & 0xFF));
    k.push_back((char)flag);
    k.append(receiver);
    put_u16be(k, receiver.size());
    return k;
}

// Parsed from both ends like RouteDetailCache.java:53-109.
bool decode_route_key(std::string_view k, RouteKeyParts& out) {
    if (k.size() < 3 + 2 + 2 + 2 || k[0] != 0) return false;
    const size_t tlen = ((size_t)(uint8_t)k[1] << 8) | (uint8_t)k[2];
    const size_t esc_start = 3 + tlen;
    const size_t rlen = ((size_t)(uint8_t)k[k.size() - 2] << 8) | (uint8_t)k[k.size() - 1];
    if (k.size() < esc_start + 4 + rlen + 2) return false;
    const size_t recv_start = k.size() - 2 - rlen;
    const size_t esc_end = recv_start - 4; // level-terminating NUL, filter-terminating NUL, bucket, flag
    if (k[esc_end] != 0 || k[esc_end + 1] != 0) return false;
    out.tenant = k.substr(3, tlen);
    out.esc_filter = k.substr(esc_start, esc_end - esc_start);
    out.bucket = (uint8_t)k[recv_start - 2];
    out.flag = (uint8_t)k[recv_start - 1];
    out.receiver = k.substr(recv_start, rlen);
    return out.flag >= 1 && out.flag <= 3;
}

} // namespace bmq
