// bmq_codec.cpp -- route-key codec on the host (see bmq_codec.h).
#include "bmq_codec.h"

#include <cstring>
#include <vector>

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


// ------------------------------------------------------------------------------------------------------------
// retain store key schema
// ------------------------------------------------------------------------------------------------------------
namespace {
template <class F> void for_each_utf16_unit(std::string_view s, F&& f) { // same decoding as java_string_hash
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
            cp = ((c & 0x07) << 18) | (((uint8_t)s[i + 1] & 0x3F) << 12) | (((uint8_t)s[i + 2] & 0x3F) << 6) | ((uint8_t)s[i + 3] & 0x3F);
            i += 4;
        } else {
            cp = 0xFFFD;
            i += 1;
        }
        if (cp >= 0x10000) {
            cp -= 0x10000;
            f(0xD800u + (cp >> 10));
            f(0xDC00u + (cp & 0x3FF));
        } else f(cp);
    }
}
std::vector<std::string_view> split_levels(std::string_view s) { // TopicUtil.parse(topic, false): empty levels kept
    std::vector<std::string_view> out;
    size_t b = 0;
    for (size_t i = 0; i <= s.size(); i++)
        if (i == s.size() || s[i] == '/') {
            out.push_back(s.substr(b, i - b));
            b = i + 1;
        }
    return out;
}
void tenant_begin_key(std::string& k, std::string_view tenant) { // 0x00 | u16be(len) | tenant
    k.push_back('\0');
    put_u16be(k, tenant.size());
    k.append(tenant);
}
} // namespace

uint8_t retain_level_hash_byte(std::string_view level) {
    uint32_t h = 0x811C9DC5u;
    for_each_utf16_unit(level, [&](uint32_t u) {
        h ^= u;
        h *= 0x01000193u;
    });
    return (uint8_t)(h & 0xFF);
}

std::string retain_message_key(std::string_view tenant, std::string_view topic) {
    const auto levels = split_levels(topic);
    std::string k;
    tenant_begin_key(k, tenant);
    put_u16be(k, levels.size());
    for (auto lv : levels) k.push_back((char)retain_level_hash_byte(lv));
    for (char c : topic) k.push_back(c == '/' ? '\0' : c); // TopicUtil.escape
    return k;
}

RetainFilterRoute retain_filter_route(std::string_view tenant, std::string_view filter) {
    RetainFilterRoute r;
    const auto levels = split_levels(filter);
    size_t first_wild = levels.size();
    for (size_t i = 0; i < levels.size(); i++)
        if (levels[i] == "+") {
            first_wild = i;
            break;
        }
    r.multi = !levels.empty() && levels.back() == "#";
    if (first_wild == levels.size() && r.multi) first_wild = levels.size() - 1; // KVSchemaUtil.filterPrefix
    r.wildcard = first_wild < levels.size();
    if (!r.wildcard) { // a plain topic: the exact key (MatchCallRangeRouter.java:91-95)
        r.key_prefix = retain_message_key(tenant, filter);
        r.levels = (uint16_t)levels.size();
        for (auto lv : levels) r.level_hash.push_back((char)retain_level_hash_byte(lv));
        return r;
    }
    r.levels = (uint16_t)(r.multi ? levels.size() - 1 : levels.size());
    for (size_t i = 0; i < first_wild; i++) r.level_hash.push_back((char)retain_level_hash_byte(levels[i]));
    tenant_begin_key(r.key_prefix, tenant);
    put_u16be(r.key_prefix, r.levels);
    r.key_prefix.append(r.level_hash);
    return r;
}

} // namespace bmq
