// bmq_router.cpp -- the KV range router arithmetic either side of the match kernels, host C++:
//   KVRangeRouterUtil.findByKey / findByBoundary   base-kv/base-kv-store-client/src/main/java/org/apache/bifromq/basekv/client/
//                                                  KVRangeRouterUtil.java:41-103
//   BoundaryUtil.compare* / inRange / upperBound   base-kv/base-kv-type-proto/src/main/java/org/apache/bifromq/basekv/utils/
//                                                  BoundaryUtil.java:122-195,241-252,299-339
//   MatchCallRangeRouter.rangeLookup               bifromq-retain/bifromq-retain-server/src/main/java/org/apache/bifromq/retain/server/
//                                                  scheduler/MatchCallRangeRouter.java:56-134
// The reference keeps the router in a TreeMap<Boundary, KVRangeSetting> ordered by BoundaryUtil.compare; here it is the caller's array
// of boundaries in that order and every NavigableMap operation becomes a binary search, so a lookup returns an index interval.
#include "../../include/bmq.h"
#include "bmq_codec.h"

#include <cstring>
#include <memory>
#include <optional>
#include <string>
#include <string_view>
#include <vector>

namespace bmq {
namespace {

using Key = std::optional<std::string_view>; // nullopt = the side is open (Boundary.hasStartKey / hasEndKey false)

int cmp_bytes(std::string_view a, std::string_view b) { // ByteString.unsignedLexicographicalComparator
    const size_t n = a.size() < b.size() ? a.size() : b.size();
    const int c = n ? memcmp(a.data(), b.data(), n) : 0;
    if (c) return c < 0 ? -1 : 1;
    return a.size() < b.size() ? -1 : a.size() > b.size() ? 1 : 0;
}
int cmp_start(const Key& a, const Key& b) { // BoundaryUtil.compareStartKey: an open start is the smallest
    if (!a && !b) return 0;
    if (!a) return -1;
    if (!b) return 1;
    return cmp_bytes(*a, *b);
}
int cmp_end(const Key& a, const Key& b) { // BoundaryUtil.compareEndKeys: an open end is the greatest
    if (!a && !b) return 0;
    if (!a) return 1;
    if (!b) return -1;
    return cmp_bytes(*a, *b);
}
struct Bnd {
    Key start, end;
};
int cmp_bnd(const Bnd& a, const Bnd& b) { // BoundaryUtil.compare(Boundary, Boundary)
    const int c = cmp_start(a.start, b.start);
    return c ? c : cmp_end(a.end, b.end);
}
std::optional<std::string> upper_bound_of(std::string_view key) { // BoundaryUtil.upperBound: nullopt = open end
    size_t i = key.size();
    while (i > 0 && (uint8_t)key[i - 1] == 0xFF) i--;
    if (i == 0) return std::nullopt;
    std::string u(key.substr(0, i));
    u[i - 1] = (char)((uint8_t)u[i - 1] + 1);
    return u;
}
bool in_range(std::string_view key, const Bnd& b) { // BoundaryUtil.inRange(key, boundary)
    if (b.start && cmp_bytes(key, *b.start) < 0) return false;
    if (b.end) return cmp_bytes(key, *b.end) < 0;
    return true;
}

struct Router {
    std::vector<Bnd> b; // ascending by cmp_bnd (checked)
    // number of boundaries < q, resp. <= q
    size_t lower(const Bnd& q) const {
        size_t lo = 0, hi = b.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (cmp_bnd(b[mid], q) < 0) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
    size_t upper(const Bnd& q) const {
        size_t lo = 0, hi = b.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (cmp_bnd(b[mid], q) <= 0) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
    // KVRangeRouterUtil.findByKey (:41-52): floorEntry(Boundary{startKey = key}) then inRange
    long find_by_key(std::string_view key) const {
        const Bnd q{key, std::nullopt};
        const size_t u = upper(q);
        if (u == 0) return -1;
        return in_range(key, b[u - 1]) ? (long)(u - 1) : -1;
    }
    // KVRangeRouterUtil.findByBoundary (:54-103) -> the index interval [lo, hi) of the boundaries it returns
    void find_by_boundary(const Bnd& q, size_t& lo, size_t& hi) const {
        lo = hi = 0;
        if (b.empty()) return;
        if (!q.start && !q.end) { // FULL_BOUNDARY
            hi = b.size();
            return;
        }
        if (!q.start) { // (null, endKey): headMap(Boundary{endKey, endKey}, false)
            hi = lower(Bnd{q.end, q.end});
            return;
        }
        const Bnd qs{q.start, q.start};
        size_t from = upper(qs); // floorKey(boundaryStart) = b[from - 1]; null -> firstKey
        from = from ? from - 1 : 0;
        const bool include = cmp_end(b[from].end, q.start) > 0;
        lo = include ? from : from + 1;
        if (!q.end) { // [startKey, null): tailMap(floor, include)
            hi = b.size();
            return;
        }
        hi = lower(Bnd{q.end, q.end}); // subMap(floor, include, Boundary{endKey, endKey}, false)
        if (hi < lo) hi = lo;          // fromKey > toKey: TreeMap.subMap throws IllegalArgumentException there; here: nothing
    }
};

int load_router(Router& r, const uint8_t* flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end, const uint32_t* end_off,
                uint32_t n) {
    if (n && (!flags || !start_off || !end_off)) return BMQ_E_INVAL;
    // offsets: start at 0 and never decrease, so that every view below lies inside [0, off[n]) of its byte array (the router object
    // copies exactly off[n] bytes: a decreasing or oversized offset would build views outside the copy and underflow their lengths)
    if (n && (start_off[0] != 0 || end_off[0] != 0)) return BMQ_E_INVAL;
    for (uint32_t i = 0; i < n; i++)
        if (start_off[i] > start_off[i + 1] || end_off[i] > end_off[i + 1]) return BMQ_E_INVAL;
    if (n && ((start_off[n] && !start) || (end_off[n] && !end))) return BMQ_E_INVAL;
    r.b.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        if (flags[i] & 1) r.b[i].start = std::string_view((const char*)start + start_off[i], start_off[i + 1] - start_off[i]);
        if (flags[i] & 2) r.b[i].end = std::string_view((const char*)end + end_off[i], end_off[i + 1] - end_off[i]);
        if (i && cmp_bnd(r.b[i - 1], r.b[i]) >= 0) return BMQ_E_INVAL; // not a TreeMap key order
    }
    return BMQ_OK;
}

// KVSchemaUtil.parseLevelHash of the retain schema (KVSchemaUtil.java:79-85); false where ByteString.substring would throw
bool parse_level_hash(std::string_view key, std::string_view& out) {
    if (key.size() < 3) return false;
    const size_t tl = ((size_t)(uint8_t)key[1] << 8) | (uint8_t)key[2];
    if (tl >= 0x8000) return false; // toShort: negative
    const size_t lv_idx = 3 + tl, hash_idx = lv_idx + 2;
    if (hash_idx > key.size()) return false;
    const size_t levels = ((size_t)(uint8_t)key[lv_idx] << 8) | (uint8_t)key[lv_idx + 1];
    if (levels >= 0x8000 || hash_idx + levels > key.size()) return false;
    out = key.substr(hash_idx, levels);
    return true;
}

} // namespace
} // namespace bmq

using namespace bmq;

extern "C" {

int bmq_router_find_by_key(const uint8_t* range_flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end,
                           const uint32_t* end_off, uint32_t n_ranges, const uint8_t* key, uint32_t key_len, int32_t* out_index) {
    if (!out_index) return BMQ_E_INVAL;
    Router r;
    if (int rc = load_router(r, range_flags, start, start_off, end, end_off, n_ranges)) return rc;
    *out_index = (int32_t)r.find_by_key(std::string_view((const char*)key, key_len));
    return BMQ_OK;
}

int bmq_router_find_by_boundary(const uint8_t* range_flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end,
                                const uint32_t* end_off, uint32_t n_ranges, uint8_t query_flags, const uint8_t* q_start, uint32_t q_start_len,
                                const uint8_t* q_end, uint32_t q_end_len, uint32_t* out_first, uint32_t* out_count) {
    if (!out_first || !out_count) return BMQ_E_INVAL;
    if (((query_flags & 1) && q_start_len && !q_start) || ((query_flags & 2) && q_end_len && !q_end)) return BMQ_E_INVAL;
    Router r;
    if (int rc = load_router(r, range_flags, start, start_off, end, end_off, n_ranges)) return rc;
    Bnd q;
    if (query_flags & 1) q.start = std::string_view((const char*)q_start, q_start_len);
    if (query_flags & 2) q.end = std::string_view((const char*)q_end, q_end_len);
    size_t lo, hi;
    r.find_by_boundary(q, lo, hi);
    *out_first = (uint32_t)lo;
    *out_count = (uint32_t)(hi - lo);
    return BMQ_OK;
}

static int retain_range_lookup(const Router& r, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filters, const uint32_t* filter_off,
                               uint32_t n_filters, uint32_t mode, uint8_t* out_keep) {
    const uint32_t n_ranges = (uint32_t)r.b.size();
    if (mode > BMQ_ROUTER_EXACT) return BMQ_E_INVAL;
    if ((n_filters && (!filters || !filter_off)) || (n_filters && n_ranges && !out_keep)) return BMQ_E_INVAL;
    const std::string_view tn((const char*)tenant, tenant_len);
    std::string tenant_begin; // KVSchemaUtil.tenantBeginKey
    tenant_begin.push_back('\0');
    tenant_begin.push_back((char)(tenant_len >> 8));
    tenant_begin.push_back((char)(tenant_len & 0xFF));
    tenant_begin.append(tn);
    const std::optional<std::string> tenant_end = upper_bound_of(tenant_begin); // never open: the key starts with 0x00
    if (n_ranges) memset(out_keep, 0, (size_t)n_filters * n_ranges);
    for (uint32_t f = 0; f < n_filters; f++) {
        const std::string_view filter((const char*)filters + filter_off[f], filter_off[f + 1] - filter_off[f]);
        const RetainFilterRoute fr = retain_filter_route(tn, filter);
        uint8_t* keep = out_keep + (size_t)f * n_ranges;
        if (!fr.wildcard) { // MatchCallRangeRouter.java:91-95 (the reference asserts that a range holds the key)
            const long i = r.find_by_key(fr.key_prefix);
            if (i < 0) return BMQ_E_INVAL;
            keep[i] = 1;
            continue;
        }
        size_t lo, hi;
        Bnd q;
        q.start = fr.key_prefix;
        if (!fr.multi) { // fixed number of levels (:68-71): [prefix, upperBound(prefix))
            const std::optional<std::string> ub = upper_bound_of(fr.key_prefix);
            if (ub) q.end = *ub;
            r.find_by_boundary(q, lo, hi);
            for (size_t i = lo; i < hi; i++) keep[i] = 1;
            continue;
        }
        q.end = *tenant_end; // '#' at the end: [prefix, upperBound(tenantBeginKey)) (:73-85, :96-105)
        r.find_by_boundary(q, lo, hi);
        if (fr.level_hash.empty()) { // the filter starts with a wildcard
            for (size_t i = lo; i < hi; i++) keep[i] = 1;
            continue;
        }
        if (mode == BMQ_ROUTER_EXACT) {
            // Keys order by (level count, LevelHash, escaped topic): for every level count L >= levels the keys the filter can match
            // are exactly the interval [tenant | L | hash, tenant | L | upperBound(hash)).  A range is needed iff it meets one of them.
            const std::optional<std::string> hash_ub = upper_bound_of(fr.level_hash);
            auto level_count_of = [&](std::string_view k, uint32_t& out) { // level count field of a key inside the tenant's key space
                if (k.size() < tenant_begin.size() + 2) return false;
                out = ((uint32_t)(uint8_t)k[tenant_begin.size()] << 8) | (uint8_t)k[tenant_begin.size() + 1];
                return true;
            };
            auto meets = [&](const Bnd& c, uint32_t L) {
                std::string lo = tenant_begin;
                lo.push_back((char)(L >> 8));
                lo.push_back((char)(L & 0xFF));
                std::string hi = lo;
                lo.append(fr.level_hash);
                Key hi_key;
                std::optional<std::string> hi_s;
                if (hash_ub) hi_s = hi + *hash_ub;
                else hi_s = upper_bound_of(hi); // every hash byte 0xFF: up to the end of this level count
                if (hi_s) hi_key = *hi_s;
                // [lo, hi) meets [c.start, c.end)  <=>  max(starts) < min(ends)
                const Key s = cmp_start(c.start, Key(lo)) > 0 ? c.start : Key(lo);
                const Key e = cmp_end(c.end, hi_key) < 0 ? c.end : hi_key;
                return !e || cmp_bytes(*s, *e) < 0;
            };
            for (size_t i = lo; i < hi; i++) {
                const Bnd& c = r.b[i];
                uint32_t l_lo = fr.levels, l_hi = 0xFFFF, l;
                bool known = true;
                if (c.start && cmp_start(c.start, q.start) > 0) {
                    if (level_count_of(*c.start, l)) l_lo = l > l_lo ? l : l_lo;
                    else known = false;
                }
                if (c.end && cmp_end(c.end, q.end) < 0) {
                    if (level_count_of(*c.end, l)) l_hi = l;
                    else known = false;
                }
                if (!known || l_hi > l_lo + 1 || meets(c, l_lo) || (l_hi > l_lo && meets(c, l_hi))) keep[i] = 1;
            }
            continue;
        }
        // findCandidates (:96-133): drop the ranges that lie completely between two of the per-level-count key prefixes
        const std::optional<std::string> hash_ub = upper_bound_of(fr.level_hash); // open when every hash byte is 0xFF (the reference
                                                                                  // then fails with a NullPointerException)
        for (size_t i = lo; i < hi; i++) {
            const Bnd& c = r.b[i];
            std::string_view h;
            if (c.start && cmp_start(c.start, q.start) > 0) {
                if (!parse_level_hash(*c.start, h)) return BMQ_E_INVAL;
                if (hash_ub && cmp_bytes(*hash_ub, h) <= 0) continue;
            }
            if (c.end && cmp_end(c.end, q.end) <= 0) {
                if (!parse_level_hash(*c.end, h)) return BMQ_E_INVAL;
                if (cmp_bytes(h, fr.level_hash) <= 0) continue;
            }
            keep[i] = 1;
        }
    }
    return BMQ_OK;
}

int bmq_retain_range_lookup(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filters, const uint32_t* filter_off, uint32_t n_filters,
                            const uint8_t* range_flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end,
                            const uint32_t* end_off, uint32_t n_ranges, uint32_t mode, uint8_t* out_keep) {
    Router r;
    if (int rc = load_router(r, range_flags, start, start_off, end, end_off, n_ranges)) return rc;
    return retain_range_lookup(r, tenant, tenant_len, filters, filter_off, n_filters, mode, out_keep);
}

// ---- the router as an object: the boundaries are copied, checked and indexed ONCE (the reference keeps its TreeMap between calls too:
// KVRangeRouter is rebuilt only when the range landscape changes, base-kv/base-kv-store-client/.../KVRangeRouter.java) ----
struct bmq_router {
    std::string start_bytes, end_bytes; // own copies: the views of `r` point into them
    Router r;
};

int bmq_router_create(const uint8_t* range_flags, const uint8_t* start, const uint32_t* start_off, const uint8_t* end, const uint32_t* end_off,
                      uint32_t n_ranges, bmq_router** out) {
    if (!out) return BMQ_E_INVAL;
    *out = nullptr;
    if (n_ranges && (!range_flags || !start_off || !end_off)) return BMQ_E_INVAL;
    auto h = std::make_unique<bmq_router>();
    if (n_ranges) {
        if (start_off[n_ranges] && !start) return BMQ_E_INVAL;
        if (end_off[n_ranges] && !end) return BMQ_E_INVAL;
        h->start_bytes.assign((const char*)start, start_off[n_ranges]);
        h->end_bytes.assign((const char*)end, end_off[n_ranges]);
    }
    if (int rc = load_router(h->r, range_flags, (const uint8_t*)h->start_bytes.data(), start_off, (const uint8_t*)h->end_bytes.data(), end_off, n_ranges))
        return rc;
    *out = h.release();
    return BMQ_OK;
}
void bmq_router_destroy(bmq_router* h) { delete h; }
int bmq_router_lookup_key(const bmq_router* h, const uint8_t* key, uint32_t key_len, int32_t* out_index) {
    if (!h || !out_index || (key_len && !key)) return BMQ_E_INVAL;
    *out_index = (int32_t)h->r.find_by_key(std::string_view((const char*)key, key_len));
    return BMQ_OK;
}
int bmq_router_lookup_boundary(const bmq_router* h, uint8_t query_flags, const uint8_t* q_start, uint32_t q_start_len, const uint8_t* q_end,
                               uint32_t q_end_len, uint32_t* out_first, uint32_t* out_count) {
    if (!h || !out_first || !out_count) return BMQ_E_INVAL;
    if (((query_flags & 1) && q_start_len && !q_start) || ((query_flags & 2) && q_end_len && !q_end)) return BMQ_E_INVAL;
    Bnd q;
    if (query_flags & 1) q.start = std::string_view((const char*)q_start, q_start_len);
    if (query_flags & 2) q.end = std::string_view((const char*)q_end, q_end_len);
    size_t lo, hi;
    h->r.find_by_boundary(q, lo, hi);
    *out_first = (uint32_t)lo;
    *out_count = (uint32_t)(hi - lo);
    return BMQ_OK;
}
int bmq_router_retain_lookup(const bmq_router* h, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* filters, const uint32_t* filter_off,
                             uint32_t n_filters, uint32_t mode, uint8_t* out_keep) {
    if (!h || (tenant_len && !tenant)) return BMQ_E_INVAL;
    return retain_range_lookup(h->r, tenant, tenant_len, filters, filter_off, n_filters, mode, out_keep);
}

} // extern "C"
