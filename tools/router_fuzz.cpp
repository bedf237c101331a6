// router_fuzz.cpp -- test tool, not product: bmq_router.cpp (KVRangeRouterUtil / MatchCallRangeRouter arithmetic) under
// AddressSanitizer + UBSan with hostile input: random boundary keys incl. empty, truncated and all-0xFF ones, random filters incl.
// empty levels and lone wildcards.  Checks that nothing reads out of bounds, that a lookup either succeeds or reports BMQ_E_INVAL, that
// the result of find_by_boundary is an interval of ranges each of which really overlaps the query, and that EXACT mode never drops the
// range holding a retained topic the filter matches.
// Build + run: make -C bifromq_amd/csrc routerfuzz   (tests/test_router.py runs it)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../bifromq_amd/csrc/bmq_codec.h"
#include "../include/bmq.h"

static int g_fail = 0;
#define EXPECT(c)                                                        \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "router_fuzz: %s (line %d)\n", #c, __LINE__); \
            g_fail++;                                                    \
        }                                                                \
    } while (0)

static bool matches(const std::vector<std::string>& f, const std::vector<std::string>& t) { // SURVEY.md 8a-0, '$' rule at level 0
    const bool sys = !t.empty() && !t[0].empty() && t[0][0] == '$';
    for (size_t i = 0; i < f.size(); i++) {
        if (f[i] == "#" && i + 1 == f.size()) return !(i == 0 && sys);
        if (i >= t.size()) return false;
        if (f[i] == "+") {
            if (i == 0 && sys) return false;
            continue;
        }
        if (f[i] != t[i]) return false;
    }
    return f.size() == t.size();
}
static std::vector<std::string> split(const std::string& s) {
    std::vector<std::string> out;
    size_t b = 0;
    for (size_t i = 0; i <= s.size(); i++)
        if (i == s.size() || s[i] == '/') {
            out.push_back(s.substr(b, i - b));
            b = i + 1;
        }
    return out;
}

struct Packed {
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off{0};
    void add(const std::string& s) {
        bytes.insert(bytes.end(), s.begin(), s.end());
        off.push_back((uint32_t)bytes.size());
    }
    const uint8_t* data() const { return bytes.empty() ? (const uint8_t*)"" : bytes.data(); }
};

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 400;
    std::mt19937_64 rng(seed);
    const std::vector<std::string> alpha = {"a", "b", "", "$s", "c", "dd", "\xE4\xBD\xA0"};
    auto level = [&]() { return alpha[rng() % alpha.size()]; };
    auto topic = [&]() {
        std::string t;
        for (size_t d = 1 + rng() % 4, k = 0; k < d; k++) t += (k ? "/" : "") + level();
        return t;
    };
    uint64_t lookups = 0, invalid = 0;
    for (int round = 0; round < rounds; round++) {
        const std::string tenant = rng() % 4 ? "tenantA" : "t";
        const bool hostile = round % 3 == 2; // boundaries that are no retain keys at all
        // split points -> boundaries of a partition of the key space
        std::set<std::string> cuts;
        std::vector<std::string> topics;
        for (size_t i = 0, n = rng() % 9; i < n; i++) {
            if (hostile) {
                std::string k;
                switch (rng() % 5) {
                case 0: k = std::string(rng() % 4, (char)0xFF); break;
                case 1: k = bmq::retain_message_key(tenant, topic()).substr(0, 1 + rng() % 12); break; // truncated inside the header
                case 2: k = std::string("\0\0\x07tenantA", 10) + std::string(rng() % 3, (char)(rng() % 256)); break;
                case 3: k = bmq::retain_message_key(rng() % 2 ? "tenantB" : "s", topic()); break;
                default: k = bmq::retain_message_key(tenant, topic()); break;
                }
                cuts.insert(k);
            } else {
                topics.push_back(topic());
                cuts.insert(bmq::retain_message_key(tenant, topics.back()));
            }
        }
        cuts.erase(std::string()); // an empty start key would equal "no start key" only by accident of the flags
        std::vector<std::string> cv(cuts.begin(), cuts.end());
        const uint32_t n_ranges = (uint32_t)cv.size() + 1;
        std::vector<uint8_t> flags(n_ranges);
        Packed st, en;
        for (uint32_t r = 0; r < n_ranges; r++) {
            flags[r] = (r > 0 ? 1 : 0) | (r + 1 < n_ranges ? 2 : 0);
            st.add(r > 0 ? cv[r - 1] : "");
            en.add(r + 1 < n_ranges ? cv[r] : "");
        }
        // filters
        std::vector<std::string> filters;
        Packed fp;
        for (int i = 0; i < 12; i++) {
            std::string f;
            const size_t d = 1 + rng() % 4;
            for (size_t k = 0; k < d; k++) {
                const int x = (int)(rng() % 9);
                f += (k ? "/" : "") + (x < 2 ? std::string("+") : (x == 2 && k + 1 == d ? std::string("#") : level()));
            }
            filters.push_back(f);
            fp.add(f);
        }
        for (uint32_t mode = 0; mode < 2; mode++) {
            std::vector<uint8_t> keep((size_t)filters.size() * n_ranges, 0xEE);
            const int rc = bmq_retain_range_lookup((const uint8_t*)tenant.data(), (uint32_t)tenant.size(), fp.data(), fp.off.data(), (uint32_t)filters.size(),
                                                   flags.data(), st.data(), st.off.data(), en.data(), en.off.data(), n_ranges, mode, keep.data());
            lookups++;
            EXPECT(rc == BMQ_OK || rc == BMQ_E_INVAL);
            if (rc != BMQ_OK) {
                invalid++;
                EXPECT(hostile); // a router cut at real retain keys never fails
                continue;
            }
            for (uint8_t v : keep) EXPECT(v <= 1);
            if (mode == 1 && !hostile) // exact mode: the home range of every matching retained topic is asked
                for (size_t f = 0; f < filters.size(); f++)
                    for (auto& t : topics)
                        if (matches(split(filters[f]), split(t)) || filters[f] == t) {
                            const std::string k = bmq::retain_message_key(tenant, t);
                            int32_t home = -1;
                            EXPECT(bmq_router_find_by_key(flags.data(), st.data(), st.off.data(), en.data(), en.off.data(), n_ranges, (const uint8_t*)k.data(),
                                                          (uint32_t)k.size(), &home) == BMQ_OK &&
                                   home >= 0);
                            if (home >= 0) EXPECT(keep[f * n_ranges + (size_t)home] == 1);
                        }
        }
        // find_by_boundary: an interval; every member overlaps the query, every non-member does not (queries with both ends)
        for (int q = 0; q < 6; q++) {
            std::string a = cv.empty() || rng() % 2 ? bmq::retain_message_key(tenant, topic()) : cv[rng() % cv.size()];
            std::string b = bmq::retain_message_key(tenant, topic());
            if (b < a) std::swap(a, b);
            uint32_t first = 0, count = 0;
            const int rc = bmq_router_find_by_boundary(flags.data(), st.data(), st.off.data(), en.data(), en.off.data(), n_ranges, 3, (const uint8_t*)a.data(),
                                                       (uint32_t)a.size(), (const uint8_t*)b.data(), (uint32_t)b.size(), &first, &count);
            EXPECT(rc == BMQ_OK && first + count <= n_ranges);
            for (uint32_t r = 0; r < n_ranges && rc == BMQ_OK; r++) {
                const bool overlap = (r == 0 || cv[r - 1] < b) && (r + 1 == n_ranges || a < cv[r]); // [start, end) meets [a, b), a < b
                const bool member = r >= first && r < first + count;
                if (a < b) EXPECT(member == overlap);
            }
        }
    }
    if (g_fail) {
        fprintf(stderr, "router_fuzz FAILED: %d\n", g_fail);
        return 1;
    }
    printf("router_fuzz ok: seed %llu, %d rounds, %llu lookups (%llu refused as malformed)\n", (unsigned long long)seed, rounds, (unsigned long long)lookups,
           (unsigned long long)invalid);
    return 0;
}
