// bmq_range_core.h -- dist-server side range pruning: TenantRangeLookupCache.lookup
// (bifromq-dist/bifromq-dist-server/src/main/java/org/apache/bifromq/dist/server/scheduler/TenantRangeLookupCache.java:62-109).
//
// The reference builds a one-topic GLOBAL trie ([tenantId, level0, level1, ...]), opens the expansion iterator
// (TRIE/TopicFilterIterator.java) and, for every candidate KV range in boundary order, seeks to the range's FIRST global filter
// (Fact.proto:27-34): the candidate is kept iff the smallest expansion filter >= first exists and is == first or <= last (compared
// as NUL-joined strings); if no expansion filter >= first exists the loop stops (later ranges start even further right).
//
// Here the smallest expansion filter >= first is computed directly.  The expansion set of topic t0/../t(n-1) under tenant T is
//   [T, x0, .., x(k-1)]        with k == n, or
//   [T, x0, .., x(k-1), '#']   with 0 <= k <= n,          x_i in {t_i, '+'}
// minus wildcards in position 0 for a '$' topic (TRIE/TopicTrieNode.java:150-152), ordered by level list (levels as byte strings;
// a proper prefix first).  Its smallest element at or after `first` shares the longest possible prefix with `first`: equal to it
// if `first` is itself in the set, else the smallest proper extension if `first` is a valid prefix, else the smallest step UP at
// the deepest position where one exists.  O(levels) per (topic, candidate), no trie, no iterator.
// BMQ_HD: k_range_lookup (gfx950) and tests/c/range_shim.cpp (host, test tool) run the same function.
#pragma once
#include <stdint.h>

#include "bmq_layout.h"

namespace bmq {

constexpr uint32_t RL_MAX_LEVELS = 64; // topics / first filters with more levels are answered conservatively: kept

struct LevelSpan {
    uint32_t beg, end; // byte range of one level
};
// compare level a (bytes pa[a.beg..a.end)) with level b: <0, 0, >0 (unsigned bytes, shorter first)
BMQ_HD int level_cmp(const uint8_t* pa, LevelSpan a, const uint8_t* pb, LevelSpan b) {
    const uint32_t la = a.end - a.beg, lb = b.end - b.beg, m = la < lb ? la : lb;
    for (uint32_t i = 0; i < m; i++) {
        const int d = (int)pa[a.beg + i] - (int)pb[b.beg + i];
        if (d) return d;
    }
    return la < lb ? -1 : (la > lb ? 1 : 0);
}
BMQ_HD int level_cmp_lit(const uint8_t* pa, LevelSpan a, char c) { // against the one-byte level "c"
    const uint32_t la = a.end - a.beg;
    if (la == 0) return -1;
    const int d = (int)pa[a.beg] - (int)(uint8_t)c;
    if (d) return d;
    return la > 1 ? 1 : 0;
}
// split [beg, end) of p on `sep` into at most cap spans; returns the level count (> cap: too many)
BMQ_HD uint32_t split_levels(const uint8_t* p, uint32_t beg, uint32_t end, uint8_t sep, LevelSpan* out, uint32_t cap) {
    uint32_t n = 0, s = beg;
    for (uint32_t i = beg; i <= end; i++)
        if (i == end || p[i] == sep) {
            if (n < cap) out[n] = LevelSpan{s, i};
            n++;
            s = i + 1;
        }
    return n;
}

// A filter of the expansion set, described symbol by symbol: sym[i] 0 = the topic's level i, 1 = '+', 2 = '#' (last only)
struct ExpFilter {
    uint32_t n;                  // symbols after the tenant level
    uint8_t sym[RL_MAX_LEVELS + 1];
};
enum : uint8_t { SYM_TOPIC = 0, SYM_PLUS = 1, SYM_HASH = 2 };

// compare the symbol `s` at position i of an expansion filter with level f of `first` / `last`
BMQ_HD int sym_cmp(uint8_t s, const uint8_t* tp, LevelSpan t_i, const uint8_t* fp, LevelSpan f) {
    if (s == SYM_TOPIC) return level_cmp(tp, t_i, fp, f);
    return -level_cmp_lit(fp, f, s == SYM_PLUS ? '+' : '#');
}
// smallest element of the expansion set that starts with the valid prefix out.sym[0..k): greedy, smallest symbol first
BMQ_HD void smallest_with_prefix(ExpFilter& out, uint32_t k, const uint8_t* tp, const LevelSpan* t, uint32_t n, bool sys) {
    while (k < n) {
        const bool wild_ok = !(k == 0 && sys);
        // options at position k: '#' (terminal), '+', t_k -- distinct strings (a topic level contains neither '+' nor '#').
        // '#' < '+' always, so the smallest is whichever of t_k and '#' sorts first
        const uint8_t best = (wild_ok && level_cmp_lit(tp, t[k], '#') > 0) ? SYM_HASH : SYM_TOPIC;
        out.sym[k] = best;
        k++;
        if (best == SYM_HASH) {
            out.n = k;
            return;
        }
    }
    out.n = k; // k == n: the prefix itself is a member (every level consumed)
}

// result of the seek: 0 = no expansion filter >= first (iterator invalid), 1 = found (in `out`), 2 = undecidable here (too deep)
BMQ_HD int expansion_seek(const uint8_t* tp, const LevelSpan* t, uint32_t n, bool sys, const uint8_t* fp, const LevelSpan* f, uint32_t m,
                          ExpFilter& out) {
    // f[0..m): the levels of `first` AFTER the tenant level.  v = longest prefix of f made of valid non-terminal symbols
    uint32_t v = 0;
    while (v < m && v < n) {
        const bool wild_ok = !(v == 0 && sys);
        if (level_cmp(tp, t[v], fp, f[v]) == 0) out.sym[v] = SYM_TOPIC;
        else if (wild_ok && level_cmp_lit(fp, f[v], '+') == 0) out.sym[v] = SYM_PLUS;
        else break;
        v++;
    }
    if (v == m) { // all of `first` is a valid prefix
        if (m == n) {
            out.n = n;
            return 1; // first itself
        }
        smallest_with_prefix(out, m, tp, t, n, sys); // its smallest proper extension
        return 1;
    }
    if (v == m - 1 && level_cmp_lit(fp, f[v], '#') == 0 && !(v == 0 && sys)) { // first = valid prefix + '#': itself (v <= n holds)
        out.sym[v] = SYM_HASH;
        out.n = m;
        return 1;
    }
    // step up at the deepest position p <= v where an allowed symbol is greater than f[p]
    for (uint32_t p = v + 1; p-- > 0;) {
        const bool wild_ok = !(p == 0 && sys);
        // allowed symbols at p: '#' (p <= n), '+' and t_p (p < n); pick the smallest one that is > f[p]
        int best = -1;
        // candidates in ascending order are not fixed (t_p may sort anywhere): test all three, keep the smallest qualifying
        uint8_t cand[3];
        uint32_t nc = 0;
        if (wild_ok) cand[nc++] = SYM_HASH;
        if (p < n) {
            if (wild_ok) cand[nc++] = SYM_PLUS;
            cand[nc++] = SYM_TOPIC;
        }
        for (uint32_t c = 0; c < nc; c++) {
            const uint8_t s = cand[c];
            const LevelSpan tl = p < n ? t[p] : LevelSpan{0, 0};
            if (sym_cmp(s, tp, tl, fp, f[p]) <= 0) continue; // not above f[p]
            if (best < 0) best = s;
            else { // is s smaller than the current best?  compare the two symbols as strings
                const uint8_t b = (uint8_t)best;
                int d;
                if (s == SYM_TOPIC && b != SYM_TOPIC) d = level_cmp_lit(tp, tl, b == SYM_PLUS ? '+' : '#');
                else if (s != SYM_TOPIC && b == SYM_TOPIC) d = -level_cmp_lit(tp, tl, s == SYM_PLUS ? '+' : '#');
                else d = (s == SYM_HASH) ? -1 : 1; // '#' < '+'
                if (d < 0) best = s;
            }
        }
        if (best >= 0) {
            out.sym[p] = (uint8_t)best;
            if (best == SYM_HASH) out.n = p + 1;
            else smallest_with_prefix(out, p + 1, tp, t, n, sys);
            return 1;
        }
    }
    return 0;
}

// compare the expansion filter g = [tenant] + out with the level list L = [L0, L1, ...] (tenant level included): <0, 0, >0
BMQ_HD int expfilter_cmp(const ExpFilter& g, const uint8_t* tenant, uint32_t tenant_len, const uint8_t* tp, const LevelSpan* t, const uint8_t* lp,
                         const LevelSpan* L, uint32_t nl) {
    if (nl == 0) return 1; // g has at least the tenant level
    int d = level_cmp(tenant, LevelSpan{0, tenant_len}, lp, L[0]);
    if (d) return d;
    for (uint32_t i = 0; i < g.n; i++) {
        if (i + 1 >= nl) return 1; // L is a proper prefix of g
        d = sym_cmp(g.sym[i], tp, g.sym[i] == SYM_TOPIC ? t[i] : LevelSpan{0, 0}, lp, L[i + 1]);
        if (d) return d;
    }
    return g.n + 1 < nl ? -1 : 0;
}

// One topic against the candidates in order.  kind[c]: 0 = no Fact (kept), 1 = Fact of an empty range (dropped), 2 = first/last given.
// first/last: NUL-joined global filter levels (tenant level first), packed.  keep[c] = 1/0.
BMQ_HD void range_lookup_one(const uint8_t* tenant, uint32_t tenant_len, const uint8_t* topics, uint32_t t_beg, uint32_t t_end, const uint8_t* kind,
                             const uint8_t* first, const uint32_t* first_off, const uint8_t* last, const uint32_t* last_off, uint32_t n_cand,
                             uint8_t* keep) {
    LevelSpan t[RL_MAX_LEVELS], f[RL_MAX_LEVELS + 1], L[RL_MAX_LEVELS + 2];
    const uint32_t n = split_levels(topics, t_beg, t_end, '/', t, RL_MAX_LEVELS);
    const bool sys = t_end > t_beg && topics[t_beg] == '$';
    bool stopped = false;
    for (uint32_t c = 0; c < n_cand; c++) {
        keep[c] = 0;
        if (stopped) continue;
        if (kind[c] == 0) {
            keep[c] = 1;
            continue;
        }
        if (kind[c] == 1) continue;
        const uint32_t nf = split_levels(first, first_off[c], first_off[c + 1], 0, f, RL_MAX_LEVELS + 1);
        const uint32_t nl = split_levels(last, last_off[c], last_off[c + 1], 0, L, RL_MAX_LEVELS + 2);
        if (n > RL_MAX_LEVELS || nf > RL_MAX_LEVELS + 1 || nl > RL_MAX_LEVELS + 2) { // too deep to decide here: never prune wrongly
            keep[c] = 1;
            continue;
        }
        ExpFilter g;
        g.n = 0;
        int found;
        // the tenant level decides first
        const int dt = nf ? level_cmp(tenant, LevelSpan{0, tenant_len}, first, f[0]) : 1;
        if (dt < 0) found = 0;                       // every expansion filter sorts before `first`
        else if (dt > 0 || nf == 1) {                // ... after it (or `first` is the bare tenant level): the smallest element
            smallest_with_prefix(g, 0, topics, t, n, sys);
            found = 1;
        } else found = expansion_seek(topics, t, n, sys, first, f + 1, nf - 1, g);
        if (!found) {
            stopped = true; // TenantRangeLookupCache.java:99-101: "endTopicFilter < firstTopicFilter, stop"
            continue;
        }
        const bool is_first = expfilter_cmp(g, tenant, tenant_len, topics, t, first, f, nf) == 0;
        keep[c] = (is_first || expfilter_cmp(g, tenant, tenant_len, topics, t, last, L, nl) <= 0) ? 1 : 0;
    }
}

} // namespace bmq
