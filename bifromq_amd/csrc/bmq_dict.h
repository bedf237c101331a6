// bmq_dict.h -- host side of the level dictionary (level string -> token), shared by the dist and retain builders.
#pragma once
#include <algorithm>
#include <cstring>
#include <string_view>
#include <vector>

#include "bmq_layout.h"

namespace bmq {

inline LevelHash hash_level(std::string_view s) {
    LevelHash h = level_hash_init();
    for (size_t i = 0; i < s.size(); i += 4) {
        uint32_t w = 0;
        for (size_t k = 0; k < 4 && i + k < s.size(); k++) w |= (uint32_t)(uint8_t)s[i + k] << (8 * k);
        level_hash_word(h, w);
    }
    return h;
}

inline uint32_t pow2_at_least(uint64_t v) {
    uint32_t p = 64;
    while (p < v && p < 0x80000000u) p <<= 1;
    return p;
}

// string -> token, open addressing over a token index; same hash as the device table
struct HostDict {
    struct Entry {
        std::string_view s;
        uint32_t slot_hash, tag;
    };
    std::vector<Entry> entries;  // token - TOK_FIRST
    std::vector<uint32_t> table; // token or 0
    uint32_t mask = 0;
    HostDict() { table.assign(1024, 0); mask = 1023; }
    void grow() {
        std::vector<uint32_t> nt(table.size() * 2, 0);
        const uint32_t nm = (uint32_t)nt.size() - 1;
        for (uint32_t t = 0; t < entries.size(); t++) {
            uint32_t i = entries[t].slot_hash & nm;
            while (nt[i]) i = (i + 1) & nm;
            nt[i] = t + TOK_FIRST;
        }
        table.swap(nt);
        mask = nm;
    }
    uint32_t find(std::string_view s) const { // read-only: safe from several threads while nobody interns
        const LevelHash h = hash_level(s);
        const uint32_t sh = level_hash_slot(h, (uint32_t)s.size()), tag = level_hash_tag(h);
        uint32_t i = sh & mask;
        while (table[i]) {
            const Entry& e = entries[table[i] - TOK_FIRST];
            if (e.tag == tag && e.s == s) return table[i];
            i = (i + 1) & mask;
        }
        return TOK_UNKNOWN;
    }
    uint32_t intern(std::string_view s) {
        const LevelHash h = hash_level(s);
        const uint32_t sh = level_hash_slot(h, (uint32_t)s.size()), tag = level_hash_tag(h);
        uint32_t i = sh & mask;
        while (table[i]) {
            const Entry& e = entries[table[i] - TOK_FIRST];
            if (e.tag == tag && e.s == s) return table[i];
            i = (i + 1) & mask;
        }
        entries.push_back({s, sh, tag});
        const uint32_t tok = (uint32_t)entries.size() - 1 + TOK_FIRST;
        table[i] = tok;
        if (entries.size() * 2 > table.size()) grow();
        return tok;
    }
};


// device layout: groups of DICT_GROUP slots (one 64-byte line), load factor <= 1/4
inline void flatten_dict(const HostDict& dict_h, std::vector<DictSlot>& dict, std::vector<uint8_t>& pool) {
    const uint32_t dslots = pow2_at_least(std::max<uint64_t>(8, (uint64_t)dict_h.entries.size() * 4));
    const uint32_t gmask = dslots / DICT_GROUP - 1;
    dict.assign(dslots, DictSlot{0, 0, 0, 0, {0, 0, 0, 0}});
    pool.clear();
    for (size_t t = 0; t < dict_h.entries.size(); t++) {
        const auto& e = dict_h.entries[t];
        uint32_t g = e.slot_hash & gmask, s = NONE;
        for (;;) {
            for (uint32_t j = 0; j < DICT_GROUP && s == NONE; j++)
                if (!dict[DICT_GROUP * g + j].tag) s = DICT_GROUP * g + j;
            if (s != NONE) break;
            g = (g + 1) & gmask;
        }
        DictSlot& d = dict[s];
        d.tag = e.tag;
        d.token = (uint32_t)t + TOK_FIRST;
        d.len = (uint32_t)e.s.size();
        for (size_t i = 0; i < e.s.size() && i < 16; i++) d.inl[i >> 2] |= (uint32_t)(uint8_t)e.s[i] << (8 * (i & 3));
        if (e.s.size() > 16) {
            d.pool_off = (uint32_t)pool.size();
            pool.insert(pool.end(), e.s.begin(), e.s.end());
        }
    }
    while (pool.size() % 16 || pool.empty()) pool.push_back(0);
}

inline uint32_t dict_find(const std::vector<DictSlot>& dict, const std::vector<uint8_t>& pool, std::string_view level) {
    if (dict.empty()) return TOK_UNKNOWN;
    const LevelHash h = hash_level(level);
    const uint32_t gmask = (uint32_t)dict.size() / DICT_GROUP - 1, tag = level_hash_tag(h);
    uint32_t g = level_hash_slot(h, (uint32_t)level.size()) & gmask;
    for (;;) {
        bool group_full = true;
        for (uint32_t j = 0; j < DICT_GROUP; j++) {
            const DictSlot& d = dict[DICT_GROUP * g + j];
            if (!d.tag) {
                group_full = false;
                continue;
            }
            if (d.tag == tag && d.len == level.size()) {
                bool eq = true;
                for (size_t i = 0; i < level.size() && i < 16 && eq; i++)
                    eq = ((d.inl[i >> 2] >> (8 * (i & 3))) & 0xFF) == (uint8_t)level[i];
                if (eq && level.size() > 16) eq = memcmp(pool.data() + d.pool_off, level.data(), level.size()) == 0;
                if (eq) return d.token;
            }
        }
        if (!group_full) return TOK_UNKNOWN;
        g = (g + 1) & gmask;
    }
}

} // namespace bmq
