// Host-side tables of the prefix-filter kernels (see pf_tables.cpp).
#pragma once
#include <stdint.h>

#include <vector>

#include "../device/hot.hpp"
#include "automaton.hpp"

namespace acgpu {

struct PfHostTables {
    bool ok = false;                 // the two-type filter serves this automaton
    bool pfx_ok = false;             // ... and so does the large-set filter (>= 256 patterns, none shorter than 4 bytes)
    std::vector<uint32_t> own;       // [hid] patterns ending exactly in this trie node
    std::vector<uint32_t> own_pid;   // [hid] the lowest id among them
    std::vector<uint32_t> atab;      // [hid << ashift | class] trie-only transitions: child hid | 1 << 31 if a pattern ends there
    std::vector<uint8_t> acls;       // [256] class map of atab (0 = byte on no trie edge)
    uint32_t ashift = 8;
    std::vector<uint32_t> bits, bits2, bits3;   // two-type filter: first / second table (64 KiB each), exact-4-byte bit table
    uint32_t bits_bytes = 0, bits3_log2 = 0;
    bool exact2 = false, use3 = false;
    bool fold = false;               // two-type filter: key bytes of both tables are taken | 0x20 (case-folded automata)
    std::vector<uint32_t> xbits;     // large-set filter: blocked Bloom table (kPfxBitsBytes)
    std::vector<uint32_t> xbits8;    // ... keyed by the first eight bytes (pfx_hash8); empty unless pfx_depth == 8
    std::vector<uint32_t> xbits8x2;  // ... probed at every other position (pfx_x2_mask); empty unless every pattern has >= 9 bytes
    std::vector<uint32_t> pfx_map, pfx_map8;    // its exact level-2 maps (HotTables::pfx_map / pfx_map8)
    std::vector<uint32_t> pfx_tails;            // chain tails behind pfx_map8 (kPfxTailWords words each; entry word 3 = index + 1): see pf_tables.cpp
    uint32_t pfx_tail_nodes = 0;                // diagnostics: depth-`pfx_depth` nodes with a tail record
    uint32_t pfx_map_log2 = 0, pfx_map8_log2 = 0, pfx_depth = 4, pfx_prefixes = 0;
    uint32_t n_patterns = 0;
    // Short mode (round 6): a dictionary of patterns of >= 9 bytes with one or two stragglers of 3..8 bytes.  The long-key
    // tables (pfx_map8, tails, xbits8, xbits8x2) are built from the LONG patterns only -- prefix depth 8, level 1 at every
    // other position -- and the stragglers are compared in the producers' registers at every position (pfx_scan.hip):
    // key / mask of the first min(len, 4) bytes there, all bytes by the verifier.  The 4-byte tables stay complete when no
    // straggler is shorter than four bytes (pfx4_complete); in short mode only the every-other-position kernel may run the
    // long-key tables.
    uint32_t short_n = 0;
    uint32_t short_lo[kPfxShortMax] = {}, short_hi[kPfxShortMax] = {};   // bytes 0..3 / 4..7, zero-padded
    uint32_t short_len[kPfxShortMax] = {}, short_node[kPfxShortMax] = {};   // length, trie node (hid)
    bool pfx4_complete = true;
};

// tails / key8_x2: build the chain-tail records / the every-other-position table of the long-key level 1 (Variants)
bool build_pf_host(const NNfa& n, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid, PfHostTables& t,
                   int tails = 2, bool key8_x2 = true, bool short_mode = true);   // tails: 2 records for small subtrees | 1 chain tails only | 0 none
// test hook: the decisions of kernel 0 (two-type filter), 1 (large-set filter, 4-byte level 2), 2 (large-set filter,
// long-prefix level 2) or 3 (the same with the eight-byte level 1) over haystack[0..len) with a cold start at 0; returns the number of occurrences level 3 finds
// (UINT64_MAX: that kernel does not serve the automaton); info[0..1] = survivors of level 1 / level 2
uint64_t pf_emulate_count(const PfHostTables& t, uint32_t start_hid, const uint8_t* hay, size_t len, int kernel, uint64_t* info);

}  // namespace acgpu
