// Host-side tables of the contiguous-NFA "shallow-skip" walk kernel (device/cnfa_tri.hip); see cnfa_tri_tables.cpp.
#pragma once
#include <stdint.h>

#include <vector>

#include "automaton.hpp"

namespace acgpu {

constexpr uint32_t kTriShallow = 0x80000000u;   // a fail word of repr3 that names a state of depth <= 2 (start, its children, theirs)
constexpr size_t kTriReprPad = 8;               // words behind repr3: a 16-byte record load of the last state stays inside
constexpr size_t kTriLdsBudget = 156 * 1024;    // dynamic LDS a workgroup of the kernel may use
constexpr size_t kTriLaneBuf = 16 * 1024;       // ... of which 16 bytes per lane hold the piece at hand (1 024 lanes)

// 16 bytes per trie node of depth 3, ordered by (pair of the two bytes before, compact class of the third):
// what a step from a depth-2 state into depth 3 needs, in ONE gather.
struct TriChild {
    uint32_t o;      // the depth-3 state (offset into repr3)
    uint32_t head;   // repr3[o]: kind | class of a one-transition state
    uint32_t fail;   // repr3[o + 1] (kTriShallow-tagged when it names a state of depth <= 2)
    uint32_t d0;     // repr3[o + 2]: target of a one-transition state / first class word / match word of a leaf
};

// One match event of the walk: the records of `state`'s match list (state & 0x80000000: the state of depth <= 2 of pair
// `state & 0x7FFFFFFF`), `pre` records into chunk `ci`'s slice of the output; the match ends behind byte `rel` of the chunk.
struct TriEvent {
    uint32_t ci, pre, state, rel;
};
constexpr uint32_t kTriSeg = 64;   // events per wave-private segment of the event buffer

struct CnfaTriHost {
    bool ok = false;
    uint32_t n_used = 0;               // U: classes that label some trie edge; compact ids 0..U-1, U = "no such edge"
    uint32_t apair = 0;                // A' = U + 1: pair index = ua * A' + ub
    uint32_t bw = 0;                   // bitmap words per pair
    uint32_t granule = 1;              // child index of a pair's first child = base[pair] * granule
    std::vector<uint8_t> uc;           // [256] byte -> compact class
    std::vector<uint8_t> inv;          // [256] compact class -> the automaton's class (inv[U]: a class no trie edge carries)
    std::vector<uint32_t> bits;        // [A'^2][bw]   bit uc of pair (ua, ub): the trie has the node ua ub uc
    std::vector<uint16_t> base;        // [A'^2]
    std::vector<TriChild> child;       // depth-3 nodes
    std::vector<uint32_t> repr3;       // repr with the fail words that name states of depth <= 2 tagged (padded)
    bool shallow_matches = false;      // some state of depth <= 2 is a match state (patterns of <= 2 bytes, empty patterns)
    std::vector<uint8_t> mc2;          // [A'^2] match-list length of the state "last two bytes = pair" (only if shallow_matches)
    std::vector<uint32_t> st2;         // [A'^2] that state (repr offset; for the records of shallow matches)
    uint32_t start_mlen = 0;           // match-list length of the start state (empty pattern)
    size_t lds_bytes = 0;
};

// false: the kernel does not serve this automaton (alphabet too large for the pair tables, no unanchored start, ...)
bool build_cnfa_tri_host(const CNfa& c, CnfaTriHost& t);
// The kernel's walk (shallow skip + failure-link walk of the deep states) over hay[0..len), cold start at 0: the
// overlapping search's match count, start-state matches included.  `steps` (optional): [0] gathers of child entries,
// [1] gathers of state records, [2] other gathers.
uint64_t cnfa_tri_emulate_count(const CnfaTriHost& t, const CNfa& c, const uint8_t* hay, size_t len, uint64_t* steps = nullptr);

}  // namespace acgpu
