// Host-side tables of the DFA "shallow-skip" transition walk (device/dfa_tri.hip); see dfa_tri_tables.cpp.
#pragma once
#include <stdint.h>

#include <vector>

#include "automaton.hpp"
#include "cnfa_tri_tables.hpp"   // kTriShallow, kTriLdsBudget, kTriLaneBuf, TriEvent

namespace acgpu {

struct DfaTriHost {
    bool ok = false;
    uint32_t n_used = 0, apair = 0, bw = 0, granule = 1;   // as in CnfaTriHost
    std::vector<uint8_t> uc, inv;      // [256] byte -> compact class; compact class -> the DFA's class
    std::vector<uint32_t> bits;        // [A'^2][bw] bit uc of pair (ua, ub): the trie has the node ua ub uc
    std::vector<uint16_t> base;        // [A'^2]
    std::vector<uint32_t> child;       // the depth-3 nodes as (premultiplied) DFA state ids, ordered by (pair, class)
    std::vector<uint32_t> trans3;      // the DFA's transition table, targets of depth <= 2 tagged kTriShallow
    bool shallow_matches = false;
    std::vector<uint8_t> mc2;          // [A'^2] match-list length of the state "last two bytes = pair" (only if shallow_matches)
    std::vector<uint32_t> st2;         // [A'^2] that state
    uint32_t start_mlen = 0;
    size_t lds_bytes = 0;
};

// false: the kernel does not serve this automaton (two-start layout, alphabet too large for the pair tables, ...)
bool build_dfa_tri_host(const NNfa& n, const Dfa& d, DfaTriHost& t);

}  // namespace acgpu
