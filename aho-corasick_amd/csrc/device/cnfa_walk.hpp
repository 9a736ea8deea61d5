// Contiguous-NFA failure-link walk with the start state and its children in LDS (cnfa_walk.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/automaton.hpp"
#include "kernels.hpp"

namespace acgpu {

struct CnfaHotDev {
    uint32_t* rows = nullptr;    // [n_slots][alphabet_len + 1]: fail state, dense transitions (slot 0 = the unanchored start state)
    uint32_t* mcnt = nullptr;    // [n_slots] match-list length of the slot's state, 0 = not a match state
    // device copy of `repr` in which every fail word and transition target that names an LDS-resident state reads
    // 0x80000000 | slot -- the walk tests one bit instead of hashing the state id at every hop (padded like the original)
    uint32_t* repr_t = nullptr;
    uint32_t* mid_rows = nullptr;   // [n_mid][1 << mid_shift] second-tier dense states: transitions (global), CnfaHotHost
    uint32_t mid_shift = 0;
    uint32_t* mid_fail = nullptr;   // [n_mid] fail states (copied to LDS)
    uint16_t* mid_mcnt = nullptr;   // [n_mid] match-list lengths (copied to LDS)
    uint32_t n_slots = 0, row_words = 0, n_mid = 0, mid_matches = 0;
    // what the tables guarantee about the states that are NOT in LDS (found by a traversal at upload):
    uint32_t dense_outside = 1;   // some dense state is not in LDS: the walk loads the dense-layout transition speculatively
    uint32_t sorted_sparse = 0;   // every sparse state lists its classes in ascending order: a lookup stops at the first larger one
    uint32_t slot_matches = 0;    // some LDS-resident state is a match state (1-byte patterns, empty patterns)
};
struct CnfaHotTables {
    bool ready = false;
    size_t repr_words = 0;   // size of the automaton (launch_cnfa_count: one or two workgroups per CU)
    CnfaHotDev dev;
    CnfaHotTables() = default;
    CnfaHotTables(const CnfaHotTables&) = delete;
    CnfaHotTables& operator=(const CnfaHotTables&) = delete;
    ~CnfaHotTables();
};

hipError_t build_cnfa_hot(const CNfa& c, CnfaHotTables& out);
hipError_t launch_cnfa_count(const CnfaHotTables& h, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts, hipStream_t s);

}  // namespace acgpu
