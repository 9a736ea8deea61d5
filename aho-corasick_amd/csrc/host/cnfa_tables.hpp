// Host-side tables of the contiguous-NFA walk kernel (see cnfa_tables.cpp).
#pragma once
#include <stdint.h>

#include <vector>

#include "automaton.hpp"

namespace acgpu {

constexpr uint32_t kCnfaSlotTag = 0x80000000u;   // state word = kCnfaSlotTag | slot: the state lives in LDS
constexpr uint32_t kCnfaMidTag = 0x40000000u;    // state word = kCnfaMidTag | idx: a dense state with its row in mid_rows, fail word + match count in LDS
constexpr uint32_t kCnfaMaxMid = 12288;          // 72 KiB of LDS for their fail words and match counts
constexpr size_t kCnfaReprPad = 320;             // words behind repr: the speculative dense-layout load of the last states stays inside

struct CnfaHotHost {
    bool ok = false;
    std::vector<uint32_t> rows;     // [n_slots][alphabet_len + 1]: fail state, dense transitions (slot 0 = the unanchored start state)
    std::vector<uint32_t> mcnt;     // [n_slots] match-list length of the slot's state, 0 = not a match state
    std::vector<uint32_t> repr_t;   // repr with every fail word / transition target that names an LDS-resident state = tag | slot (padded)
    std::vector<uint32_t> mid_rows; // [n_mid][1 << mid_shift] dense transitions of the second-tier states (global memory)
    uint32_t mid_shift = 0;
    std::vector<uint32_t> mid_fail; // [n_mid] their fail states (LDS)
    std::vector<uint16_t> mid_mcnt; // [n_mid] their match-list lengths (LDS)
    uint32_t n_slots = 0, row_words = 0, n_mid = 0;
    bool mid_matches = false;       // some second-tier state is a match state
    bool dense_outside = true;      // some dense state is not in LDS
    bool sorted_sparse = false;     // every sparse state lists its classes in ascending order
    bool slot_matches = false;      // some LDS-resident state is a match state
};

bool build_cnfa_hot_host(const CNfa& c, CnfaHotHost& t);
uint64_t cnfa_emulate_count(const CnfaHotHost& t, const CNfa& c, const uint8_t* hay, size_t len);

}  // namespace acgpu
