// Internals of the C ABI implementation shared by capi.cpp and the test hooks (test_hooks.cpp): the automaton handle.
#pragma once
#include <map>
#include <memory>
#include <mutex>

#include "acgpu.h"
#include "host/automaton.hpp"
#include "host/variants.hpp"

namespace acgpu_capi { struct DeviceState; }

struct acgpu_automaton {
    acgpu_config cfg{};
    int kind = ACGPU_KIND_NONCONTIGUOUS_NFA;  // resolved AhoCorasickKind
    acgpu::NNfa nnfa;
    acgpu::Dfa dfa;
    acgpu::CNfa cnfa;
    bool has_dfa = false, has_cnfa = false;
    // For leftmost match kinds: the MatchKind::Standard automaton of the same patterns.  Its overlapping stream is
    // "every occurrence of every pattern", from which the parallel find_iter selects (device/select.hpp).
    std::unique_ptr<acgpu_automaton> occ;
    // Split pattern set (Standard / unanchored automata with at least 1 000 long patterns and 1..64 short ones, the shortest of
    // at most 6 bytes): part[0] = the patterns of nine bytes and more, part[1] = the others, both reporting the ids of the
    // full set.  The overlapping search runs both and merges their record streams (capi_overlap.cpp: overlapping_split): the
    // large-set filter's long-key level 1 is ten times faster over natural text than anything a 3-byte word lets it use.
    std::unique_ptr<acgpu_automaton> part[2];
    acgpu::Variants var;   // engine variants (acgpu_set_variant): copied into the device tables at upload
    std::mutex mu;
    std::map<int, std::unique_ptr<acgpu_capi::DeviceState>> devs;
    acgpu_automaton();
    ~acgpu_automaton();
};


// bytes per lane-chunk of a search of this automaton (capi.cpp::default_chunk)
uint32_t acgpu_default_chunk(const acgpu_automaton* aut, size_t span_len);
// sets what acgpu_last_error() returns on this thread (multi.cpp mirrors its errors into it)
void acgpu_set_last_error(const char* msg);
