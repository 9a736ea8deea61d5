// Host-side automaton tables of the acgpu engine.
//
// Construction stays on the CPU (north star); these are the tables the reference
// builds (noncontiguous NFA -> DFA | contiguous NFA), produced by our own builder
// (builder.cpp) with the SAME state numbering and the SAME match-list order, so
// that every search result is bit-identical.  Layouts are CSR/flat, ready to be
// re-encoded for the device (device_tables.hpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "acgpu.h"

namespace acgpu {

constexpr uint32_t kDead = 0;  // src/nfa/noncontiguous.rs:214
constexpr uint32_t kFail = 1;  // src/nfa/noncontiguous.rs:221
// StateID/PatternID/SmallIndex::MAX == i32::MAX - 1, src/util/primitives.rs:95-111
constexpr uint64_t kSmallIndexMax = 0x7FFFFFFEull;

// src/util/special.rs:10-28
struct Special {
    uint32_t max_special_id = 0, max_match_id = 0, start_unanchored_id = 0, start_anchored_id = 0;
};

// Noncontiguous NFA in its final (post-shuffle) numbering, src/nfa/noncontiguous.rs:102-175.
// Transitions and match lists are CSR instead of in-vector linked lists.
struct NNfa {
    int match_kind = ACGPU_MATCH_STANDARD;
    std::vector<uint32_t> fail, depth;          // [states]
    std::vector<uint32_t> toff;                 // [states+1] into tbyte/tnext (sorted by byte)
    std::vector<uint8_t> tbyte;
    std::vector<uint32_t> tnext;
    std::vector<uint32_t> moff, mpid;           // match lists, reference order
    std::vector<uint32_t> pattern_lens;         // indexed by pattern id (see BuildOptions::pattern_ids: may be sparser than the patterns given)
    size_t n_patterns = 0;                      // patterns this automaton was built from
    std::vector<uint32_t> bfs;                  // non-sentinel states in breadth-first (fail-closed) order
    uint8_t byte_classes[256] = {0};
    size_t min_pattern_len = SIZE_MAX, max_pattern_len = 0;
    size_t dense_states = 0;                    // how many states the reference would densify (memory_usage only)
    Special special;

    size_t states() const { return fail.size(); }
    size_t alphabet_len() const { return size_t(byte_classes[255]) + 1; }
    bool is_match(uint32_t sid) const { return moff[sid + 1] != moff[sid]; }
    // explicit transition or kFail
    uint32_t follow(uint32_t sid, uint8_t byte) const;
    // src/nfa/noncontiguous.rs:601-626
    uint32_t next_state(bool anchored, uint32_t sid, uint8_t byte) const;
};

// src/dfa.rs:86-132
struct Dfa {
    std::vector<uint32_t> trans;                // premultiplied ids, row-major, stride = 1<<stride2
    std::vector<uint32_t> moff, mpid;           // CSR over match-state index (sid>>stride2)-2
    size_t state_len = 0, alphabet_len = 0, stride2 = 0, num_match_states = 0;
    uint8_t byte_classes[256] = {0};
    Special special;
};

// src/nfa/contiguous.rs:86-135
struct CNfa {
    std::vector<uint32_t> repr;
    size_t state_len = 0, alphabet_len = 0;
    uint8_t byte_classes[256] = {0};
    Special special;
};

struct BuildOptions {
    int match_kind = ACGPU_MATCH_STANDARD;
    bool ascii_case_insensitive = false;
    size_t nnfa_dense_depth = 3;   // src/nfa/noncontiguous.rs:855
    size_t cnfa_dense_depth = 2;   // src/nfa/contiguous.rs:904
    bool byte_classes = true;
    int start_kind = ACGPU_START_UNANCHORED;
    // Explicit pattern ids (ascending, one per pattern given) in an id space of `id_space` patterns: the automaton of a SUBSET
    // of a pattern set reports the ids -- and looks up the lengths -- of the full set (capi.cpp: split sets).  nullptr: 0..n-1.
    const uint32_t* pattern_ids = nullptr;
    size_t id_space = 0;
};

acgpu_status build_nnfa(const BuildOptions& o, const uint8_t* const* pats, const size_t* lens, size_t n, NNfa& out);
// `fill`: optional replacement for the host loop that computes the transition rows of the non-Both layouts (the
// GPU-side fill, device/dfa_fill.hip); returns false on failure.  The tables it must produce are the host loop's.
using DfaRowFill = bool (*)(const NNfa& n, const uint8_t* classes, size_t alen, size_t s2, bool anchored, uint32_t* trans);
acgpu_status build_dfa(const NNfa& n, int start_kind, bool byte_classes, Dfa& out, DfaRowFill fill = nullptr);
acgpu_status build_cnfa(const NNfa& n, size_t dense_depth, bool byte_classes, CNfa& out);

}  // namespace acgpu
