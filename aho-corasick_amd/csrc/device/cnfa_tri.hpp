// Contiguous-NFA failure-link walk with an exact skip of the depth <= 2 regime (cnfa_tri.hip; host/cnfa_tri_tables.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/automaton.hpp"
#include "../host/cnfa_tri_tables.hpp"
#include "../host/devbuf.hpp"
#include "kernels.hpp"

namespace acgpu {

struct CnfaTriDev {
    const uint32_t* bits = nullptr;    // [pairs][bw] (copied to LDS)
    const uint16_t* base = nullptr;    // [pairs]     (LDS)
    const uint8_t* uc = nullptr;       // [256]       (LDS) byte -> compact class
    const uint8_t* inv = nullptr;      // [256]       (LDS) compact class -> class
    const uint16_t* mc2 = nullptr;     // [pairs]     (LDS, only with shallow_matches)
    const uint32_t* st2 = nullptr;     // [pairs]     (global: records of shallow matches)
    const TriChild* child = nullptr;   // depth-3 nodes
    const uint32_t* repr3 = nullptr;   // repr, fail words into depth <= 2 tagged
    uint32_t pairs = 0, apair = 0, bw = 0, gshift = 0, n_used = 0, shallow_matches = 0, start_mlen = 0;
    uint32_t alen = 0, max_match_id = 0;
    uint32_t repr_words = 0, n_child = 0;   // sizes of repr3 / child (bounds-checked flavour)
};

// One match event of the walk: the records of `state`'s match list, `pre` records into chunk `ci`'s slice of the output.
struct TriEvent {
    uint32_t ci, pre, state, rel;   // rel: end position - 1 relative to the chunk grid origin of chunk ci (at - grid0 - ci * chunk)
};

struct CnfaTriTables {
    bool ready = false;
    CnfaTriDev dev;
    size_t lds_bytes = 0;
    DevBuf b_bits, b_base, b_uc, b_inv, b_mc2, b_st2, b_child, b_repr3;
};

hipError_t build_cnfa_tri(const CNfa& c, CnfaTriTables& out);
hipError_t launch_cnfa_tri_count(const CnfaTriTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s);

}  // namespace acgpu
