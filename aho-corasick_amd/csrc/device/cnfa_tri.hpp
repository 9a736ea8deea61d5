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
    const uint8_t* mc2 = nullptr;      // [pairs]     (LDS, only with shallow_matches)
    const uint32_t* st2 = nullptr;     // [pairs]     (global: records of shallow matches)
    const TriChild* child = nullptr;   // depth-3 nodes
    const uint32_t* repr3 = nullptr;   // repr, fail words into depth <= 2 tagged
    uint32_t pairs = 0, apair = 0, bw = 0, gshift = 0, n_used = 0, shallow_matches = 0, start_mlen = 0;
    uint32_t alen = 0, max_match_id = 0;
    uint32_t repr_words = 0, n_child = 0;   // sizes of repr3 / child (bounds-checked flavour)
};

// Event buffer of one scan (count pass -> k_cnfa_tri_emit).
struct TriEvents {
    TriEvent* ev = nullptr;              // [max_segs * kTriSeg]
    uint32_t* seg_fill = nullptr;        // [max_segs]
    unsigned long long* ctr = nullptr;   // [0] segments handed out, [1] overflow flag (zeroed before the count pass)
    uint32_t max_segs = 0;
};
inline uint32_t tri_event_segments(uint64_t span_bytes) {   // one event per 64 haystack bytes, 64 Ki to 12 Mi events
    const uint64_t ev = span_bytes / 64 < (uint64_t(1) << 16) ? (uint64_t(1) << 16) : (span_bytes / 64 > (uint64_t(12) << 20) ? (uint64_t(12) << 20) : span_bytes / 64);
    return uint32_t(ev / kTriSeg);
}

struct CnfaTriTables {
    bool ready = false;
    CnfaTriDev dev;
    size_t lds_bytes = 0;
    DevBuf b_bits, b_base, b_uc, b_inv, b_mc2, b_st2, b_child, b_repr3;
};

hipError_t build_cnfa_tri(const CNfa& c, CnfaTriTables& out);
hipError_t launch_cnfa_tri_count(const CnfaTriTables& h, const ScanGeom& g, uint32_t* counts, const TriEvents* evs, hipStream_t s);
// Ordered records from the events: record k of an event goes to out[offsets[ci] + pre + k].  Reads the totals on the
// device: writes nothing when the events overflowed their buffer (ctr[1]) or the records do not fit `cap`.
hipError_t launch_cnfa_tri_emit(const CnfaTriTables& h, const uint32_t* plens, const ScanGeom& g, const TriEvents& evs,
                                const uint64_t* offsets, const uint64_t* totals, uint64_t cap, acgpu_match* out, hipStream_t s);

}  // namespace acgpu
