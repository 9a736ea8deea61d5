// Contiguous-NFA failure-link walk with an exact skip of the depth <= 2 regime (cnfa_tri.hip; host/cnfa_tri_tables.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/automaton.hpp"
#include "../host/cnfa_tri_tables.hpp"
#include "../host/devbuf.hpp"
#include "kernels.hpp"
#include "tri_kernel.hpp"
#include "cnfa_tri_step.hpp"

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
    __host__ __device__ void setup(TriWalk& f) const {
        f.child = child; f.repr3 = repr3; f.alen = alen; f.max_match = max_match_id; f.repr_words = repr_words;
    }
};

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
