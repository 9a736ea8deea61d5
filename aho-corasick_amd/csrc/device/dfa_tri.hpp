// DFA transition walk with an exact skip of the depth <= 2 regime (dfa_tri.hip; host/dfa_tri_tables.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../host/automaton.hpp"
#include "../host/dfa_tri_tables.hpp"
#include "../host/devbuf.hpp"
#include "dfa_tri_step.hpp"
#include "kernels.hpp"
#include "tri_kernel.hpp"

namespace acgpu {

struct DfaTriDev {
    const uint32_t* bits = nullptr;    // [pairs][bw] (copied to LDS)
    const uint16_t* base = nullptr;    // [pairs]     (LDS)
    const uint8_t* uc = nullptr;       // [256]       (LDS) byte -> compact class
    const uint8_t* inv = nullptr;      // [256]       (LDS) compact class -> class
    const uint8_t* mc2 = nullptr;      // [pairs]     (LDS, only with shallow_matches)
    const uint32_t* st2 = nullptr;     // [pairs]     (global: records of shallow matches)
    const uint32_t* child = nullptr;   // depth-3 nodes (state ids)
    const uint32_t* trans3 = nullptr;  // transition table, targets of depth <= 2 tagged
    const uint32_t* moff = nullptr;    // the DFA's match-list offsets
    uint32_t pairs = 0, apair = 0, bw = 0, gshift = 0, n_used = 0, shallow_matches = 0, start_mlen = 0;
    uint32_t stride2 = 0, max_match_id = 0, trans_words = 0, n_child = 0;
    __host__ __device__ void setup(DfaTriWalk& f) const {
        f.child = child; f.trans3 = trans3; f.moff = moff; f.stride2 = stride2; f.max_match = max_match_id; f.trans_words = trans_words;
    }
};

struct DfaTriTables {
    bool ready = false;
    DfaTriDev dev;
    size_t lds_bytes = 0;
    DevBuf b_bits, b_base, b_uc, b_inv, b_mc2, b_st2, b_child, b_trans3;
};

// dev_moff: the uploaded DFA's match-list offsets (DevDfa::moff)
hipError_t build_dfa_tri(const NNfa& n, const Dfa& d, const uint32_t* dev_moff, DfaTriTables& out);
hipError_t launch_dfa_tri_count(const DfaTriTables& h, const ScanGeom& g, uint32_t* counts, const TriEvents* evs, hipStream_t s);
// Ordered records from the events (as launch_cnfa_tri_emit): record k of an event goes to out[offsets[ci] + pre + k].
hipError_t launch_dfa_tri_emit(const DfaTriTables& h, const DevAutomaton& a, const ScanGeom& g, const TriEvents& evs,
                               const uint64_t* offsets, const uint64_t* totals, uint64_t cap, acgpu_match* out, hipStream_t s);

}  // namespace acgpu
