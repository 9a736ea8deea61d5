// Merge of two ordered match-record streams (merge.hip): the overlapping search of a SPLIT pattern set (capi_overlap.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acgpu.h"

namespace acgpu {

// out[0 .. na + nb) = the records of a[] and b[] in the order of the reference's overlapping iterator: by end, then by
// decreasing pattern length (a state's match list: its own patterns, then those of its failure chain,
// src/nfa/noncontiguous.rs:466-523), then by pattern id (duplicates of one pattern text).  Both inputs are in that order
// already and share no record.  n_ab (device, [0] = na, [1] = nb) or the host values na / nb when n_ab == nullptr;
// totals (optional, device): [0] receives na + nb.
hipError_t launch_merge_records(const acgpu_match* a, const acgpu_match* b, uint64_t na, uint64_t nb, acgpu_match* out,
                                uint64_t* totals, hipStream_t s);

}  // namespace acgpu
