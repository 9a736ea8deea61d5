// GPU-side DFA transition-table fill (dfa_fill.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace acgpu {
struct NNfa;
// Fills host_trans[(s << s2) + k] (premultiplied ids, StartKind::Unanchored or Anchored layout) on the current device.
hipError_t device_fill_dfa(const NNfa& n, const uint8_t* classes, size_t alen, size_t s2, bool anchored,
                           uint32_t* host_trans);
}  // namespace acgpu
