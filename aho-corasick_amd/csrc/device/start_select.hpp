// Leftmost find_iter from a per-start candidate table (start_select.hip): launch interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acgpu.h"
#include "kernels.hpp"

namespace acgpu {

constexpr uint32_t kSsBlock = 1024;   // start positions per block; also the longest pattern the path serves

struct SsTables {
    const uint32_t* atab = nullptr;      // HotTables::atab (trie-only transitions, own flag in bit 31)
    const uint8_t* acls = nullptr;
    const uint32_t* own_pid = nullptr;   // HotTables::own_pid
    const uint32_t* plens = nullptr;     // pattern lengths
    uint32_t ashift = 0, root = 0, L = 0;   // L = longest pattern
    uint32_t n_states = 0;                  // trie nodes (rows of atab)
};

size_t start_select_work_bytes(uint64_t win_n, uint32_t L);
// One window of the span: start positions [win_lo, win_lo + win_n), win_lo = span_start + k * window.  The chain enters the
// first window at its first position and every later one at the exit offset of its predecessor (kept in `work`, which must
// therefore be the same buffer, on the same stream, for all windows of a span).  sc.counts / offsets / active / aoff / bsum /
// bact / totals sized for win_n / kSsBlock blocks; afterwards sc.totals[0] = selected matches of this window.
hipError_t launch_start_select(const SsTables& t, const uint8_t* hay, uint64_t span_end, uint64_t win_lo, uint64_t win_n,
                               int longest, void* work, bool first_window, const ScanScratch& sc, hipStream_t s);
// ... and their records, written to out[out_base ..) while they fit `cap` (same work / sc as the call above)
hipError_t launch_start_select_emit(const SsTables& t, uint64_t win_lo, uint64_t win_n, void* work, const ScanScratch& sc,
                                    uint64_t out_base, uint64_t cap, acgpu_match* out, hipStream_t s);

}  // namespace acgpu
