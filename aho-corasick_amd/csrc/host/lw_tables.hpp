// Host-side tables of the LDS walk engine (see device/lds_walk.hip for the design).
#pragma once
#include <cstdint>
#include <vector>

#include "automaton.hpp"

namespace acgpu {

constexpr uint32_t kLwLdsBudget = 160 * 1024;   // the whole LDS of a gfx950 CU: one 1024-thread workgroup per CU
constexpr uint32_t kLwClsBytes = 256;           // the class map occupies image bytes [0, 256); table offsets are relative to 256

struct LwHostTables {
    bool ok = false;
    std::vector<uint32_t> image;   // class map | rows | deep | nxt | vhid | mlen   (copied to LDS address 0)
    uint32_t row_bytes = 0;        // bytes per row: an odd number of dwords (bank spread), see lw_tables.cpp
    bool wide = false;             // handle layout: false = base 8 | e 8 | idx 16 bits, true = base 10 | e 6 | idx 16
    uint32_t deep_off = 0, nxt_off = 0, vhid_off = 0, mlen_off = 0;   // byte offsets behind the class map
    uint32_t fm_addr = 0;          // deep_off + 4 * first_match
    uint32_t poison_row = 0, start = 0, first_match = 0, n_states = 0, n_idx = 0;
    uint32_t n_dense = 0, n_multi = 0, classes = 0;   // diagnostics
};

void hid_order(const NNfa& n, std::vector<uint32_t>& order, std::vector<uint32_t>& sid2hid, uint32_t& first_match);
// false = the automaton does not fit the engine (too many states for LDS, a multi state at distance <= 1, ...)
bool build_lw_host(const NNfa& n, const Dfa& d, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid,
                   uint32_t first_match, LwHostTables& out);
// test hook: the kernel's walk (fast steps, flags, exact redo) over one cold-started range on the CPU; returns the number
// of matches (start-state matches included); *redo_dwords = how many dwords took the exact path
uint64_t lw_emulate_count(const LwHostTables& t, const uint8_t* hay, size_t len, uint64_t* redo_dwords);
// share of the dwords on the exact path for pattern-like input (see lw_tables.cpp); prices the walk in the routing rule
double lw_estimate_redo(const LwHostTables& t);

}  // namespace acgpu
