// Host-side tables of the LDS walk engine (see device/lds_walk.hip for the design).
#pragma once
#include <cstdint>
#include <vector>

#include "automaton.hpp"

namespace acgpu {

constexpr uint32_t kLwLdsBudget = 160 * 1024;   // the whole LDS of a gfx950 CU: one 1024-thread workgroup per CU
constexpr uint32_t kLwClsBytes = 512;           // the class map (u16 per byte value) occupies image bytes [0, 512); table offsets are relative to 512

// Three table flavours, one kernel skeleton (device/lds_walk.hip):
//   kLwFull    every state owns a class-compressed row: the literal   sid = trans[sid + class]   of src/dfa.rs:218-226 with
//              premultiplied ids.  handle = BYTE address of the state's row << 16 | match-list length (rows within 64 KiB,
//              lengths <= 4 095); class values are premultiplied by four.  No exceptions, no flags, matches counted by adding
//              the handles up (the low half of the sum is the count).  For the reference's small-set definitions.
//   kLwNarrow  dense rows + single-exception handles + exception chains: handle = base 8 | e 8 | da 16 bits
//   kLwWide    the same with base 10 | e 6 | da 16 bits (alphabets of at most 64 classes that want more than 254 rows)
// where da = BYTE address of deep[idx] (deep[] is the first table, so da = 4 * idx < 64 KiB goes into the LDS address
// as it stands -- an SDWA word select, no arithmetic).
enum LwFlavour : uint32_t { kLwNarrow = 0, kLwWide = 1, kLwFull = 2 };

// kLwFull handle, low half: bits 0..11 the length of the state's match list (<= 4 095: four handles are summed before the
// count is taken out of bits 0..13), bit 15 "sync": no occurrence can begin before the position this state was entered at
// and end behind it -- every node on the state's failure chain (the state included) below the root is a leaf of the trie --
// so that a non-overlapping iteration (FindIter, src/automaton.rs:857-936) is in its initial condition there whatever came
// before: the streaming chain of lds_emit.hip starts from such states.  The sum of four handles carries the flags in bits 15..17.
constexpr uint32_t kLwFullLenMask = 0x0FFFu, kLwFullSumMask = 0x3FFFu, kLwFullSync = 0x8000u, kLwFullSyncSum = 0x38000u;

struct LwHostTables {
    bool ok = false;
    std::vector<uint32_t> image;   // class map | tables   (copied to LDS address 0)
    uint32_t flavour = kLwNarrow;
    // Class of a byte.  Every class value carries rows_k, the dword offset of row 0 behind the class map (a multiple of 256,
    // so the low byte of the value is the class itself: the exception compare reads it with a byte select), hence
    //   row address = base * row_bytes + 4 * class_value      needs no further add.
    // computed_cls: class_value(b) = med3(int(b) + cc_add, cc_lo, cc_hi) -- the class map is a clamp of the byte onto the
    // range of bytes the patterns use (one class per byte of the range, one "other" class on either side): two VALU
    // operations instead of an LDS gather.  Otherwise the value is read from the u16 map at image byte 2 * b.
    bool computed_cls = false;
    int32_t cc_add = 0, cc_lo = 0, cc_hi = 0;
    uint32_t rows_k = 0;
    uint32_t row_bytes = 0;        // bytes per row: an odd number of dwords (bank spread), see lw_tables.cpp
    uint32_t rows_off = 0, nxt_off = 0, vhid_off = 0, mlen_off = 0;   // byte offsets behind the class map (deep[] is at 0)
    // kLwFull only: the match lists {pattern id, pattern length} of every match state in the reference's order
    // (src/dfa.rs:275-279), behind the rows; column `classes` of a state's row holds the byte offset of its list.
    // 0 = the lists did not fit LDS (the record fill of lds_walk.hip is then unavailable; counting is not affected)
    uint32_t mlist_off = 0;
    // kLwFull only: every match state is a leaf of the trie (no pattern is a proper prefix or a proper infix of another):
    // occurrences are then met in the order of their starts, which is what lets the streaming chain serve the leftmost kinds
    bool monotone = false;
    // kLwFull only: occurrences of these patterns never overlap and never share an end (every match state is a sync state
    // holding one pattern): the non-overlapping iteration of any match kind then reports every occurrence -- find_iter IS the
    // overlapping search (one-byte pattern sets: the reference's memchr / jetscii / teddy1 definitions)
    bool disjoint = false;
    uint32_t fm_addr = 0;          // 4 * first_match: handles whose da is >= this are match / multi / poison
    uint32_t virt_addr = 0;        // 4 * n_states:    ... >= this are multi / poison (not exact)
    uint32_t poison_row = 0, start = 0, first_match = 0, n_states = 0, n_idx = 0;
    uint32_t n_dense = 0, n_multi = 0, classes = 0;   // diagnostics
    bool wide() const { return flavour == kLwWide; }
};

void hid_order(const NNfa& n, std::vector<uint32_t>& order, std::vector<uint32_t>& sid2hid, uint32_t& first_match);
// false = the automaton does not fit the engine (too many states for LDS, a multi state at distance <= 1, ...)
// force_flavour: -1 = best fit (full, else narrow / wide); force_cls: -1 = computed when the class map allows it,
// 0 = LDS map, 1 = computed or fail (tests and A/B runs)
bool build_lw_host(const NNfa& n, const Dfa& d, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid,
                   uint32_t first_match, LwHostTables& out, int force_flavour = -1, int force_cls = -1);
// test hook: the kernel's walk (fast steps, flags, inline counts, exact redo) over one cold-started range on the CPU;
// returns the number of matches (start-state matches included); *redo_dwords = how many dwords took the exact path
uint64_t lw_emulate_count(const LwHostTables& t, const uint8_t* hay, size_t len, uint64_t* redo_dwords);
// test hook: the records k_lw_fill would write for hay[0..len) (kLwFull with match lists; false otherwise)
bool lw_emulate_records(const LwHostTables& t, const uint8_t* hay, size_t len, std::vector<acgpu_match>& out);
// test hook: the records the EVENT form (device/lds_emit.hip) produces for hay[0..len) with lane-chunks of `chunk` bytes (a power of
// two) and a warm-up of `halo` bytes: events per lane-chunk, scan, re-walk of the events in reverse order of arrival
bool lw_emulate_event_records(const LwHostTables& t, const uint8_t* hay, size_t len, uint32_t chunk, uint32_t halo, std::vector<acgpu_match>& out);
// share of the dwords on the exact path for pattern-like input (see lw_tables.cpp); prices the walk in the routing rule
double lw_estimate_redo(const LwHostTables& t);

}  // namespace acgpu
