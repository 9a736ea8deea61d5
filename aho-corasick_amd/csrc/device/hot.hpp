// LDS-resident fast path for the Standard/unanchored full DFA (hot_scan.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../host/automaton.hpp"
#include "../host/lw_tables.hpp"
#include "../host/variants.hpp"
#include "kernels.hpp"

namespace acgpu {

// Device tables of the fast path.  States are renumbered ("hid"):
//   0 = DEAD, then the non-match states in breadth-first order (start state first, then distance 1,
//   2, ...), then the match states in breadth-first order.  Hence
//     is_match(hid)  <=>  hid >= first_match
// `tab` = the 256-wide u16 transition table over hids ("256-wide transition table"; the fill kernel k_hot_fill builds
// its LDS rows from it: the n_hot shallowest non-match states).
constexpr uint32_t kPfxShortMax = 2;   // stragglers of 3..8 bytes compared in registers beside a dictionary of >= 9-byte patterns (short mode, host/pf_tables.hpp)
struct HotTables {
    Variants var;               // the automaton's engine variants at upload (host/variants.hpp)
    bool ready = false;
    uint32_t n_states = 0;      // number of hids
    uint32_t first_match = 0;
    uint32_t n_hot = 0;
    uint32_t start = 0;         // hid of the unanchored start state
    uint16_t* tab = nullptr;    // [n_states][256] (global, L2-resident for small automata)
    uint32_t* hid2sid = nullptr;  // [n_states] premultiplied DFA state id (for match-list lookup)

    // --- LDS walk engine (lds_walk.hip): the whole automaton in LDS, one row per state or dense rows + exception handles ---
    bool lw_ready = false;
    uint32_t lw_route_cb = 124;     // the walk's price in the prefix filter's routing rule (lds_walk.hip: build_lw_tables)
    uint32_t* lw_image = nullptr;   // LDS image: class map (512 B) | tables (host/lw_tables.hpp)
    uint32_t lw_image_bytes = 0;
    LwHostTables lw;                // flavour, class form and table offsets of the image (its host copy is dropped after the upload)

    // --- prefix-filter engine (pf_scan.hip) ---
    bool pf_ready = false;
    uint8_t* acls = nullptr;        // [256] class map of atab (0 = byte on no trie edge)
    uint32_t ashift = 8;            // log2(entries per atab row)
    uint32_t* atab = nullptr;       // [n_states][1 << ashift] anchored (trie-only) transitions: child hid | 1<<31 if the
                                    // child ends a pattern; 0 = no trie edge
    uint32_t* own_cnt = nullptr;    // [n_states] number of patterns ending exactly in this trie node
    uint32_t* own_pid = nullptr;    // [n_states] the lowest pattern id among them (start_select.hip)
    // first-level Bloom table, probed at every other haystack position q (pf_scan.hip) with the 5-byte window b[q..q+4]:
    //   word byte-address = (mul24(b[q+1] | b[q+2]<<8 | b[q+3]<<16, kPfHashMul) >> 16) & (pf_bits_bytes-1) & ~3
    //   survivor <=> bit 31-(b[q] & 31) set (a pattern may start at q)  OR  bit 31-(b[q+4] & 31) set (at q+1)
    // built from every trie path of depth <= 4 (hot_scan.hip), wildcarding the bytes short patterns do not have,
    // so the filter has no false negatives.
    uint32_t* pf_bits = nullptr;
    uint32_t pf_bits_bytes = 0;
    // second table: same construction and probe as the first one under an unrelated hash (pf_hash2), kPfBits2Bytes;
    // consulted only for the survivors of the first table
    uint32_t* pf_bits2 = nullptr;
    // large pattern sets (> kPfExact2Patterns): the second table instead holds one entry per pattern keyed by its TRUE
    // start (word = hash2(b[v..v+2]), bit 31-(b[v+3] & 31)) and a level-1 survivor probes it twice, once per start it
    // stands for: half the fill and no "either start" pass, worth the second gather once the tables saturate
    bool pf_exact2 = false;
    bool pf_fold = false;   // keys of both tables are case-folded (| 0x20 per byte): the kernel folds its row registers (host/pf_tables.cpp)
    // mid-size and large sets (>= kPfBits3Patterns, no pattern shorter than 3 bytes): a third, L2-resident bit table
    // keyed by the exact first four bytes of every pattern (2^pf_bits3_log2 bits, ~1 % fill; 3-byte patterns enter with
    // all 256 fourth bytes).  Level 3 consults it with one gather per candidate start before walking the trie, which
    // otherwise costs three to four dependent L2 gathers for every false candidate the LDS tables let through.
    uint32_t* pf_bits3 = nullptr;
    uint32_t pf_bits3_log2 = 0;
    // --- large pattern sets (pfx_scan.hip): 1 Mi-bit blocked Bloom table keyed by the first FOUR bytes of every pattern
    bool pfx_ready = false;
    uint32_t* pfx_bits = nullptr;   // kPfxBitsBytes
    uint32_t* pfx_bits8 = nullptr;  // the same table keyed by the whole long prefix: pfx_depth = 5..8 bytes, zero-padded to eight (pfx_hash8)
    uint32_t* pfx_bits8x2 = nullptr;   // ... probed at every other position (pfx_x2_mask; only when every pattern has >= 9 bytes)
    // exact level 2 of that engine: open-addressing hash map  first four bytes -> trie node at depth 4 (hid | 1<<31 if a
    // pattern ends there), buckets of two {key, value} pairs (one 16-byte gather); value 0 = empty slot; bit 30 of the
    // first value = "a key whose home is this bucket was placed further on" (kPfxMapOverflow): a lookup that does not
    // find its key continues with the next bucket only then (0.1 % of the buckets at load 1/8)
    uint4* pfx_map = nullptr;
    uint32_t pfx_map_log2 = 0;      // log2(number of buckets)
    // exact level 2 on a LONGER prefix when every pattern has one (pfx_depth = min(8, shortest pattern) > 4): buckets of ONE
    // entry {bytes 0..3, bytes 4..depth-1 (zero-padded), trie node at that depth | own flag | overflow, 0}.  Natural
    // text is full of true 4-byte prefixes of dictionary words (7 % of the positions of English text for 5 000 long
    // words) and almost free of 8-byte ones
    uint4* pfx_map8 = nullptr;
    uint32_t pfx_map8_log2 = 0;
    uint32_t* pfx_tails = nullptr;   // chain-tail records behind pfx_map8 (kPfxTailWords words each), or nullptr
    uint32_t pfx_depth = 4;
    uint32_t pfx_prefixes = 0;      // distinct 4-byte prefixes in the Bloom table
    uint32_t n_patterns = 0;
    // short mode (PfHostTables::short_n): the long-key tables hold the patterns of >= 9 bytes, these are the others
    uint32_t pfx_short_n = 0;
    uint32_t pfx_short_lo[kPfxShortMax] = {}, pfx_short_hi[kPfxShortMax] = {}, pfx_short_len[kPfxShortMax] = {}, pfx_short_node[kPfxShortMax] = {};
    bool pfx4_complete = true;      // the 4-byte tables know every pattern (false: a straggler of three bytes)
    ~HotTables() {
        if (pfx_bits) (void)hipFree(pfx_bits);
        if (pfx_bits8) (void)hipFree(pfx_bits8);
        if (pfx_bits8x2) (void)hipFree(pfx_bits8x2);
        if (pfx_map8) (void)hipFree(pfx_map8);
        if (pfx_tails) (void)hipFree(pfx_tails);
        if (pfx_map) (void)hipFree(pfx_map);
        if (lw_image) (void)hipFree(lw_image);
        if (pf_bits3) (void)hipFree(pf_bits3);
        if (pf_bits) (void)hipFree(pf_bits);
        if (pf_bits2) (void)hipFree(pf_bits2);
        if (tab) (void)hipFree(tab);
        if (hid2sid) (void)hipFree(hid2sid);
        if (atab) (void)hipFree(atab);
        if (acls) (void)hipFree(acls);
        if (own_cnt) (void)hipFree(own_cnt);
        if (own_pid) (void)hipFree(own_pid);
    }
};

constexpr size_t kPfMaxStates = size_t(1) << 20;   // trie-table budget of the prefix filter (1 KiB per state)
constexpr size_t kPfMaxPatterns = 131072;          // beyond this the 64 KiB Bloom table passes too much

constexpr uint32_t kPfHashMul = 0x9E3779u;   // 24-bit golden-ratio multiplier
constexpr uint32_t kPfHashMul2 = 0xC2B2AFu;  // second table: an unrelated odd 24-bit multiplier
constexpr uint32_t kPfExact2Patterns = 24000;
constexpr uint32_t kPfBits3Patterns = 4096;
__host__ __device__ __forceinline__ uint32_t pf_hash3(uint32_t key4, uint32_t log2_bits) { return (key4 * 0x9E3779B1u) >> (32u - log2_bits); }
constexpr uint32_t kPfBits2Bytes = 64 * 1024;
__host__ __device__ __forceinline__ uint32_t pf_hash2(uint32_t key) { return ((key & 0xFFFFFFu) * kPfHashMul2) >> 16; }
__host__ __device__ __forceinline__ uint32_t pf_hash(uint32_t key) {
    // bits 16..31 of the 24x24-bit product: every key byte reaches them (the high half of the 48-bit product
    // all but ignores the low key byte: 11x the false-positive rate on printable text)
    return ((key & 0xFFFFFFu) * kPfHashMul) >> 16;  // v_mul_u32_u24 + WORD_1 select
}

// pfx_scan.hip: one 32-bit hash of the 4-byte window picks the table word and three bits inside it (a "blocked" Bloom
// filter: one LDS gather tests all three).  h = 24x24-bit product of the low three bytes + the high byte times another odd
// constant (v_mul_u32_u24 + v_mad_u32_u24).
constexpr uint32_t kPfxBitsBytes = 128 * 1024;
constexpr uint32_t kPfxMinPatterns = 10000;   // below this the two-type tables of pf_scan.hip are the faster filter
__host__ __device__ __forceinline__ uint32_t pfx_hash(uint32_t w) {
    return (w & 0xFFFFFFu) * 0x9E3779u + (w >> 24) * 0x85EBCBu;
}
// word of the table: the hash folded once, so that the index also depends on the upper key bits (as a byte address:
// v_lshrrev + v_bitop3 on the device)
__host__ __device__ __forceinline__ uint32_t pfx_word_addr(uint32_t h) { return (h ^ (h >> 15)) & (kPfxBitsBytes - 4); }
__host__ __device__ __forceinline__ uint32_t pfx_word(uint32_t h) { return pfx_word_addr(h) >> 2; }
// the three bits of the word a key owns, MSB-first like the other tables (tested as word << sel, sign bit): selected by
// the low five bits of BYTES 0, 2 and 3 of the hash -- the device shifts by an SDWA byte operand (the hardware takes the
// low five bits of the selected byte), no separate shift to extract a selector.  2.97 % of random probes pass at
// 100 000 patterns (bits 27.., 22.., 17.. of the hash: 2.70 %, for three more operations per position).
__host__ __device__ __forceinline__ uint32_t pfx_mask(uint32_t h) {
    return (0x80000000u >> (h & 31u)) | (0x80000000u >> ((h >> 16) & 31u)) | (0x80000000u >> ((h >> 24) & 31u));
}

// Level 1 keyed by EIGHT bytes (k_pfx_count<true, ..., kKey8>: sets whose shortest pattern has >= 8 bytes -- the reference's
// dictionaries): the 24-bit chunks bytes 0-2, bytes 3-5 and bytes 6-7 of the window times three odd constants
// (v_alignbit + v_mul_u32_u24 with a WORD_1 operand + two v_mad_u32_u24); word and bits of the table as for pfx_hash.
__host__ __device__ __forceinline__ uint32_t pfx_hash8(uint32_t lo, uint32_t hi) {
    const uint32_t mid = ((hi << 8) | (lo >> 24)) & 0xFFFFFFu;
    return (lo & 0xFFFFFFu) * 0x9E3779u + mid * 0x85EBCBu + (hi >> 16) * 0xC2B2AFu;
}
// The same level 1 probing only the ODD offsets p of a lane's 16 bytes (sets whose shortest pattern has >= 9 bytes): one
// hash and one gather per p -- key = b[p+1..p+8] -- serve the TWO start positions p and p+1, as in the two-type filter of
// pf_scan.hip.  Every 9-byte prefix P of the trie is in the table twice: "type 0" under the key P[1..8] with the bits
// selected by P[0] (tested with b[p]: a pattern may start at p), "type 1" under the key P[0..7] with the bits selected by
// P[8] (tested with b[p+9]: at p+1).  Two bits per entry, MSB-first: (sel & 31) and ((sel + a byte of the hash) & 31) --
// byte 0 of the hash for type 0, byte 2 for type 1; on the device the sums and the shifts take their operands as SDWA bytes.
__host__ __device__ __forceinline__ uint32_t pfx_x2_mask(uint32_t h, uint32_t sel, int type) {
    const uint32_t hb = type == 0 ? (h & 0xFFu) : ((h >> 16) & 0xFFu);
    return (0x80000000u >> (sel & 31u)) | (0x80000000u >> ((sel + hb) & 31u));
}
constexpr uint32_t kPfxKey8MaxPrefixes = 200000;   // beyond this the 1 Mi-bit table is too full to be worth the longer key

// Chain tails of the long-prefix map (round 4).  Below most nodes of a dictionary's depth-8 prefixes the trie is a CHAIN: one
// child per node down to a single leaf that ends the only pattern(s) of the subtree ("internat" -> "ional").  Level 3 walked
// that chain one dependent trie-row gather per byte.  For such nodes (no pattern ends before the leaf, chain of 1..16
// bytes, one byte per edge) the map entry's fourth word names a 32-byte tail record {the chain's bytes, the leaf's trie
// node, length | own count << 8}: level 3 becomes ONE gather of the record beside the 16-byte haystack gather and a masked
// compare.  Everything else keeps the walk.  Record: words 0-3 the chain's bytes, 4 the leaf's trie node, 5 length | own
// count << 8, 6 the prefix node itself (a hit within 16 bytes of the span's end takes the walk from there), 7 unused.
// Round 6: the same for nodes with a SMALL subtree -- up to kPfxTailMaxRecs pattern ends at or below the node, all within 16
// bytes of it ("consider" -> "", "able", "ably", "ation", "ations", "ed", "ing", "s"): one record per pattern end, consecutive,
// word 7 = the number of records that follow; the map entry's fourth word carries kPfxTailMulti when there are several (they
// are compared in the batches of the walks, a lane loops over its node's records: neighbours in one or two cache lines,
// where the walk gathers one trie row -- an L2 miss -- per byte).  Word 6 carries the node's own-pattern flag in bit 31.
constexpr uint32_t kPfxTailWords = 8;
constexpr uint32_t kPfxTailMaxLen = 16;
constexpr uint32_t kPfxTailMaxRecs = 8;
constexpr uint32_t kPfxTailMulti = 1u << 30;
constexpr uint32_t kPfxMapOverflow = 1u << 30;
__host__ __device__ __forceinline__ uint32_t pfx_map8_bucket(uint32_t lo, uint32_t hi, uint32_t log2_buckets) {
    return ((lo * 0x9E3779B1u + hi * 0x85EBCA77u) * 0xC2B2AE35u) >> (32u - log2_buckets);
}
__host__ __device__ __forceinline__ uint32_t pfx_map_bucket(uint32_t key4, uint32_t log2_buckets) { return (key4 * 0x9E3779B1u) >> (32u - log2_buckets); }

// One level-3 event of the prefix filter: a (start, pattern end) pair.  key = end << 16 | 0xFFFF - length orders the
// events exactly like the reference's overlapping iterator (pf_scan.hip); node = the trie node (hid) whose `cnt` own
// patterns end there.
struct PfEvent { uint64_t key; uint32_t node; uint32_t cnt; };

hipError_t build_hot_tables(const NNfa& n, const Dfa& d, const Variants& var, HotTables& out);
hipError_t build_lw_tables(const NNfa& n, const Dfa& d, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid,
                           uint32_t first_match, HotTables& out);
// classic mode: per-chunk match counts.  Direct mode (events != nullptr): level 3 appends {end, length, node} events
// (at most ev_cap are stored; ev_ctr[0] counts all of them, ev_ctr[1] their records) and `counts` is not touched.
// Routing coefficients of the prefix filter's cost model (pf_scan.hip, PfArgs::route_*): cb == 0 disables it.
// The order pass's histogram done by the scan itself (event_order.hip, fused chain): a wavefront that appends an event to the
// list also bumps the word of the event's bucket and stores the old event count -- the arrival slot -- beside the event.
struct PfEoHist {
    unsigned long long* bb = nullptr;   // [buckets] records | events << 32 (zero on entry); nullptr = no histogram
    uint32_t* slot = nullptr;           // [ev_cap]
    uint64_t origin = 0;                // end position - 1 - origin = offset into the bucket grid
    uint32_t shift = 0;                 // bucket = 2^shift end positions
};
struct PfRoute {
    uint32_t cb = 0, cr = 0;
    PfEoHist hist;
    // large-set kernel only: device scratch for its global hit list (level 3 as a second pass), pfx_hit_work_bytes(span);
    // force_pfx: run the large-set kernel whatever the pattern count (the two-type filter abandoned this input)
    void* hit_work = nullptr;
    size_t hit_work_bytes = 0;
    bool force_pfx = false;
    // gate (device word written by the probe, launch_pf_probe): the scan kernel runs only if *gate == gate_val
    const uint32_t* gate = nullptr;
    uint32_t gate_val = 0;
};
// Probe of the two-type filter (pf_scan.hip): 256 wavefronts run its levels 1-3 over 8 KB samples spread over the shard
// and the last one to finish applies the routing rule of PfRoute to their totals: *decision = 1 if the filter would
// abandon this input (the alternative engine should scan it), else 0.  probe_ctr: 8 zeroed 64-bit words (the probe
// leaves them zeroed).  ~10 us; only used while an automaton's recent scans were abandoned (capi_overlap.cpp).
hipError_t launch_pf_probe(const HotTables& h, const ScanGeom& g, PfRoute route, uint32_t* decision, unsigned long long* probe_ctr, hipStream_t s);
inline PfRoute pf_route(uint32_t cb, uint32_t cr) { PfRoute r; r.cb = cb; r.cr = cr; return r; }
inline PfRoute pf_route_to_lds_walk(const HotTables& h) { return pf_route(h.lw_route_cb, 762); }   // alternative = LDS transition walk (HotTables::lw_ready)
inline PfRoute kPfRouteToDfaWalk() { return pf_route(1675, 0); }     // alternative = global-table DFA walk
inline PfRoute kPfRouteToLargeSet() { return pf_route(300, 0); }     // alternative = large-set filter with its second-pass level 3 (measured: 200..300 route dictionary text, 450 decides too late)
constexpr size_t kPfCtrWords = 4;                // ev_ctr: [0] events, [1] records, [2] scan abandoned, [3] spare
hipError_t launch_pf_count(const HotTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s, void* events = nullptr,
                           unsigned long long* ev_ctr = nullptr, uint64_t ev_cap = 0, PfRoute route = PfRoute());
// the same contract for large pattern sets (pfx_scan.hip; no routing: nothing faster exists for those automata), and the
// dispatcher every caller uses
hipError_t launch_pfx_count(const HotTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s, void* events = nullptr,
                            unsigned long long* ev_ctr = nullptr, uint64_t ev_cap = 0, void* hit_work = nullptr, size_t hit_work_bytes = 0,
                            const uint32_t* gate = nullptr, uint32_t gate_val = 0, PfEoHist hist = PfEoHist());
size_t pfx_hit_work_bytes(uint64_t span_bytes);
bool pf_uses_large_set(const HotTables& h, const PfRoute& route);   // which of the two filters launch_pf_any runs
hipError_t launch_pf_any(const HotTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s, void* events = nullptr,
                         unsigned long long* ev_ctr = nullptr, uint64_t ev_cap = 0, PfRoute route = PfRoute());
size_t pf_event_bytes();
// Large result sets (event_order.hip): the events grouped by 2 KiB bucket of end positions (histogram, scan, scatter),
// ordered inside every bucket, records written -- hand-written, O(n), sizes read on the device: does nothing unless
// min_events < totals[1] <= max_events and totals[0] <= max_records.  span_begin: the haystack offset the shard's
// ownership starts at; work: event_order_work_bytes(...) bytes of device scratch; done_totals (enqueue-only form):
// totals[1] is set to 0 when this pass delivered the records.
size_t event_order_work_bytes(uint64_t max_events, uint64_t max_records, uint64_t span_bytes);
// the first event_order_zero_bytes(...) bytes of `work` must be zero when the pass starts: k_eo_zero does it, unless the
// caller did (zeroed = true: launch_pf_event_write takes the region along)
size_t event_order_zero_bytes(uint64_t max_events, uint64_t max_records, uint64_t span_bytes);
// log2 of the bucket size the pass takes for these bounds (host rule, tests/test_engine_plan.py): 2 KiB, or more when the
// events are few for the span -- at least max_events / 4 buckets, at most 16 MiB each; 2 KiB from 2^31 records on
uint32_t event_order_shift(uint64_t max_events, uint64_t max_records, uint64_t span_bytes);
// Fused chain: the scan kernels did the histogram (PfRoute::hist = event_order_hist(...) with the same bounds; the zero
// region was zero when the scan started) and nothing copied their counters out: the pass reads ctr[0] events, ctr[1]
// records, ctr[2] abandoned, serves ANY number of events up to max_events (min_events = 0), and its last workgroup writes
// totals = {records, delivered ? 0 : UINT64_MAX} (device; host_totals: the same and the number of events, page-locked host
// memory or nullptr; then `seq` at word 3) and zeroes ctr[0..2].  launch_event_order_zero re-arms the zero region for the next call.
struct EoFused {
    unsigned long long* ctr = nullptr;
    uint64_t* totals = nullptr;
    uint64_t* host_totals = nullptr;   // [4]
    uint64_t seq = 0;                  // stored at host_totals[3] after the three words (release, system scope): a host thread may poll it
};
PfEoHist event_order_hist(uint64_t max_events, uint64_t max_records, uint64_t span_begin, uint64_t span_bytes, void* work);
hipError_t launch_event_order_zero(void* work, size_t bytes, hipStream_t s);
hipError_t launch_event_order_emit(const HotTables& h, const DevAutomaton& a, const void* events, const uint64_t* totals,
                                   uint64_t min_events, uint64_t max_events, uint64_t max_records, uint64_t span_begin,
                                   uint64_t span_bytes, void* work, acgpu_match* out, hipStream_t s, uint64_t* done_totals = nullptr,
                                   bool zeroed = false, const EoFused* fused = nullptr);
hipError_t launch_pf_event_rank(const void* events, const unsigned long long* ev_ctr, uint32_t ev_cap, uint32_t* rank,
                                uint64_t* totals, uint32_t n_hint, hipStream_t s);
hipError_t launch_pf_event_write(const HotTables& h, const DevAutomaton& a, const void* events, unsigned long long* ev_ctr,
                                 uint32_t ev_cap, uint32_t* rank, const uint64_t* totals, uint64_t out_cap,
                                 acgpu_match* out, hipStream_t s, void* also_zero = nullptr, size_t also_zero_bytes = 0);
hipError_t launch_hot_count(const HotTables& h, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts,
                            hipStream_t s);
// record fill from the LDS image of the one-row-per-state form (lds_walk.hip: k_lw_fill); same contract as launch_hot_fill
bool lw_fill_supported(const HotTables& h);
// (ev_overflow / gen: the fill runs only if *ev_overflow == gen -- a slab of the count walk's events overflowed)
hipError_t launch_lw_fill(const HotTables& h, const ScanGeom& g, const uint64_t* active, const uint64_t* totals, uint64_t cap,
                          uint64_t max_waves, const uint64_t* aoff, acgpu_match* out, hipStream_t s,
                          const uint32_t* ev_overflow = nullptr, uint32_t gen = 0,
                          // (fine_off: instead of active / aoff -- every chunk of g is filled, chunk ci at record fine_off[ci * stride] of n_fine offsets)
                          const uint64_t* fine_off = nullptr, uint32_t stride = 1, uint64_t n_fine = 0);
// Event form of the one-row-per-state walk (lds_emit.hip): the count walk notes every dword that gained a record as a 16-byte
// event in the slab of its task (64 lane-chunks), k_lw_ev_emit turns the slabs into ordered records without a second walk.
// lw_events_chunk: the lane-chunk the scan geometry must be made with (ScanGeom::chunk; counts / offsets are per lane-chunk),
// 0 = the form does not apply.  lw_events_sizes: its scratch for a geometry.  overflow: one 32-bit word, zeroed when it is
// made; a task with more events than its slab holds stores `gen` there (a new value per call), launch_lw_ev_emit then
// writes nothing and the caller fills the records with launch_lw_fill.
struct LwEvSizes { uint64_t n_tasks = 0; uint32_t slab_events = 0; size_t ev_bytes = 0, task_n_bytes = 0; };
uint32_t lw_events_chunk(const HotTables& h, uint32_t halo, uint64_t span_bytes);
LwEvSizes lw_events_sizes(const ScanGeom& g);
hipError_t launch_lw_count_ev(const HotTables& h, const ScanGeom& g, uint32_t* counts, void* events, uint32_t* task_n, uint32_t* overflow,
                              uint32_t gen, hipStream_t s);
// the scan of the event form: ONE workgroup over the tasks' record counts (the count walk left them behind task_n) -> the tasks'
// record offsets, totals = {records, non-empty lane-chunks} on the device and, if given, in page-locked host memory (+ *extra32)
hipError_t launch_lw_task_scan(const ScanGeom& g, uint32_t* task_n, uint64_t* totals, uint64_t* host_totals, const uint32_t* extra32, hipStream_t s);
hipError_t launch_lw_ev_emit(const HotTables& h, const ScanGeom& g, const void* events, const uint32_t* task_n, const uint32_t* overflow,
                             uint32_t gen, const uint32_t* counts, const uint64_t* totals, uint64_t cap, acgpu_match* out, hipStream_t s,
                             // (enqueue-only form: report[0] <- records; report[1] <- 0, or UINT64_MAX on an overflow nobody fills)
                             uint64_t* report = nullptr, bool report_fail = false);
bool hot_fill_supported(const HotTables& h, const ScanGeom& g);
hipError_t launch_hot_fill(const HotTables& h, const DevAutomaton& a, const ScanGeom& g, const uint64_t* active,
                           const uint64_t* totals, uint64_t cap, uint64_t max_waves, const uint64_t* aoff,
                           acgpu_match* out, hipStream_t s);

}  // namespace acgpu
