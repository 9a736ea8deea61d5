// Pieces shared by the two prefix-filter kernels (pf_scan.hip: two-type 3-byte-key tables, level 3 inline;
// pfx_scan.hip: 4-byte-key table for large pattern sets, level 3 on dedicated verifier wavefronts).
#pragma once
#include <hip/hip_runtime.h>

#include "hot.hpp"

namespace acgpu {
namespace pfdev {

constexpr int kPfBlock = 1024;
constexpr int kPfWaves = kPfBlock / 64;
constexpr int kQueue = 128;           // per-wave survivor queues (drained in batches of 64)
constexpr int kEvBuf = 48;            // per-wave LDS event buffer (event modes), flushed from kEvFlush entries on
constexpr int kEvFlush = 24;
constexpr uint32_t kRowBytes = 1008;  // one wave-row: 63 lanes x 16 B of start positions (lane 63 only supplies
                                      // the 4-byte look-ahead of lane 62 and repeats as lane 0 of the next row)
constexpr uint32_t kTaskRows = 40;    // rows per wave task (5 iterations of kSets row pairs)
constexpr int kSets = 4;                // row-pair register sets in rotation (software pipeline depth kSets-1)
constexpr uint32_t kBitsBytes = 64 * 1024;  // level-1 Bloom table (static LDS at offset 0: no base add per gather)

struct PfArgs {
    const uint32_t* bits;   // level-1 Bloom table (global copy)
    const uint32_t* bits2;  // second Bloom table (global copy), kPfBits2Bytes
    const uint32_t* atab;   // trie-only transitions, class-compressed: [state][1 << ashift]
    const uint8_t* acls;    // its class map (global copy; the kernels stage it into LDS)
    uint32_t ashift;
    const uint32_t* own_cnt;
    const uint32_t* bits3;  // third table (global, L2-resident; nullptr = none): exact first four bytes, see hot.hpp
    uint32_t bits3_log2;
    uint32_t bits_bytes, root;
    uint64_t scan_lo;     // first start position that may begin an owned match (virtual)
    uint64_t row0;        // scan_lo rounded down to 16
    uint64_t hull_end;    // emit_hi rounded up to 16: no load touches bytes at or beyond it
    uint64_t n_tasks;
    // direct mode (events != nullptr): level 3 appends one event per (start, pattern end) instead of crediting the
    // chunk counters; the records are then ordered by k_ev_rank / k_ev_write without re-walking the haystack
    PfEvent* events;
    unsigned long long* ev_ctr;   // [0] events appended, [1] records they stand for, [2] != 0: scan abandoned (see route_*)
    uint64_t ev_cap;
    // Routing (route_cb != 0, event modes only): a wavefront that has handed X >= 2048 start positions to level 3 compares
    // the filter's cost model with that of the alternative engine the host has ready for this automaton --
    //     filter:  B / 5000 + X / 130        (B = bytes of the tasks it has started; level 1 streams at ~5 TB/s, level 3
    //                                          verifies ~130 G starts/s chip-wide: dependent L2 gathers)
    //     LDS transition walk:  B / 3200 * (1 + 3 min(1, 256 M / B))    (M = pattern ends found: dwords with a match
    //                                          take the exact path);     global-table DFA walk:  B / 450
    // -- and abandons the scan when the filter is predicted >= 25 % slower:  5000 X > route_cb B + route_cr min(B, 256 M).
    // It raises ev_ctr[2] and stops verifying; every wavefront decides from its own counters (on a stationary input they
    // all reach the same verdict within a few KiB).  The host (or, in the enqueue-only form, the caller) then repeats the
    // search with the other engine.  Exactness is unaffected: an abandoned scan's result is never used.
    // The per-wave counters live in LDS (PfWave::rt), not in registers: the row loop's register budget is untouched.
    uint32_t route_cb, route_cr;
    // Gate (gate != nullptr): the kernel runs only if *gate == gate_val -- the probe (k_pf_probe) decided on the device
    // which of the two filters scans this input, both are enqueued, one returns at once.
    const uint32_t* gate;
    uint32_t gate_val;
    // probe (k_pf_probe): the starts that reach level 3 are counted, not verified -- their dependent trie walks are what
    // makes the filter slow on the inputs the probe exists to detect, and made the probe itself take 0.19 ms
    uint32_t skip_verify;
    // pfx_scan.hip only: exact level 2, HotTables::pfx_map
    const uint4* xmap;
    uint32_t xmap_log2;
    uint32_t xdepth;      // prefix bytes level 2 compares exactly: 4 (xmap: two pairs per bucket) or 5..8 (one entry per bucket)
    const uint32_t* tails;   // chain-tail records behind the long-prefix map (hot.hpp: kPfxTailWords words each), or nullptr
    // short mode (hot.hpp: pfx_short_*): the stragglers the producers compare at every position -- first min(len, 4) bytes
    // (key / mask) there, every byte by the verifier
    uint32_t short_n;
    uint32_t short_lo[kPfxShortMax], short_hi[kPfxShortMax], short_len[kPfxShortMax], short_node[kPfxShortMax];
    // the order pass's histogram on the way (PfEoHist, hot.hpp; eo_bb == nullptr: none)
    unsigned long long* eo_bb;
    uint32_t* eo_slot;
    uint64_t eo_origin;
    uint32_t eo_shift;
};

// event `idx` of the list (key, cnt records) is counted in its bucket of end positions; the old count is its arrival slot
__device__ __forceinline__ void pf_eo_hist(const PfArgs& a, uint64_t key, uint32_t cnt, unsigned long long idx) {
    const uint64_t b = ((key >> 16) - 1 - a.eo_origin) >> a.eo_shift;
    a.eo_slot[idx] = uint32_t(atomicAdd(&a.eo_bb[b], (1ull << 32) | cnt) >> 32);
}

// Orders the queue traffic of one wavefront: LDS executes a wave's instructions in issue order, so the entries other
// lanes wrote are visible once the wave's own LDS operations have retired.  Deliberately NOT a workgroup fence: that
// would also wait (vmcnt) for the row pairs in flight and stall the software pipeline at every batch.
__device__ __forceinline__ void pf_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// level 3: exact verification of one start position: trie-only walk; every pattern end is credited to its chunk
// (classic mode) or recorded as an event {end, length, trie node} (event modes).  Events are collected in a small
// per-wave LDS buffer (slot from an LDS atomic) and appended to the global list kEvBuf at a time by flush_events: one
// global atomic per flush instead of two per event -- on a match-dense haystack the single global counter was the
// bottleneck of the whole scan (1.4 M events: 20 ms of serialized L2 atomics).  A lane that finds the buffer full
// appends its event directly.  Returns whether this lane recorded an event in LDS.
__device__ __forceinline__ void pf_append_event(const PfArgs& a, uint64_t key, uint32_t node, uint32_t cnt) {
    const unsigned long long idx = atomicAdd(&a.ev_ctr[0], 1ull);
    atomicAdd(&a.ev_ctr[1], static_cast<unsigned long long>(cnt));
    if (idx < a.ev_cap) {
        a.events[idx].key = key; a.events[idx].node = node; a.events[idx].cnt = cnt;
        if (a.eo_bb) pf_eo_hist(a, key, cnt, idx);
    }
}
__device__ __forceinline__ bool pf_verify(const PfArgs& a, const ScanGeom& g, uint32_t* counts, uint64_t v,
                                          PfEvent* ebuf, uint32_t* ecnt, const uint8_t* s_acls) {
    uint32_t s = a.root;
    bool buffered = false;
    for (uint64_t at = v; at < g.emit_hi; at++) {
        ACGPU_HAY_CHECK(g, at, 1);
        const uint32_t e = a.atab[(s << a.ashift) | s_acls[g.hay16[at]]];
        if (e == 0) break;
        s = e & 0x7FFFFFFFu;
        if ((e >> 31) && at >= g.emit_lo) {
            const uint32_t cnt = a.own_cnt[s];
            if (a.events) {
                const uint64_t key = ((at + 1 - g.base_mis) << 16) | (0xFFFFull - (at + 1 - v));   // end asc, then longer first
                const uint32_t slot = atomicAdd(ecnt, 1u);   // ds_add_rtn_u32
                if (slot < uint32_t(kEvBuf)) { ebuf[slot].key = key; ebuf[slot].node = s; ebuf[slot].cnt = cnt; buffered = true; }
                else pf_append_event(a, key, s, cnt);
            } else {
                atomicAdd(&counts[(at - g.grid0) / g.chunk], cnt);
            }
        }
    }
    return buffered;
}

}  // namespace pfdev
}  // namespace acgpu
