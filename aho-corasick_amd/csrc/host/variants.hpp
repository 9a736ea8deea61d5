// Engine variants of ONE automaton (include/acgpu.h: acgpu_set_variant): the forms of the device engines that tests, the
// fuzzer and the A/B scripts select explicitly -- state of the automaton they are set on, not of the process: the library
// reads no environment variable for them (rounds 1-4 did, at 36 sites).  Every variant returns identical results.
#pragma once
#include <cstdint>
#include <cstring>

namespace acgpu {

struct Variants {
    // LDS walk (lds_walk.hip), read when the automaton is uploaded
    int32_t lw_flavour = -1;        // -1 best fit | 0 narrow | 1 wide | 2 one row per state (refused when the form does not fit)
    int32_t lw_cls = -1;            // -1 engine's choice | 0 LDS class map | 1 computed classes
    int32_t lw_lane_chunk = 0;      // bytes per lane-chunk (0: 512, 1 024 from 6 GiB shards on)
    int32_t lw_first = 1;           // automata with a one-row-per-state image are searched by the LDS walk first (0: the prefix filter, with routing)
    int32_t lw_events = 1;          // ... whose count walk notes its matches as events (lds_emit.hip); 0: count -> scan -> chunk fill
    // large-set filter (pfx_scan.hip)
    int32_t pfx_min_patterns = -1;  // -1: kPfxMinPatterns; sets of at least this many patterns use the large-set filter
    int32_t pfx_gate = 1;           // the L2-resident exact-prefix bit table in front of the 4-byte map
    int32_t pfx_tails = 2;          // tail records behind the long-prefix map (read when the tables are built, and per launch): 2 small subtrees | 1 chains only | 0 none
    int32_t pfx_key8 = 1;           // level 1 on the whole long prefix
    int32_t pfx_key8_roles = 12;    // producers of that kernel: 12 | 14
    int32_t pfx_key8_x2 = 1;        // ... probed at every other position (every pattern >= 9 bytes)
    int32_t pfx_short = 1;          // ... with one or two stragglers of 3..8 bytes compared in the producers' registers (0: the set as a whole; read at upload)
    // transition walks
    int32_t walk_literal = 0;       // 1: the contiguous-NFA walk as the reference loop verbatim (no LDS rows, no shallow skip)
    int32_t walk_tri = 1;           // 0: no shallow-skip kernels (cnfa_tri.hip / dfa_tri.hip)
    int32_t tri_events = 1;         // 0: the shallow-skip walks count -> scan -> re-walking fill instead of recording events
    // prefix filter / routing
    int32_t pf_classic = 0;         // 1: chunk counters + scan + fill instead of the event forms
    int32_t routing = 1;            // 0: an abandoned scan is not handed to another engine
    int32_t eo_fused = 1;           // the order pass's histogram inside the scan kernels, its totals reported by its last kernel (0: separate launches)
    // non-overlapping searches (capi_find.cpp) and the stream search
    int32_t start_table = 1;        // 0: never select from the per-start table
    int32_t ss_window_kib = 0;      // window of the per-start table (0: 256 MiB)
    int32_t find_iter_windows = 0;  // 1: force the windowed form
    int32_t find_iter_start_table = 0;   // 1: force the per-start table
    int32_t find_iter_disjoint = 1;      // 0: never serve find_iter by the overlapping search (pattern sets whose occurrences cannot overlap)
    int32_t stream_split = 0;       // 1: feed large chunks as two halves

    // name -> field (nullptr: unknown name)
    int32_t* field(const char* name) {
#define ACGPU_VARIANT(f) if (std::strcmp(name, #f) == 0) return &f;
        ACGPU_VARIANT(lw_flavour) ACGPU_VARIANT(lw_cls) ACGPU_VARIANT(lw_lane_chunk) ACGPU_VARIANT(lw_first) ACGPU_VARIANT(lw_events) ACGPU_VARIANT(pfx_min_patterns) ACGPU_VARIANT(pfx_gate)
        ACGPU_VARIANT(pfx_tails) ACGPU_VARIANT(pfx_key8) ACGPU_VARIANT(pfx_key8_roles) ACGPU_VARIANT(pfx_key8_x2) ACGPU_VARIANT(pfx_short) ACGPU_VARIANT(walk_literal)
        ACGPU_VARIANT(walk_tri) ACGPU_VARIANT(tri_events) ACGPU_VARIANT(pf_classic) ACGPU_VARIANT(routing) ACGPU_VARIANT(eo_fused) ACGPU_VARIANT(start_table)
        ACGPU_VARIANT(ss_window_kib) ACGPU_VARIANT(find_iter_windows) ACGPU_VARIANT(find_iter_start_table) ACGPU_VARIANT(find_iter_disjoint) ACGPU_VARIANT(stream_split)
#undef ACGPU_VARIANT
        return nullptr;
    }
};

}  // namespace acgpu
