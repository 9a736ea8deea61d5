// Which engine a search starts with, which one it may be handed to, and what the adaptive hints of an automaton change
// about that -- as PURE functions of a handful of facts, so that the routing rules of capi_overlap.cpp (overlapping_impl and the
// enqueue-only form share them) can be tabulated and tested on the host (tests/test_engine_plan.py through
// acgpu_test_engine_plan).  Every engine returns identical results; the plan only decides cost.
#pragma once
#include <cstddef>
#include <cstdint>

namespace acgpu {

// (numbering of device/kernels.hpp: EngineId, + the prefix filter's other kernel, reported as the prefix filter)
constexpr uint32_t kPlanDfaWalk = 1, kPlanCnfaWalk = 2, kPlanLdsWalk = 3, kPlanPrefixFilter = 4, kPlanLargeSetFilter = 100;

struct EngineFacts {
    bool has_dfa = false;        // the device holds a full DFA of the automaton (its own, or derived at upload)
    bool pf_ready = false;       // prefix-filter tables (no empty pattern, <= 131 072 patterns, <= 2^20 states)
    bool lw_ready = false;       // the automaton fits the LDS walk
    bool lw_full = false;        // ... in its one-row-per-state form with the match lists in LDS: the reference's walk word for word,
                                 // the same speed whatever the match density, records from its events (device/lds_emit.hip)
    bool pfx_ready = false;      // large-set filter tables (>= 256 patterns, every pattern >= 4 bytes)
    size_t min_pattern_len = 0;
    int want = 0;                // acgpu_config.engine as the pipelines test it: 0 auto, 1 transition walk, 2 LDS walk, 3 prefix filter
    bool routing = true;         // variant `routing`
};

struct EnginePlan {
    uint32_t first = 0;          // engine of the first scan; 0 = the requested engine is unavailable (ACGPU_ERR_INVALID_ARGUMENT)
    uint32_t alternative = 0;    // engine an abandoned prefix-filter scan is handed to (automatic choice only); 0 = none
};

inline EnginePlan plan_engines(const EngineFacts& f) {
    EnginePlan p;
    p.first = f.has_dfa ? kPlanDfaWalk : kPlanCnfaWalk;
    if (f.has_dfa) {
        // small automata: the transition walk from LDS first.  The prefix filter is 15 % faster on match-free input and up to
        // 20x slower where the patterns occur (profiles/r06_call_timelines_before.txt: 1.5 ms of a 256 MiB scan before its
        // routing rule gives up); the walk does not care
        if (f.want == 0 && f.lw_ready && f.lw_full && f.min_pattern_len > 0) p.first = kPlanLdsWalk;
        else if ((f.want == 0 || f.want == 3) && f.pf_ready) p.first = kPlanPrefixFilter;
        // (with an empty pattern every state is a match state: the LDS walk is not offered automatically)
        else if (((f.want == 0 && f.min_pattern_len > 0) || f.want == 2) && f.lw_ready) p.first = kPlanLdsWalk;
    }
    if ((f.want == 2 && p.first != kPlanLdsWalk) || (f.want == 3 && p.first != kPlanPrefixFilter)) { p.first = 0; return p; }
    if (p.first == kPlanPrefixFilter && f.want == 0 && f.routing) {
        if (f.lw_ready && f.min_pattern_len > 0) p.alternative = kPlanLdsWalk;
        else if (f.pfx_ready) p.alternative = kPlanLargeSetFilter;   // (automata too large for LDS)
        else if (f.has_dfa) p.alternative = kPlanDfaWalk;
    }
    return p;
}

// What the hints of the automaton make of a prefix-filter scan that has an alternative (spans of kProbeMinSpan and more,
// and only while the first kernel is the two-type filter): take the alternative unasked, ask the probe first, or just scan.
enum class PfStart : uint32_t { Scan = 0, Probe = 1, TakeAlternative = 2 };
constexpr uint64_t kPlanProbeMinSpan = uint64_t(16) << 20;
inline PfStart plan_pf_start(const EnginePlan& p, int probe_skip, int route_hint, uint64_t span_bytes, bool first_kernel_is_large_set) {
    if (p.first != kPlanPrefixFilter || !p.alternative || span_bytes < kPlanProbeMinSpan || first_kernel_is_large_set) return PfStart::Scan;
    if (probe_skip > 0) return PfStart::TakeAlternative;   // the last four probes in a row chose it: the next 32 searches do not ask
    if (route_hint > 0) return PfStart::Probe;             // recent scans were abandoned
    return PfStart::Scan;
}

}  // namespace acgpu
