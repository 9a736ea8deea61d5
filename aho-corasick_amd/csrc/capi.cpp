// C ABI of libacgpu.so (include/acgpu.h): host orchestration of the device pipeline.
//
// No CPU search path exists here by design: every search entry point launches HIP
// kernels and fails with ACGPU_ERR_NO_DEVICE / ACGPU_ERR_HIP when that is impossible.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "acgpu.h"
#include "capi_internal.hpp"
#include "device/cnfa_walk.hpp"
#include "device/cnfa_tri.hpp"
#include "device/cnfa_tri_step.hpp"
#include "device/dfa_tri.hpp"
#include "device/dfa_fill.hpp"
#include "device/hot.hpp"
#include "device/kernels.hpp"
#include "device/select.hpp"
#include "host/automaton.hpp"
#include "host/cnfa_tables.hpp"
#include "host/cnfa_tri_tables.hpp"
#include "host/devbuf.hpp"
#include "host/lw_tables.hpp"
#include "host/pf_tables.hpp"

using namespace acgpu;

namespace acgpu_capi {

thread_local std::string g_last_error;
// Set by overlapping_impl in its internal (dev_result) mode when the occurrence stream is too dense to be worth
// materialising: the callers (find_iter / find / replace_all) then run the reference loop on one lane instead, which
// costs ~30 ns per haystack byte whatever the number of occurrences.
thread_local bool g_too_dense = false;
thread_local bool g_dense_guard = true;   // off inside the stream search, which has no serial alternative
inline bool too_dense(uint64_t records, uint64_t span_bytes) {
    return g_dense_guard && records > std::max<uint64_t>(uint64_t(1) << 24, 32 * span_bytes);
}

acgpu_status hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();   // reset the thread's last-error slot: a later launch check must not see this failure
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) return ACGPU_ERR_NO_DEVICE;
    return e == hipErrorOutOfMemory ? ACGPU_ERR_NOMEM : ACGPU_ERR_HIP;
}
#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);     \
    } while (0)

// One scan's worth of scratch; pooled per device so concurrent searches do not share state.
struct Scratch {
    DevBuf counts, offsets, active, aoff, bsum, bact, totals, result, hay, sel, selwork, seltot;
    DevBuf rhay, rmatch, rtab, roff, rwork, rout;  // replace_all
    DevBuf events, evrank, evctr, eswork;          // prefix-filter direct / sorted-events modes (level-3 events -> ordered records)
    DevBuf hitwork;                                // large-set filter: global hit list of its second-pass level 3
    DevBuf triev, triseg, trictr;                  // contiguous-NFA walk: match events of the count pass (cnfa_tri.hip)
    DevBuf probe;                                  // prefix-filter probe: 8 counters + the decision word at byte 64 (zeroed once)
    bool probe_ready = false;
    size_t eswork_inited = 0;                      // size of eswork when its barrier words were last zeroed (event_order.hip)
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_armed = false;        // evrank[] == 0 and evctr[] == 0 (the invariant k_ev_write restores; false after a failed call)
    uint32_t rank_hint = 0;       // events of the previous event-mode call on this scratch: sizes the all-pairs grid only
    uint64_t* pinned = nullptr;   // [4] page-locked landing zone for the totals (a pageable target makes the copy a staged, blocking one)
    hipError_t ensure_pinned() { return pinned ? hipSuccess : hipHostMalloc(reinterpret_cast<void**>(&pinned), 4 * sizeof(uint64_t)); }
    ~Scratch() {
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (pinned) (void)hipHostFree(pinned);
    }
};

struct DeviceState {
    int device = -1;
    DevAutomaton da;
    DevBuf dfa_trans, dfa_moff, dfa_mpid, dfa_cls, cnfa_repr, cnfa_cls, plens;
    HotTables hot;   // LDS-resident fast path (hot_scan.hip), optional
    CnfaHotTables cnfa_hot;   // contiguous-NFA walk with the start state's neighbourhood in LDS (cnfa_walk.hip)
    CnfaTriTables cnfa_tri;   // contiguous-NFA walk that skips the depth <= 2 regime by a trigram bitmap in LDS (cnfa_tri.hip)
    DfaTriTables dfa_tri;     // the same skip in front of the DFA transition walk (dfa_tri.hip)
    bool derived_dfa = false;  // da.dfa was derived from an NFA-kind automaton at upload (device only)
    // > 0 while recent scans of this automaton were abandoned by the two-type filter (PfArgs::route_*): the next scans
    // ask the probe (launch_pf_probe, ~10 us) which engine to run instead of paying for an abandoned pass each; every
    // probe that finds the filter adequate counts it down, so a caller with harmless input stops paying for probes
    std::atomic<int> route_hint{0};
    std::mutex pool_mu;
    std::vector<std::unique_ptr<Scratch>> pool;
    // enqueue-only calls: one scratch per stream, never pooled (work of earlier calls may still be in flight on it;
    // stream order makes the reuse by the next call on the same stream safe)
    struct AsyncCtx {
        Scratch sc;
        hipEvent_t ev[128] = {};
        ~AsyncCtx() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); }
    };
    std::mutex async_mu;
    std::map<hipStream_t, std::unique_ptr<AsyncCtx>> async;
    AsyncCtx* async_ctx(hipStream_t s) {
        std::lock_guard<std::mutex> lk(async_mu);
        auto& p = async[s];
        if (!p) p = std::make_unique<AsyncCtx>();
        return p.get();
    }

    std::unique_ptr<Scratch> take() {
        std::lock_guard<std::mutex> lk(pool_mu);
        if (!pool.empty()) { auto s = std::move(pool.back()); pool.pop_back(); return s; }
        return std::make_unique<Scratch>();
    }
    void give(std::unique_ptr<Scratch> s) {
        std::lock_guard<std::mutex> lk(pool_mu);
        if (pool.size() < 4) pool.push_back(std::move(s));
    }
};

}  // namespace acgpu_capi
using namespace acgpu_capi;

acgpu_automaton::acgpu_automaton() = default;
acgpu_automaton::~acgpu_automaton() = default;   // (here DeviceState is complete)

// Stream search state (src/automaton.rs:1036-1244): what StreamChunkIter carries between reads.
struct acgpu_stream {
    acgpu_automaton* aut = nullptr;
    DevBuf buf;                     // [halo | chunk] on the device
    DevBuf stage;                   // device copy of a large host feed, filled piece by piece under the search (HostPipe)
    std::vector<uint8_t> halo;      // last max_pattern_len-1 bytes of the stream so far
    std::vector<acgpu_match> last;  // matches completed by the most recent feed (absolute offsets)
    uint64_t total = 0;             // bytes consumed so far
    uint64_t pos = 0;               // end of the last reported match
};

namespace {

struct ScratchLease {
    DeviceState* ds;
    std::unique_ptr<Scratch> s;
    ScratchLease(DeviceState* d) : ds(d), s(d->take()) {}
    ~ScratchLease() { ds->give(std::move(s)); }
    Scratch* operator->() { return s.get(); }
};

acgpu_status get_device_state(acgpu_automaton* aut, DeviceState** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    {
        std::lock_guard<std::mutex> lk(aut->mu);
        auto it = aut->devs.find(dev);
        if (it != aut->devs.end()) { *out = it->second.get(); return ACGPU_OK; }
    }
    acgpu_status st = acgpu_upload(aut, dev);
    if (st != ACGPU_OK) return st;
    std::lock_guard<std::mutex> lk(aut->mu);
    *out = aut->devs[dev].get();
    return ACGPU_OK;
}

// ahocorasick.rs:2778-2789
acgpu_status enforce_anchored_consistency(int have, bool want_anchored) {
    switch (have) {
        case ACGPU_START_BOTH: return ACGPU_OK;
        case ACGPU_START_UNANCHORED: return want_anchored ? ACGPU_ERR_INVALID_INPUT_ANCHORED : ACGPU_OK;
        default: return want_anchored ? ACGPU_OK : ACGPU_ERR_INVALID_INPUT_UNANCHORED;
    }
}

// search.rs:332-342
acgpu_status check_input(const acgpu_input* in) {
    if (!in) return ACGPU_ERR_INVALID_ARGUMENT;
    if (!(in->span_end <= in->haystack_len && in->span_start <= in->span_end + 1)) return ACGPU_ERR_INVALID_SPAN;
    if (in->haystack_len > 0 && !in->haystack) return ACGPU_ERR_INVALID_ARGUMENT;
    return ACGPU_OK;
}

// The reference-faithful walk engine of this automaton on device `ds`: the DFA when the device holds one (the
// automaton's own, or the one derived from an NFA-kind automaton at upload), else the contiguous-NFA walk.
uint32_t generic_engine(const acgpu_automaton* aut, const DeviceState* ds) {
    if (ds ? ds->da.has_dfa : aut->kind == ACGPU_KIND_DFA) return ENG_DFA;
    return ENG_CNFA;
}

// Start state availability: DFA::start_state, src/dfa.rs:190-215
acgpu_status check_start(const acgpu_automaton* aut, bool anchored) {
    if (aut->kind != ACGPU_KIND_DFA) return ACGPU_OK;
    uint32_t s = anchored ? aut->dfa.special.start_anchored_id : aut->dfa.special.start_unanchored_id;
    if (s == kDead) return anchored ? ACGPU_ERR_INVALID_INPUT_ANCHORED : ACGPU_ERR_INVALID_INPUT_UNANCHORED;
    return ACGPU_OK;
}

// Makes [lo, hi) of the haystack addressable on the device; returns a pointer p such that p[i] is
// haystack byte i for i in [lo, hi).
acgpu_status device_haystack(const acgpu_input* in, size_t lo, size_t hi, Scratch* sc, hipStream_t stream,
                             const uint8_t** out) {
    if (in->haystack_on_device) { *out = in->haystack; return ACGPU_OK; }
    const size_t n = hi > lo ? hi - lo : 0;
    HIP_TRY(sc->hay.ensure(n + 32));
    if (n) HIP_TRY(hipMemcpyAsync(sc->hay.p, in->haystack + lo, n, hipMemcpyHostToDevice, stream));
    *out = sc->hay.as<uint8_t>() - lo;
    return ACGPU_OK;
}

acgpu_status ensure_events(Scratch* sc) {
    for (auto& e : sc->ev) if (!e) HIP_TRY(hipEventCreate(&e));
    return ACGPU_OK;
}

uint32_t default_chunk(const acgpu_automaton* aut, size_t span_len) {
    uint32_t c = aut->cfg.chunk_bytes ? aut->cfg.chunk_bytes : 2048u;
    c = (c + 63u) & ~63u;
    if (c < 64) c = 64;
    (void)span_len;
    return c;
}

// Scan geometry of one shard: 16-byte aligned base, ownership window, lane-chunk grid.
#ifdef ACGPU_GUARD
// Bounds-checked debug build: one violation counter per device, allocated on first use and never freed.
std::mutex g_guard_mu;
std::map<int, unsigned long long*> g_guard_ctrs;
unsigned long long* guard_counter() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_guard_mu);
    unsigned long long*& p = g_guard_ctrs[dev];
    if (!p) {
        if (hipMalloc(reinterpret_cast<void**>(&p), sizeof(unsigned long long)) != hipSuccess) { p = nullptr; return nullptr; }
        (void)hipMemset(p, 0, sizeof(unsigned long long));
    }
    return p;
}
#endif

ScanGeom make_geom(const acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end,
                   const uint8_t* dhay, size_t halo) {
    ScanGeom g{};
    // 64-byte alignment of the virtual origin: chunk boundaries (multiples of 64 in virtual coordinates) then fall on
    // 64-byte memory segments, so the per-lane streams of the LDS walk engine request every segment exactly once
    const uint64_t mis = uint64_t(reinterpret_cast<uintptr_t>(dhay) & 63);
    g.hay16 = dhay - mis;
    g.base_mis = mis;
    g.cold_floor = in->span_start + mis;
    g.emit_lo = shard_begin + mis;
    g.emit_hi = shard_end + mis;
    g.chunk = default_chunk(aut, shard_end - shard_begin);
    g.halo = uint32_t(halo);
    g.grid0 = (g.emit_lo / g.chunk) * g.chunk;
    g.n_chunks = std::max<uint64_t>(1, (g.emit_hi - g.grid0 + g.chunk - 1) / g.chunk);
    g.emit_start_matches = shard_begin == in->span_start ? 1u : 0u;
#ifdef ACGPU_GUARD
    g.guard = guard_counter();
    g.guard_lo = g.cold_floor & ~uint64_t(15);
    g.guard_hi = (in->span_end + mis + 15) & ~uint64_t(15);
    if (std::getenv("ACGPU_GUARD_SHRINK")) {   // positive control of the test: a hull 16 bytes too small on both sides must be noticed
        g.guard_lo += 16;
        if (g.guard_hi >= g.guard_lo + 16) g.guard_hi -= 16;
    }
#endif
    return g;
}

// ---- overlapping search of one shard --------------------------------------------------------------------------------
// Everything one call needs, resolved once by overlapping_impl and shared by the pipelines below.
struct OvCtx {
    acgpu_automaton* aut = nullptr;
    DeviceState* ds = nullptr;
    Scratch* sc = nullptr;
    const acgpu_input* in = nullptr;
    hipStream_t stream = nullptr;
    size_t shard_begin = 0, shard_end = 0;
    uint64_t span_bytes = 0;
    ScanGeom g{};
    ScanScratch ss;
    acgpu_match* out = nullptr;       // caller's buffer (host, or device when to_caller)
    size_t cap = 0;
    size_t* n_out = nullptr;
    acgpu_profile* prof = nullptr;
    acgpu_match** dev_result = nullptr;   // internal mode (parallel find_iter): leave the records in sc->result
    bool to_caller = false;               // records go straight into the caller's device buffer
    uint32_t routed = 0;                  // the prefix filter abandoned the scan; another engine repeated it
    bool force_large_set = false;         // ... namely the large-set filter (whatever the pattern count)
};

constexpr uint32_t ENG_PF_LARGE = 100;   // pf_alternative only: the prefix filter's other kernel (reported as ENG_PF)

// Scratch of the large-set filter's second pass, when launch_pf_any is going to run that filter.
acgpu_status pf_route_prepare(Scratch* sc, const HotTables& h, uint64_t span_bytes, PfRoute* r) {
    if (!pf_uses_large_set(h, *r)) return ACGPU_OK;
    const size_t need = pfx_hit_work_bytes(span_bytes);
    HIP_TRY(sc->hitwork.ensure(need));
    r->hit_work = sc->hitwork.p;
    r->hit_work_bytes = need;
    return ACGPU_OK;
}

// Shared epilogue: what every pipeline reports once the record count is known.
void ov_profile(const OvCtx& c, uint32_t eng, uint64_t n_records, uint64_t n_active) {
    if (!c.prof) return;
    c.prof->bytes_scanned = c.span_bytes;
    c.prof->n_chunks = c.g.n_chunks;
    c.prof->n_active_chunks = n_active;
    c.prof->n_matches = n_records;
    c.prof->engine_used = eng;
    c.prof->routed = c.routed;
}
acgpu_status ov_events_ms(const OvCtx& c, bool have_rank) {   // ev[0] count start, [1] count end, [2] rank end, [3]/[4] around the emit
    if (!c.prof) return ACGPU_OK;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c.sc->ev[0], c.sc->ev[1])); c.prof->ms_scan = ms;
    if (have_rank) { HIP_TRY(hipEventElapsedTime(&ms, c.sc->ev[1], c.sc->ev[2])); c.prof->ms_compact = ms; }
    else c.prof->ms_compact = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c.sc->ev[3], c.sc->ev[4])); c.prof->ms_fill = ms;
    HIP_TRY(hipEventElapsedTime(&ms, c.sc->ev[0], c.sc->ev[4])); c.prof->ms_total = ms;
    return ACGPU_OK;
}
acgpu_status ov_result(const OvCtx& c, uint64_t n_records, acgpu_match* dev_records) {
    if (c.dev_result) { *c.dev_result = dev_records; return ACGPU_OK; }
    if (n_records > c.cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (n_records && !c.out) return ACGPU_ERR_INVALID_ARGUMENT;
    return ACGPU_OK;
}

constexpr uint32_t kEvAllPairs = 16384;                 // events the all-pairs rank orders (k_ev_rank)
constexpr uint64_t kSortMaxEvents = uint64_t(12) << 20;  // events the bucket order pass takes (event_order.hip)

enum class PfOutcome { Done, Abandoned, TooManyEvents };

// Prefix filter, event modes.  ONE scan records every occurrence as an event {end, length, trie node} (level 3 knows
// them exactly); the ordered records then come from the events without another look at the haystack: up to kEvAllPairs
// events by the all-pairs rank + scatter (k_ev_rank / k_ev_write, enqueued right behind the scan, no host decision
// needed), beyond that by the device radix sort of the same buffer (event_sort.hip).  The event buffer is sized from the
// span (one event per 64 haystack bytes, at most kSortMaxEvents), so which of the two runs is decided by the count this
// very call produced -- no state carried between calls.  Outcomes other than Done leave no result: the scan was
// abandoned by its routing rule (PfArgs::route_*), or produced more events than the buffer holds.
// scratch of the bucket order pass (event_order.hip), its barrier words zeroed whenever it was (re)allocated
acgpu_status ensure_order_work(Scratch* sc, size_t bytes, hipStream_t stream) {
    HIP_TRY(sc->eswork.ensure(bytes));
    if (sc->eswork_inited != sc->eswork.bytes) {
        HIP_TRY(event_order_init(sc->eswork.p, stream));
        sc->eswork_inited = sc->eswork.bytes;
    }
    return ACGPU_OK;
}
constexpr uint64_t kProbeMinSpan = uint64_t(16) << 20;   // shards below this pay less for an abandoned pass than a probe is worth
acgpu_status ensure_probe(Scratch* sc, hipStream_t stream) {
    if (sc->probe_ready) return ACGPU_OK;
    HIP_TRY(sc->probe.ensure(128));
    HIP_TRY(hipMemsetAsync(sc->probe.p, 0, 128, stream));
    sc->probe_ready = true;
    return ACGPU_OK;
}

acgpu_status pf_events(OvCtx& c, PfRoute route, PfOutcome* outcome, acgpu_status* result) {
    Scratch* sc = c.sc;
    hipStream_t stream = c.stream;
    *outcome = PfOutcome::Done;
    *result = ACGPU_OK;
    const uint64_t cap_ev = std::min<uint64_t>(kSortMaxEvents, std::max<uint64_t>(uint64_t(1) << 16, c.span_bytes / 64));
    // invariant between calls: rank[] == 0 and the counters == 0 (k_ev_write restores it).  A call that fails between
    // the scan and k_ev_write leaves them dirty; ev_armed says whether the invariant holds.
    HIP_TRY(sc->events.ensure(size_t(cap_ev) * pf_event_bytes()));
    HIP_TRY(sc->evrank.ensure(size_t(kEvAllPairs) * sizeof(uint32_t)));
    HIP_TRY(sc->evctr.ensure(kPfCtrWords * sizeof(unsigned long long)));
    if (!sc->ev_armed) {
        HIP_TRY(hipMemsetAsync(sc->evrank.p, 0, size_t(kEvAllPairs) * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(sc->evctr.p, 0, kPfCtrWords * sizeof(unsigned long long), stream));
    }
    sc->ev_armed = false;
    unsigned long long* ctr = sc->evctr.as<unsigned long long>();
    uint32_t* rank = sc->evrank.as<uint32_t>();
    if (acgpu_status st = pf_route_prepare(sc, c.ds->hot, c.span_bytes, &route)) return st;
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[0], stream));
    HIP_TRY(launch_pf_any(c.ds->hot, c.g, nullptr, stream, sc->events.p, ctr, cap_ev, route));
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[1], stream));
    HIP_TRY(launch_pf_event_rank(sc->events.p, ctr, kEvAllPairs, rank, c.ss.totals, sc->rank_hint, stream));
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[2], stream));
    if (c.to_caller) {   // device-resident output: the scatter is enqueued without a host round trip
        if (c.prof) HIP_TRY(hipEventRecord(sc->ev[3], stream));
        HIP_TRY(launch_pf_event_write(c.ds->hot, c.ds->da, sc->events.p, ctr, kEvAllPairs, rank, c.ss.totals, c.out ? c.cap : 0, c.out, stream));
        if (c.prof) HIP_TRY(hipEventRecord(sc->ev[4], stream));
        sc->ev_armed = true;
    }
    HIP_TRY(sc->ensure_pinned());
    HIP_TRY(hipMemcpyAsync(sc->pinned, c.ss.totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const uint64_t n_records = sc->pinned[0], n_events = sc->pinned[1];
    const bool abandoned = n_events == ~uint64_t(0);
    const bool all_pairs = n_events <= kEvAllPairs;
    acgpu_match* dout = nullptr;
    if (!c.to_caller) {   // host / scratch output: size the buffer first, then scatter (always launched: it re-arms)
        const bool emit = all_pairs && n_records > 0 && (c.dev_result || (n_records <= c.cap && c.out));
        if (emit) { HIP_TRY(sc->result.ensure(n_records * sizeof(acgpu_match))); dout = sc->result.as<acgpu_match>(); }
        if (c.prof) HIP_TRY(hipEventRecord(sc->ev[3], stream));
        HIP_TRY(launch_pf_event_write(c.ds->hot, c.ds->da, sc->events.p, ctr, kEvAllPairs, rank, c.ss.totals, emit ? n_records : 0, dout, stream));
        if (c.prof) HIP_TRY(hipEventRecord(sc->ev[4], stream));
        sc->ev_armed = true;
        if (emit && !c.dev_result)
            HIP_TRY(hipMemcpyAsync(c.out, dout, n_records * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    if (abandoned) { *outcome = PfOutcome::Abandoned; return ACGPU_OK; }
    if (n_events > cap_ev) { *outcome = PfOutcome::TooManyEvents; return ACGPU_OK; }
    sc->rank_hint = uint32_t(std::min<uint64_t>(n_events, kEvAllPairs));
    *c.n_out = size_t(n_records);
    if (all_pairs) {
        ov_profile(c, ENG_PF, n_records, n_events);
        acgpu_status st = ov_events_ms(c, true);
        if (st) return st;
        *result = ov_result(c, n_records, dout);
        return ACGPU_OK;
    }
    // bucket order pass over the events this scan recorded (event_order.hip; the counters were re-armed by k_ev_write, the
    // counts are still in the device totals)
    acgpu_match* dst = nullptr;
    if (c.to_caller) { if (c.out && n_records <= c.cap) dst = c.out; }
    else if (n_records > 0 && (c.dev_result || (n_records <= c.cap && c.out))) {
        if (c.dev_result && too_dense(n_records, c.span_bytes)) { g_too_dense = true; *result = ACGPU_ERR_NOMEM; return ACGPU_OK; }
        HIP_TRY(sc->result.ensure(n_records * sizeof(acgpu_match)));
        dst = sc->result.as<acgpu_match>();
    }
    // (the selection kernels of the parallel find_iter read the record count from the device totals: still there)
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[3], stream));
    if (dst && n_events) {
        if (acgpu_status st = ensure_order_work(sc, event_order_work_bytes(n_events, n_records, c.span_bytes), stream)) return st;
        HIP_TRY(launch_event_order_emit(c.ds->hot, c.ds->da, sc->events.p, c.ss.totals, kEvAllPairs, n_events, n_records,
                                        c.shard_begin, c.span_bytes, sc->eswork.p, dst, stream));
    }
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[4], stream));
    if (dst && !c.to_caller && !c.dev_result)
        HIP_TRY(hipMemcpyAsync(c.out, dst, n_records * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    ov_profile(c, ENG_PF, n_records, n_events);
    acgpu_status st = ov_events_ms(c, false);
    if (st) return st;
    *result = ov_result(c, n_records, dst);
    return ACGPU_OK;
}

// the transition-walk count kernel of `eng` (global tables; the contiguous NFA through its LDS-assisted form when available)
// does the contiguous-NFA walk of `ds` run the shallow-skip kernel (cnfa_tri.hip)?
bool cnfa_tri_selected(const DeviceState* ds) {
    static const bool literal = std::getenv("ACGPU_CNFA_LITERAL") != nullptr;   // A/B knob: the reference loop verbatim
    static const bool no_tri = std::getenv("ACGPU_CNFA_NO_TRI") != nullptr;     // A/B knob: the LDS-row walk (cnfa_walk.hip)
    return ds->cnfa_tri.ready && !literal && !no_tri;
}
// ... and does the DFA walk run its shallow-skip kernel (dfa_tri.hip)?
bool dfa_tri_selected(const DeviceState* ds) {
    static const bool no_tri = std::getenv("ACGPU_DFA_NO_TRI") != nullptr;      // A/B knob: the global-table walk of kernels.hip
    return ds->dfa_tri.ready && !no_tri;
}
bool tri_walk_selected(uint32_t eng, const DeviceState* ds) {
    return (eng == ENG_CNFA && cnfa_tri_selected(ds)) || (eng == ENG_DFA && dfa_tri_selected(ds));
}
// Event buffer for a scan by that kernel (zeroed counters enqueued on `stream`): the count pass then records every
// match state it enters, and k_cnfa_tri_emit writes the ordered records without walking the haystack again.
acgpu_status cnfa_tri_events(Scratch* sc, const ScanGeom& g, uint64_t span_bytes, hipStream_t stream, TriEvents* ev) {
    *ev = TriEvents();
    static const bool off = std::getenv("ACGPU_CNFA_NO_EVENTS") != nullptr;   // A/B knob: count -> scan -> re-walking fill
    if (off || g.n_chunks >= 0xFFFFFFFFull) return ACGPU_OK;
    const uint32_t segs = tri_event_segments(span_bytes);
    HIP_TRY(sc->triev.ensure(size_t(segs) * kTriSeg * sizeof(TriEvent)));
    HIP_TRY(sc->triseg.ensure(size_t(segs) * sizeof(uint32_t)));
    HIP_TRY(sc->trictr.ensure(2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(sc->trictr.p, 0, 2 * sizeof(unsigned long long), stream));
    ev->ev = sc->triev.as<TriEvent>(); ev->seg_fill = sc->triseg.as<uint32_t>();
    ev->ctr = sc->trictr.as<unsigned long long>(); ev->max_segs = segs;
    return ACGPU_OK;
}

hipError_t launch_generic_count(uint32_t eng, DeviceState* ds, const ScanGeom& g, uint32_t* counts, hipStream_t stream,
                                const TriEvents* tev = nullptr) {
    static const bool literal = std::getenv("ACGPU_CNFA_LITERAL") != nullptr;   // A/B knob: the reference loop verbatim
    if (eng == ENG_CNFA && cnfa_tri_selected(ds)) return launch_cnfa_tri_count(ds->cnfa_tri, g, counts, tev && tev->ev ? tev : nullptr, stream);
    if (eng == ENG_DFA && dfa_tri_selected(ds)) return launch_dfa_tri_count(ds->dfa_tri, g, counts, tev && tev->ev ? tev : nullptr, stream);
    if (eng == ENG_CNFA && ds->cnfa_hot.ready && !literal) return launch_cnfa_count(ds->cnfa_hot, ds->da, g, counts, stream);
    return launch_walk_count(eng, ds->da, g, counts, stream);
}

// Classic pipeline, any count engine: per-chunk counts -> scan + compaction -> fill of the non-empty chunks by the
// reference-faithful walk (from LDS-resident rows when the automaton has them: same states, same match lists).
acgpu_status classic_pipeline(OvCtx& c, uint32_t eng) {
    Scratch* sc = c.sc;
    hipStream_t stream = c.stream;
    acgpu_automaton* aut = c.aut;
    DeviceState* ds = c.ds;
    const ScanGeom& g = c.g;
    PfRoute pfr;
    pfr.force_pfx = c.force_large_set;
    if (eng == ENG_PF) if (acgpu_status st = pf_route_prepare(sc, ds->hot, c.span_bytes, &pfr)) return st;
    TriEvents tev;   // shallow-skip walks: records from the count pass's events (no second walk)
    if (tri_walk_selected(eng, ds)) {
        if (acgpu_status st = cnfa_tri_events(sc, g, c.span_bytes, stream, &tev)) return st;
        if (tev.ev) {   // the emit kernel looks up every chunk's output offset
            HIP_TRY(sc->offsets.ensure(g.n_chunks * sizeof(uint64_t)));
            c.ss.offsets = sc->offsets.as<uint64_t>();
        }
    }
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[0], stream));
    if (eng == ENG_PF) HIP_TRY(launch_pf_any(ds->hot, g, c.ss.counts, stream, nullptr, nullptr, 0, pfr));
    else if (eng == ENG_HOT) HIP_TRY(launch_hot_count(ds->hot, ds->da, g, c.ss.counts, stream));
    else HIP_TRY(launch_generic_count(eng, ds, g, c.ss.counts, stream, &tev));
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[1], stream));
    HIP_TRY(launch_scan(c.ss, g.n_chunks, stream));
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[2], stream));
    const uint32_t fill_eng = generic_engine(aut, ds);
    const bool hot_fill = fill_eng == ENG_DFA && aut->cfg.engine != 1 && hot_fill_supported(ds->hot, g);
    bool events_ok = tev.ev != nullptr;   // (host path: cleared below when the buffer overflowed)
    auto fill = [&](uint64_t fcap, uint64_t max_waves, acgpu_match* dst) -> hipError_t {
        if (tev.ev) {
            // event form: the emit kernel; the re-walking fill behind it only runs if the events overflowed (gate)
            if (events_ok) {
                const hipError_t e = eng == ENG_CNFA
                    ? launch_cnfa_tri_emit(ds->cnfa_tri, ds->da.cnfa.plens, g, tev, c.ss.offsets, c.ss.totals, fcap, dst, stream)
                    : launch_dfa_tri_emit(ds->dfa_tri, ds->da, g, tev, c.ss.offsets, c.ss.totals, fcap, dst, stream);
                if (e != hipSuccess) return e;
            }
            return launch_walk_fill(eng, ds->da, g, c.ss.active, c.ss.totals, fcap, max_waves, c.ss.aoff, dst, stream, tev.ctr + 1);
        }
        if (hot_fill) return launch_hot_fill(ds->hot, ds->da, g, c.ss.active, c.ss.totals, fcap, max_waves, c.ss.aoff, dst, stream);
        return launch_walk_fill(fill_eng, ds->da, g, c.ss.active, c.ss.totals, fcap, max_waves, c.ss.aoff, dst, stream);
    };
    if (c.to_caller) {
        // Device-resident output: the fill kernel reads the totals on the device, so it is enqueued right behind
        // the scan without a host round trip; it writes nothing if the records would not fit into `cap`.
        if (c.prof) HIP_TRY(hipEventRecord(sc->ev[3], stream));
        if (c.cap > 0 && c.out) HIP_TRY(fill(c.cap, 16384, c.out));
        if (c.prof) HIP_TRY(hipEventRecord(sc->ev[4], stream));
    }
    HIP_TRY(sc->ensure_pinned());
    HIP_TRY(hipMemcpyAsync(sc->pinned, c.ss.totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    if (tev.ev) HIP_TRY(hipMemcpyAsync(sc->pinned + 2, tev.ctr, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const uint64_t n_records = sc->pinned[0], n_active = sc->pinned[1];
    if (tev.ev && sc->pinned[3] != 0) events_ok = false;   // more events than the buffer holds: the re-walking fill alone
    *c.n_out = size_t(n_records);
    ov_profile(c, eng, n_records, n_active);
    if (c.prof) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev[0], sc->ev[1])); c.prof->ms_scan = ms;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev[1], sc->ev[2])); c.prof->ms_compact = ms;
        c.prof->ms_total = c.prof->ms_scan + c.prof->ms_compact;
        if (c.to_caller) {
            HIP_TRY(hipEventElapsedTime(&ms, sc->ev[3], sc->ev[4])); c.prof->ms_fill = ms;
            HIP_TRY(hipEventElapsedTime(&ms, sc->ev[0], sc->ev[4])); c.prof->ms_total = ms;
        }
    }
    if (c.dev_result) *c.dev_result = nullptr;
    if (!c.dev_result && n_records > c.cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (n_records == 0 || c.to_caller) return ACGPU_OK;
    if (!c.out && !c.dev_result) return ACGPU_ERR_INVALID_ARGUMENT;
    if (c.dev_result && too_dense(n_records, c.span_bytes)) { g_too_dense = true; return ACGPU_ERR_NOMEM; }
    HIP_TRY(sc->result.ensure(n_records * sizeof(acgpu_match)));
    acgpu_match* dout = sc->result.as<acgpu_match>();
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[3], stream));
    HIP_TRY(fill(n_records, n_active, dout));
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[4], stream));
    if (c.dev_result) *c.dev_result = dout;  // records stay in scratch->result; the caller continues on the same stream
    else HIP_TRY(hipMemcpyAsync(c.out, dout, n_records * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (c.prof) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev[3], sc->ev[4])); c.prof->ms_fill = ms;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev[0], sc->ev[4])); c.prof->ms_total = ms;
    }
    return ACGPU_OK;
}

// The engine a search may be handed to when the prefix filter abandons it (PfArgs::route_*), and the cost-model
// coefficients that go with it.  Only the automatic engine choice routes; an explicitly requested engine is kept.
uint32_t pf_alternative(const acgpu_automaton* aut, const DeviceState* ds, PfRoute* route) {
    *route = PfRoute();
    if (aut->cfg.engine != 0) return 0;
    static const bool off = std::getenv("ACGPU_NO_ROUTING") != nullptr;   // A/B knob
    if (off) return 0;
    if (ds->hot.lw_ready && aut->nnfa.min_pattern_len > 0) { *route = kPfRouteToLdsWalk(); return ENG_HOT; }
    // (automata too large for LDS) the large-set filter: its level 3 is a second, throughput-oriented pass, so inputs
    // that drown the two-type filter's inline level 3 -- natural text against a dictionary -- cost it far less
    static const bool no_ls = std::getenv("ACGPU_NO_ROUTE_LARGE_SET") != nullptr;   // A/B knob
    if (ds->hot.pfx_ready && !no_ls) {
        *route = kPfRouteToLargeSet();
        if (const char* cb = std::getenv("ACGPU_ROUTE_LS_CB")) route->cb = uint32_t(std::atoi(cb));   // tuning knob
        return ENG_PF_LARGE;
    }
    if (ds->da.has_dfa) { *route = kPfRouteToDfaWalk(); return ENG_DFA; }
    return 0;
}

// `ext` / `dev_result`: internal mode used by the parallel find_iter -- run on the caller's scratch and leave the
// ordered records in scratch->result (returned through *dev_result) instead of copying them anywhere.
acgpu_status overlapping_impl(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end,
                              acgpu_match* out, size_t cap, size_t* n_out, acgpu_profile* prof,
                              Scratch* ext = nullptr, acgpu_match** dev_result = nullptr) {
    if (!aut || !n_out) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = 0;
    if (prof) std::memset(prof, 0, sizeof *prof);
    acgpu_status st = check_input(in);
    if (st) return st;
    const bool anchored = in->anchored != 0;
    if ((st = enforce_anchored_consistency(aut->cfg.start_kind, anchored))) return st;
    // automaton.rs:397-423
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD) return ACGPU_ERR_UNSUPPORTED_OVERLAPPING;
    if (anchored) return ACGPU_ERR_INVALID_INPUT_ANCHORED;
    if ((st = check_start(aut, false))) return st;
    if (in->span_start > in->span_end) return ACGPU_OK;  // Input::is_done
    if (!(in->span_start <= shard_begin && shard_begin <= shard_end && shard_end <= in->span_end))
        return ACGPU_ERR_INVALID_ARGUMENT;

    // StartKind::Both: the unanchored side is served by the twin automaton built with an unanchored start (the same
    // noncontiguous NFA, hence the same match lists in the same order), to which the LDS engines apply; the
    // interleaved two-start DFA layout (dfa.rs:617-724) itself only has the reference-faithful walk
    if (aut->cfg.start_kind == ACGPU_START_BOTH && aut->occ)
        return overlapping_impl(aut->occ.get(), in, shard_begin, shard_end, out, cap, n_out, prof, ext, dev_result);

    DeviceState* ds = nullptr;
    if ((st = get_device_state(aut, &ds))) return st;
    std::unique_ptr<ScratchLease> lease;
    if (!ext) lease = std::make_unique<ScratchLease>(ds);
    OvCtx c;
    c.aut = aut; c.ds = ds; c.sc = ext ? ext : lease->s.get(); c.in = in;
    c.stream = static_cast<hipStream_t>(in->stream);
    c.shard_begin = shard_begin; c.shard_end = shard_end; c.span_bytes = shard_end - shard_begin;
    c.out = out; c.cap = cap; c.n_out = n_out; c.prof = prof; c.dev_result = dev_result;
    c.to_caller = in->out_on_device && !dev_result;
    Scratch* sc = c.sc;
    if (prof && (st = ensure_events(sc))) return st;

    const size_t halo = aut->nnfa.max_pattern_len > 0 ? aut->nnfa.max_pattern_len - 1 : 0;
    if (halo > 0xFFFFFF00ull) return ACGPU_ERR_INVALID_ARGUMENT;
    const size_t need_lo = std::max(in->span_start, shard_begin >= halo ? shard_begin - halo : size_t(0));
    const uint8_t* dhay = nullptr;
    if ((st = device_haystack(in, need_lo, shard_end, sc, c.stream, &dhay))) return st;
    c.g = make_geom(aut, in, shard_begin, shard_end, dhay, halo);

    const uint64_t nb = (c.g.n_chunks + 255) / 256;
    HIP_TRY(sc->counts.ensure(c.g.n_chunks * sizeof(uint32_t)));
    HIP_TRY(sc->active.ensure(c.g.n_chunks * sizeof(uint64_t)));
    HIP_TRY(sc->aoff.ensure(c.g.n_chunks * sizeof(uint64_t)));
    HIP_TRY(sc->bsum.ensure(nb * sizeof(uint64_t)));
    HIP_TRY(sc->bact.ensure(nb * sizeof(uint32_t)));
    HIP_TRY(sc->totals.ensure(2 * sizeof(uint64_t)));
    c.ss.counts = sc->counts.as<uint32_t>(); c.ss.offsets = nullptr;   // the fill only needs the active chunks' offsets
    c.ss.active = sc->active.as<uint64_t>(); c.ss.aoff = sc->aoff.as<uint64_t>(); c.ss.bsum = sc->bsum.as<uint64_t>();
    c.ss.bact = sc->bact.as<uint32_t>(); c.ss.totals = sc->totals.as<uint64_t>();

    // engine choice (cfg.engine: 0 auto, 1 walk, 2 LDS walk, 3 prefix filter); auto prefers the fastest engine that is
    // available for this automaton.  All engines produce identical results.
    uint32_t eng = generic_engine(aut, ds);
    const int want = aut->cfg.engine;
    if (eng == ENG_DFA) {
        if ((want == 0 || want == 3) && ds->hot.pf_ready) eng = ENG_PF;
        // (with an empty pattern every state is a match state: the LDS walk would run its exact path throughout)
        else if (((want == 0 && aut->nnfa.min_pattern_len > 0) || want == 2) && ds->hot.lw_ready) eng = ENG_HOT;
    }
    if ((want == 2 && eng != ENG_HOT) || (want == 3 && eng != ENG_PF)) {
        g_last_error = "requested engine is unavailable for this automaton";
        return ACGPU_ERR_INVALID_ARGUMENT;
    }

    static const bool no_events = std::getenv("ACGPU_PF_CLASSIC") != nullptr;   // A/B knob: chunk counters + scan + fill
    if (eng == ENG_PF && aut->nnfa.max_pattern_len <= 0xFFFF && !no_events) {
        PfRoute route;
        const uint32_t alt = pf_alternative(aut, ds, &route);
        PfOutcome outcome = PfOutcome::Done;
        acgpu_status result;
        bool probed_away = false;
        if (alt && ds->route_hint.load(std::memory_order_relaxed) > 0 && c.span_bytes >= kProbeMinSpan && !pf_uses_large_set(ds->hot, route)) {
            // recent scans of this automaton were abandoned: ask the probe first (256 samples of 8 KB through the filter)
            if ((st = ensure_probe(sc, c.stream))) return st;
            uint32_t* flag = reinterpret_cast<uint32_t*>(sc->probe.as<uint8_t>() + 64);
            HIP_TRY(launch_pf_probe(ds->hot, c.g, route, flag, sc->probe.as<unsigned long long>(), c.stream));
            HIP_TRY(sc->ensure_pinned());
            HIP_TRY(hipMemcpyAsync(sc->pinned, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipStreamSynchronize(c.stream));
            probed_away = (sc->pinned[0] & 0xFFFFFFFFull) != 0;
            if (probed_away) ds->route_hint.store(8, std::memory_order_relaxed);
            else ds->route_hint.fetch_sub(1, std::memory_order_relaxed);
        }
        if (probed_away) outcome = PfOutcome::Abandoned;
        else {
            if ((st = pf_events(c, route, &outcome, &result))) return st;
            if (outcome == PfOutcome::Done) return result;
            if (outcome == PfOutcome::Abandoned) ds->route_hint.store(8, std::memory_order_relaxed);
        }
        if (outcome == PfOutcome::Abandoned && alt == ENG_PF_LARGE) {   // same pipeline, the other filter
            c.routed = 1;
            c.force_large_set = true;
            PfRoute again;
            again.force_pfx = true;
            if ((st = pf_events(c, again, &outcome, &result))) return st;
            if (outcome == PfOutcome::Done) return result;
        } else if (outcome == PfOutcome::Abandoned && alt) { eng = alt; c.routed = 1; }
        // TooManyEvents: the chunk-counter form of the same filter below
    }
    return classic_pipeline(c, eng);
}

acgpu_status serial_impl(acgpu_automaton* aut, const acgpu_input* in, bool single, acgpu_match* out, size_t cap,
                         size_t* n_out, acgpu_profile* prof) {
    if (!aut || !n_out) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = 0;
    if (prof) std::memset(prof, 0, sizeof *prof);
    acgpu_status st = check_input(in);
    if (st) return st;
    const bool anchored = in->anchored != 0;
    if ((st = enforce_anchored_consistency(aut->cfg.start_kind, anchored))) return st;
    if ((st = check_start(aut, anchored))) return st;
    if (in->span_start > in->span_end) return ACGPU_OK;

    DeviceState* ds = nullptr;
    if ((st = get_device_state(aut, &ds))) return st;
    ScratchLease sc(ds);
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    if (prof && (st = ensure_events(sc.s.get()))) return st;
    const uint8_t* dhay = nullptr;
    if ((st = device_haystack(in, in->span_start, in->span_end, sc.s.get(), stream, &dhay))) return st;
    HIP_TRY(sc->totals.ensure(2 * sizeof(uint64_t)));

    uint64_t dev_cap = single ? 1 : std::min<uint64_t>(std::max<uint64_t>(cap, 1), 1ull << 20);
    for (;;) {
        acgpu_match* dout = nullptr;
        if (in->out_on_device && !single && cap <= dev_cap) dout = out;
        else { HIP_TRY(sc->result.ensure(dev_cap * sizeof(acgpu_match))); dout = sc->result.as<acgpu_match>(); }
        SerialArgs a{};
        a.hay = dhay; a.span_start = in->span_start; a.span_end = in->span_end;
        a.anchored = in->anchored; a.earliest = in->earliest; a.match_kind = aut->cfg.match_kind;
        a.out = dout; a.cap = dev_cap; a.n_out = sc->totals.as<uint64_t>();
        if (prof) HIP_TRY(hipEventRecord(sc->ev[0], stream));
        if (single) HIP_TRY(launch_find_serial(generic_engine(aut, ds), ds->da, a, stream));
        else HIP_TRY(launch_find_iter_serial(generic_engine(aut, ds), ds->da, a, stream));
        if (prof) HIP_TRY(hipEventRecord(sc->ev[1], stream));
        uint64_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, a.n_out, sizeof total, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        *n_out = size_t(total);
        if (prof) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, sc->ev[0], sc->ev[1]));
            prof->ms_scan = ms; prof->ms_total = ms;
            prof->bytes_scanned = in->span_end - in->span_start;
            prof->n_matches = total; prof->engine_used = generic_engine(aut, ds);
        }
        if (total > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
        if (total > dev_cap) { dev_cap = total; continue; }  // grow the staging buffer and rerun
        if (total && dout != out) {
            if (!out) return ACGPU_ERR_INVALID_ARGUMENT;
            HIP_TRY(hipMemcpyAsync(out, dout, total * sizeof(acgpu_match),
                                   in->out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        return ACGPU_OK;
    }
}

// Parallel find_iter: enumerate every occurrence with the chunked overlapping pipeline (all CUs), then select the
// non-overlapping matches from the ordered stream (device/select.hpp).  Eligible when the reference semantics are a
// function of the occurrence set: unanchored search, at least one pattern, no empty pattern.
bool parallel_find_eligible(const acgpu_automaton* aut, const acgpu_input* in) {
    if (in->anchored) return false;
    if (aut->nnfa.pattern_lens.empty() || aut->nnfa.min_pattern_len == 0) return false;
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD && !aut->occ) return false;
    const acgpu_automaton* o = aut->occ ? aut->occ.get() : aut;
    return o->cfg.start_kind != ACGPU_START_ANCHORED;
}

// Core of the parallel find_iter: occurrences whose end lies in (shard_begin, shard_end] (the whole span when the
// shard is the span), selection starting at position pos0.  The chosen records are left in sc->sel (device);
// *n_sel receives their number.
acgpu_status nonoverlapping_core(acgpu_automaton* occ, DeviceState* ds, Scratch* sc, const acgpu_input* in,
                                 size_t shard_begin, size_t shard_end, size_t pos0, int rule_kind, uint64_t* n_sel,
                                 acgpu_profile* prof) {
    *n_sel = 0;
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    acgpu_input oin = *in;
    oin.anchored = 0; oin.earliest = 0; oin.out_on_device = 0;
    size_t m_total = 0;
    acgpu_match* dS = nullptr;
    acgpu_status st = overlapping_impl(occ, &oin, shard_begin, shard_end, nullptr, 0, &m_total, prof, sc, &dS);
    if (st) return st;
    if (m_total == 0) return ACGPU_OK;
    // selection on the device (select.hip): succ pointers for all occurrences in parallel, block-wise orbit, ordered
    // compaction; the single-lane form only for streams beyond the u32 index range
    HIP_TRY(sc->sel.ensure(m_total * sizeof(acgpu_match)));
    uint64_t* d_tot = sc->totals.as<uint64_t>();  // [0] = number of stream records (written by the scan)
    if (m_total < 0xFFFFFFF0ull) {
        HIP_TRY(sc->selwork.ensure(select_scratch_bytes(m_total)));
        HIP_TRY(sc->seltot.ensure(2 * sizeof(uint64_t)));
        HIP_TRY(hipMemcpyAsync(sc->seltot.p, d_tot, sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));  // n_in survives the scan below
        const uint64_t nblk = (m_total + 1023) / 1024;
        HIP_TRY(sc->counts.ensure(nblk * sizeof(uint32_t) + 16));
        HIP_TRY(sc->offsets.ensure(nblk * sizeof(uint64_t)));
        HIP_TRY(sc->active.ensure(nblk * sizeof(uint64_t)));
        HIP_TRY(sc->aoff.ensure(nblk * sizeof(uint64_t)));
        HIP_TRY(sc->bsum.ensure(((nblk + 255) / 256 + 1) * sizeof(uint64_t)));
        HIP_TRY(sc->bact.ensure(((nblk + 255) / 256 + 1) * sizeof(uint32_t)));
        ScanScratch ss;
        ss.counts = sc->counts.as<uint32_t>(); ss.offsets = sc->offsets.as<uint64_t>(); ss.active = sc->active.as<uint64_t>();
        ss.aoff = sc->aoff.as<uint64_t>(); ss.bsum = sc->bsum.as<uint64_t>(); ss.bact = sc->bact.as<uint32_t>(); ss.totals = d_tot;
        HIP_TRY(launch_select_parallel(dS, m_total, sc->seltot.as<uint64_t>(), rule_kind, pos0,
                                       occ->nnfa.max_pattern_len, sc->selwork.p, ss, sc->sel.as<acgpu_match>(), m_total,
                                       stream));
        HIP_TRY(hipMemcpyAsync(n_sel, d_tot, sizeof *n_sel, hipMemcpyDeviceToHost, stream));
    } else {
        HIP_TRY(launch_select_nonoverlapping(dS, d_tot, rule_kind, pos0, occ->nnfa.max_pattern_len,
                                             sc->sel.as<acgpu_match>(), m_total, d_tot + 1, stream));
        HIP_TRY(hipMemcpyAsync(n_sel, d_tot + 1, sizeof *n_sel, hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (prof) prof->n_matches = *n_sel;
    return ACGPU_OK;
}

acgpu_status nonoverlapping_parallel(acgpu_automaton* aut, const acgpu_input* in, int rule_kind, acgpu_match* out,
                                     size_t cap, size_t* n_out, acgpu_profile* prof) {
    *n_out = 0;
    acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(occ, &ds);
    if (st) return st;
    ScratchLease sc(ds);
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    uint64_t n_sel = 0;
    if ((st = nonoverlapping_core(occ, ds, sc.s.get(), in, in->span_start, in->span_end, in->span_start, rule_kind,
                                  &n_sel, prof)))
        return st;
    *n_out = size_t(n_sel);
    if (n_sel > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (n_sel == 0) return ACGPU_OK;
    if (!out) return ACGPU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipMemcpyAsync(out, sc->sel.p, n_sel * sizeof(acgpu_match),
                           in->out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return ACGPU_OK;
}

// find_iter when the occurrence stream of the whole span does not fit in device memory (small alphabets x thousands of
// patterns: thousands of occurrences per byte).  The span is processed in windows: window (pos, b] yields the
// occurrences that end in it, the selection runs from `pos`, and the selected matches are final
//   - always for Standard (a later occurrence ends later, the rule takes the earliest end),
//   - for the leftmost kinds when start + L <= b (an unseen occurrence ends after b, hence starts after b - L);
// the next window starts at the end of the last final match, or at b + 1 - L if that is later (no candidate starts
// before it).  A window that still does not fit is retried at an eighth of its size.
acgpu_status nonoverlapping_windowed(acgpu_automaton* aut, const acgpu_input* in, int rule, acgpu_match* out, size_t cap,
                                     size_t* n_out) {
    *n_out = 0;
    acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(occ, &ds);
    if (st) return st;
    ScratchLease sc(ds);
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    const uint64_t L = std::max<uint64_t>(occ->nnfa.max_pattern_len, 1);
    const uint64_t w_min = std::max<uint64_t>(4 * L, 4096);
    auto trim = [&]() {
        for (DevBuf* b : {&sc->result, &sc->sel, &sc->selwork, &sc->events, &sc->eswork})
            if (b->bytes > (size_t(1) << 30)) b->release();
    };
    trim();
    uint64_t pos = in->span_start;
    uint64_t w = std::max<uint64_t>(w_min, std::min<uint64_t>((in->span_end - in->span_start) / 4, uint64_t(64) << 20));
    size_t total = 0;
    bool grow = true;
    std::vector<acgpu_match> tail;
    while (pos < in->span_end) {
        const uint64_t b = std::min<uint64_t>(in->span_end, pos + w);
        const bool last = b == in->span_end;
        uint64_t n_sel = 0;
        st = nonoverlapping_core(occ, ds, sc.s.get(), in, size_t(pos), size_t(b), size_t(pos), rule, &n_sel, nullptr);
        if (st == ACGPU_ERR_NOMEM && !g_too_dense && w > w_min) { trim(); w = std::max<uint64_t>(w / 8, w_min); grow = false; continue; }
        if (st) return st;
        const uint64_t floor_next = b + 1 > L ? b + 1 - L : 0;   // no unseen occurrence starts before this
        uint64_t n_acc = n_sel, last_end = pos;
        if (n_sel) {
            const uint64_t t = (rule == ACGPU_MATCH_STANDARD || last) ? 1 : std::min<uint64_t>(n_sel, L);
            tail.resize(t);
            HIP_TRY(hipMemcpyAsync(tail.data(), sc->sel.as<acgpu_match>() + (n_sel - t), t * sizeof(acgpu_match),
                                   hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            uint64_t k = t;   // records of the tail that are final
            if (rule != ACGPU_MATCH_STANDARD && !last)
                while (k > 0 && tail[k - 1].start + L > b) k--;
            n_acc = n_sel - (t - k);
            if (k > 0) last_end = tail[k - 1].end;
            else if (n_acc > 0) {   // the whole tail was dropped but earlier records stay: read the last one kept
                acgpu_match m{};
                HIP_TRY(hipMemcpyAsync(&m, sc->sel.as<acgpu_match>() + (n_acc - 1), sizeof m, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                last_end = m.end;
            }
        }
        if (n_acc) {
            if (out && total + n_acc <= cap)
                HIP_TRY(hipMemcpyAsync(out + total, sc->sel.p, n_acc * sizeof(acgpu_match),
                                       in->out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
            total += n_acc;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        pos = last ? in->span_end : std::max<uint64_t>(n_acc ? last_end : pos, floor_next);
        if (grow && w < (uint64_t(1) << 30)) w *= 2;   // (a window size that failed once is not tried again)
    }
    trim();
    *n_out = total;
    if (total > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (total && !out) return ACGPU_ERR_INVALID_ARGUMENT;
    return ACGPU_OK;
}

// Argument checks shared by the non-overlapping entry points (same order as the reference facade).
acgpu_status check_nonoverlapping(acgpu_automaton* aut, const acgpu_input* in) {
    if (!aut) return ACGPU_ERR_INVALID_ARGUMENT;
    acgpu_status st = check_input(in);
    if (st) return st;
    if ((st = enforce_anchored_consistency(aut->cfg.start_kind, in->anchored != 0))) return st;
    return check_start(aut, in->anchored != 0);
}

// ---- host haystacks: copy / scan overlap ---------------------------------------------------------------------------
// A helper thread copies the haystack to the device piece by piece on its own stream (a hipMemcpyAsync from pageable
// memory returns only when the runtime has staged the source, so it has to be a thread, not just a second stream) and
// records one event per piece; the caller waits for piece k (condition variable, then hipStreamWaitEvent on its compute
// stream) and scans it while pieces k+1.. are still crossing PCIe.  Mirrors the roll buffer of the reference's stream
// searcher (src/util/buffer.rs:113-123: keep min_buffer_len bytes, refill behind the search), with the search side on
// all CUs: only the last piece's scan is not hidden by a copy.
struct HostPipe {
    int device = 0;
    uint8_t* dst = nullptr;
    const uint8_t* src = nullptr;
    size_t len = 0, piece = 0, n_pieces = 0;
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> ev;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    size_t submitted = 0;          // pieces whose copy is enqueued and whose event is recorded
    hipError_t err = hipSuccess;

    hipError_t start(int dev, uint8_t* d, const uint8_t* s, size_t n, size_t piece_bytes) {
        device = dev; dst = d; src = s; len = n; piece = piece_bytes;
        n_pieces = (n + piece - 1) / piece;
        hipError_t e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        ev.assign(n_pieces, nullptr);
        for (auto& x : ev) if ((e = hipEventCreateWithFlags(&x, hipEventDisableTiming)) != hipSuccess) return e;
        th = std::thread([this] {
            hipError_t e2 = hipSetDevice(device);
            for (size_t k = 0; k < n_pieces; k++) {
                const size_t off = k * piece, nb = std::min(piece, len - off);
                if (e2 == hipSuccess) e2 = hipMemcpyAsync(dst + off, src + off, nb, hipMemcpyHostToDevice, copy_stream);
                if (e2 == hipSuccess) e2 = hipEventRecord(ev[k], copy_stream);
                std::lock_guard<std::mutex> lk(mu);
                if (e2 != hipSuccess && err == hipSuccess) err = e2;
                submitted = k + 1;
                cv.notify_all();
            }
        });
        return hipSuccess;
    }
    // blocks until piece k's copy has been enqueued, then orders `compute` behind it
    hipError_t wait(size_t k, hipStream_t compute) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return submitted > k; });
        if (err != hipSuccess) return err;
        lk.unlock();
        return hipStreamWaitEvent(compute, ev[k], 0);
    }
    ~HostPipe() {
        if (th.joinable()) th.join();
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
        for (auto& x : ev) if (x) (void)hipEventDestroy(x);
    }
};

size_t host_piece_bytes() {
    const char* e = std::getenv("ACGPU_HOST_PIECE_MIB");   // tuning / test knob
    const size_t mib = e ? size_t(std::atoi(e)) : 256;   // (64 MiB pieces measured 1 ms slower per 2 GiB than one copy: per-copy setup)
    return std::max<size_t>(mib, 1) << 20;
}

// acgpu_find_overlapping* over a HOST haystack that is large enough to be worth pipelining: the span is searched piece
// by piece (consecutive shards: the concatenation is the full stream by the seam rule), each as soon as its bytes have
// arrived.  Argument checks in the same order as overlapping_impl.
acgpu_status overlapping_host_pipelined(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end,
                                        acgpu_match* out, size_t cap, size_t* n_out, acgpu_profile* prof) {
    *n_out = 0;
    if (prof) std::memset(prof, 0, sizeof *prof);
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(aut, &ds);
    if (st) return st;
    ScratchLease stage(ds);   // holds the device copy of the haystack for the whole call
    const size_t halo = aut->nnfa.max_pattern_len > 0 ? aut->nnfa.max_pattern_len - 1 : 0;
    const size_t need_lo = std::max(in->span_start, shard_begin >= halo ? shard_begin - halo : size_t(0));
    const size_t n = shard_end - need_lo;
    HIP_TRY(stage->hay.ensure(n + 64));
    uint8_t* dbuf = stage->hay.as<uint8_t>();
    const size_t piece = host_piece_bytes();
    HostPipe pipe;
    HIP_TRY(pipe.start(ds->device, dbuf, in->haystack + need_lo, n, piece));
    acgpu_input din = *in;
    din.haystack = dbuf - need_lo;   // din.haystack[i] is haystack byte i for i in [need_lo, shard_end)
    din.haystack_on_device = 1;
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    size_t total = 0;
    for (size_t k = 0; k < pipe.n_pieces; k++) {
        HIP_TRY(pipe.wait(k, stream));
        const size_t pb = std::max(shard_begin, need_lo + k * piece), pe = std::min(shard_end, need_lo + (k + 1) * piece);
        if (pe <= pb && !(k == 0 && shard_begin == shard_end)) continue;
        size_t m = 0;
        acgpu_profile pp;
        const bool room = out && total < cap;
        st = overlapping_impl(aut, &din, pb, pe, room ? out + total : nullptr, room ? cap - total : 0, &m, prof ? &pp : nullptr);
        if (st != ACGPU_OK && st != ACGPU_ERR_BUFFER_TOO_SMALL) return st;
        total += m;
        if (prof) {
            prof->ms_scan += pp.ms_scan; prof->ms_compact += pp.ms_compact; prof->ms_fill += pp.ms_fill; prof->ms_total += pp.ms_total;
            prof->bytes_scanned += pp.bytes_scanned; prof->n_chunks += pp.n_chunks; prof->n_active_chunks += pp.n_active_chunks;
            prof->n_matches += pp.n_matches; prof->engine_used = pp.engine_used; prof->routed |= pp.routed;
        }
    }
    *n_out = total;
    if (total > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (total && !out) return ACGPU_ERR_INVALID_ARGUMENT;
    return ACGPU_OK;
}

// routes a host-haystack / host-output call through the pipelined form when it pays (two pieces or more)
acgpu_status overlapping_entry(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end,
                               acgpu_match* out, size_t cap, size_t* n_out, acgpu_profile* prof) {
    if (aut && n_out && in && in->haystack && !in->haystack_on_device && !in->out_on_device && check_input(in) == ACGPU_OK &&
        in->span_start <= shard_begin && shard_begin <= shard_end && shard_end <= in->span_end &&
        shard_end - shard_begin >= 2 * host_piece_bytes() && aut->cfg.match_kind == ACGPU_MATCH_STANDARD && !in->anchored &&
        enforce_anchored_consistency(aut->cfg.start_kind, false) == ACGPU_OK && check_start(aut, false) == ACGPU_OK) {
        acgpu_automaton* target = (aut->cfg.start_kind == ACGPU_START_BOTH && aut->occ) ? aut->occ.get() : aut;
        return overlapping_host_pipelined(target, in, shard_begin, shard_end, out, cap, n_out, prof);
    }
    return overlapping_impl(aut, in, shard_begin, shard_end, out, cap, n_out, prof);
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI
extern "C" {

uint32_t acgpu_abi_version(void) { return ACGPU_ABI_VERSION; }

const char* acgpu_last_error(void) { return g_last_error.c_str(); }

const char* acgpu_status_str(acgpu_status s) {
    switch (s) {
        case ACGPU_OK: return "ok";
        case ACGPU_ERR_STATE_ID_OVERFLOW: return "state identifier overflow";
        case ACGPU_ERR_PATTERN_ID_OVERFLOW: return "pattern identifier overflow";
        case ACGPU_ERR_PATTERN_TOO_LONG: return "pattern too long";
        case ACGPU_ERR_INVALID_INPUT_ANCHORED: return "anchored searches are not supported or enabled";
        case ACGPU_ERR_INVALID_INPUT_UNANCHORED: return "unanchored searches are not supported or enabled";
        case ACGPU_ERR_UNSUPPORTED_STREAM: return "match kind does not support stream searching";
        case ACGPU_ERR_UNSUPPORTED_OVERLAPPING: return "match kind does not support overlapping searches";
        case ACGPU_ERR_UNSUPPORTED_EMPTY: return "matching with an empty pattern string is not supported for this operation";
        case ACGPU_ERR_INVALID_SPAN: return "invalid span for haystack";
        case ACGPU_ERR_BUFFER_TOO_SMALL: return "output buffer too small";
        case ACGPU_ERR_INVALID_ARGUMENT: return "invalid argument";
        case ACGPU_ERR_NOMEM: return "out of memory";
        case ACGPU_ERR_HIP: return "HIP runtime error";
        case ACGPU_ERR_NO_DEVICE: return "no usable HIP device";
    }
    return "unknown";
}

void acgpu_config_init(acgpu_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->match_kind = ACGPU_MATCH_STANDARD;
    c->start_kind = ACGPU_START_UNANCHORED;
    c->kind = ACGPU_KIND_AUTO;
    c->byte_classes = 1;
    c->prefilter = 1;
}

acgpu_status acgpu_build(const acgpu_config* cfg_in, const uint8_t* const* patterns, const size_t* lens, size_t n,
                         acgpu_automaton** out) {
    if (!out) return ACGPU_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    acgpu_config cfg;
    if (cfg_in) cfg = *cfg_in; else acgpu_config_init(&cfg);
    if (n && (!patterns || !lens)) return ACGPU_ERR_INVALID_ARGUMENT;
    if (cfg.match_kind < 0 || cfg.match_kind > 2 || cfg.start_kind < 0 || cfg.start_kind > 2 || cfg.kind < 0 || cfg.kind > 3)
        return ACGPU_ERR_INVALID_ARGUMENT;
    std::unique_ptr<acgpu_automaton> a;
    try {
        a = std::make_unique<acgpu_automaton>();
        a->cfg = cfg;
        BuildOptions o;
        o.match_kind = cfg.match_kind;
        o.ascii_case_insensitive = cfg.ascii_case_insensitive != 0;
        o.byte_classes = cfg.byte_classes != 0;
        o.start_kind = cfg.start_kind;
        if (cfg.dense_depth_set) {  // AhoCorasickBuilder::dense_depth sets both, ahocorasick.rs:2581-2585
            size_t dd = cfg.dense_depth == UINT32_MAX ? SIZE_MAX : cfg.dense_depth;
            o.nnfa_dense_depth = dd; o.cnfa_dense_depth = dd;
        }
        acgpu_status st = build_nnfa(o, patterns, lens, n, a->nnfa);
        if (st) return st;
        // opt-in: DFA rows computed on the device (needs a HIP device at build time; identical table)
        DfaRowFill dfa_fill = nullptr;
        if (cfg.gpu_dfa_fill)
            dfa_fill = [](const NNfa& nn, const uint8_t* classes, size_t alen, size_t s2, bool anchored, uint32_t* trans) {
                const hipError_t e = device_fill_dfa(nn, classes, alen, s2, anchored, trans);
                if (e != hipSuccess) g_last_error = std::string("device_fill_dfa: ") + hipGetErrorString(e);
                return e == hipSuccess;
            };
        int kind = cfg.kind;
        if (kind == ACGPU_KIND_AUTO) {  // build_auto, ahocorasick.rs:2213-2261
            const bool try_dfa = cfg.start_kind != ACGPU_START_BOTH && a->nnfa.pattern_lens.size() <= 100;
            if (try_dfa && build_dfa(a->nnfa, cfg.start_kind, o.byte_classes, a->dfa, dfa_fill) == ACGPU_OK) {
                a->has_dfa = true; kind = ACGPU_KIND_DFA;
            } else if (build_cnfa(a->nnfa, o.cnfa_dense_depth, o.byte_classes, a->cnfa) == ACGPU_OK) {
                a->dfa = Dfa(); a->has_cnfa = true; kind = ACGPU_KIND_CONTIGUOUS_NFA;
            } else {
                a->cnfa = CNfa(); kind = ACGPU_KIND_NONCONTIGUOUS_NFA;
            }
        } else if (kind == ACGPU_KIND_DFA) {
            if ((st = build_dfa(a->nnfa, cfg.start_kind, o.byte_classes, a->dfa, dfa_fill))) return st;
            a->has_dfa = true;
        } else if (kind == ACGPU_KIND_CONTIGUOUS_NFA) {
            if ((st = build_cnfa(a->nnfa, o.cnfa_dense_depth, o.byte_classes, a->cnfa))) return st;
            a->has_cnfa = true;
        }
        if (kind == ACGPU_KIND_NONCONTIGUOUS_NFA) {
            // The noncontiguous NFA's own search path (noncontiguous.rs:601-626) is the slowest encoding of
            // the same automaton; on the device it is walked in its contiguous encoding (identical results).
            if ((st = build_cnfa(a->nnfa, 2, true, a->cnfa))) return st;
            a->has_cnfa = true;
        }
        a->kind = kind;
        // leftmost kinds: also build the Standard automaton of the same patterns (see acgpu_automaton::occ)
        // StartKind::Both with Standard semantics: the same twin serves the unanchored searches (overlapping_impl)
        const bool twin_for_both = cfg.match_kind == ACGPU_MATCH_STANDARD && cfg.start_kind == ACGPU_START_BOTH && cfg.engine != 1;
        if ((cfg.match_kind != ACGPU_MATCH_STANDARD || twin_for_both) && n > 0 && a->nnfa.min_pattern_len > 0 &&
            cfg.start_kind != ACGPU_START_ANCHORED) {
            acgpu_config oc = cfg;
            oc.match_kind = ACGPU_MATCH_STANDARD;
            oc.start_kind = ACGPU_START_UNANCHORED;
            oc.byte_classes = 1;
            oc.dense_depth_set = 0;
            // full DFA while its table stays below ~1 GiB (u32 x stride <= 256 per state), else contiguous NFA
            oc.kind = a->nnfa.states() <= (size_t(1) << 20) ? ACGPU_KIND_DFA : ACGPU_KIND_CONTIGUOUS_NFA;
            acgpu_automaton* o = nullptr;
            acgpu_status ost = acgpu_build(&oc, patterns, lens, n, &o);
            if (ost == ACGPU_OK) a->occ.reset(o);
        }
    } catch (const std::bad_alloc&) {
        return ACGPU_ERR_NOMEM;
    }
    *out = a.release();
    return ACGPU_OK;
}

// Bounds-checked debug build (make -C csrc guard): haystack accesses outside the 16-byte-aligned hull of the searched
// span, summed over all devices used so far, since the process started.  -1: this is not a guard build.
long long acgpu_guard_violations(void) {
#ifdef ACGPU_GUARD
    std::lock_guard<std::mutex> lk(g_guard_mu);
    long long total = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& kv : g_guard_ctrs) {
        if (!kv.second) continue;
        unsigned long long v = 0;
        if (hipSetDevice(kv.first) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(&v, kv.second, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) { total = -2; break; }
        total += static_cast<long long>(v);
    }
    (void)hipSetDevice(cur);
    return total;
#else
    return -1;
#endif
}

void acgpu_free(acgpu_automaton* aut) { delete aut; }

int32_t acgpu_kind_of(const acgpu_automaton* a) { return a->kind; }
int32_t acgpu_match_kind_of(const acgpu_automaton* a) { return a->cfg.match_kind; }
int32_t acgpu_start_kind_of(const acgpu_automaton* a) { return a->cfg.start_kind; }
size_t acgpu_patterns_len(const acgpu_automaton* a) { return a->nnfa.pattern_lens.size(); }
size_t acgpu_min_pattern_len(const acgpu_automaton* a) { return a->nnfa.min_pattern_len; }
size_t acgpu_max_pattern_len(const acgpu_automaton* a) { return a->nnfa.max_pattern_len; }

// dfa.rs:289-297, contiguous.rs:310-316, noncontiguous.rs:689-696 (prefilter term is always 0 here)
size_t acgpu_memory_usage(const acgpu_automaton* a) {
    const size_t pl = a->nnfa.pattern_lens.size() * 4;
    switch (a->kind) {
        case ACGPU_KIND_DFA:
            return a->dfa.trans.size() * 4 + a->dfa.num_match_states * 24 + a->dfa.mpid.size() * 4 + pl;
        case ACGPU_KIND_CONTIGUOUS_NFA:
            return a->cnfa.repr.size() * 4 + pl;
        default:
            return a->nnfa.states() * 20 + (a->nnfa.tbyte.size() + 1) * 9 + (a->nnfa.mpid.size() + 1) * 8 +
                   (1 + a->nnfa.dense_states * a->nnfa.alphabet_len()) * 4 + pl;
    }
}

acgpu_status acgpu_upload(acgpu_automaton* aut, int device) {
    if (!aut) return ACGPU_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(aut->mu);
    if (aut->devs.count(device)) return ACGPU_OK;
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    auto ds = std::make_unique<DeviceState>();
    ds->device = device;
    acgpu_status st = ACGPU_OK;
    auto body = [&]() -> acgpu_status {
        HIP_TRY(ds->plens.upload(aut->nnfa.pattern_lens));
        auto upload_dfa = [&](const Dfa& d, bool unanchored_standard) -> acgpu_status {
            HIP_TRY(ds->dfa_trans.upload(d.trans));
            HIP_TRY(ds->dfa_moff.upload(d.moff));
            HIP_TRY(ds->dfa_mpid.upload(d.mpid));
            std::vector<uint8_t> cls(d.byte_classes, d.byte_classes + 256);
            HIP_TRY(ds->dfa_cls.upload(cls));
            ds->da.has_dfa = true;
            ds->da.dfa.trans = ds->dfa_trans.as<uint32_t>();
            ds->da.dfa.moff = ds->dfa_moff.as<uint32_t>();
            ds->da.dfa.mpid = ds->dfa_mpid.as<uint32_t>();
            ds->da.dfa.plens = ds->plens.as<uint32_t>();
            ds->da.dfa.classes = ds->dfa_cls.as<uint8_t>();
            ds->da.dfa.stride2 = uint32_t(d.stride2);
            ds->da.dfa.sp = {d.special.max_special_id, d.special.max_match_id, d.special.start_unanchored_id,
                             d.special.start_anchored_id};
            // LDS-resident fast path: Standard semantics, unanchored start
            // (StartKind::Both interleaves anchored copies, dfa.rs:617-724: generic walk only)
            if (unanchored_standard && aut->cfg.engine != 1) {
                hipError_t e = build_hot_tables(aut->nnfa, d, ds->hot);
                if (e != hipSuccess) return hip_fail(e, "build_hot_tables");
            }
            if (aut->cfg.match_kind == ACGPU_MATCH_STANDARD && aut->cfg.start_kind == ACGPU_START_UNANCHORED) {
                // the walk engine of this table: the shallow-skip form of the transition walk (single-start layout)
                hipError_t e = build_dfa_tri(aut->nnfa, d, ds->da.dfa.moff, ds->dfa_tri);
                if (e != hipSuccess) return hip_fail(e, "build_dfa_tri");
            }
            return ACGPU_OK;
        };
        const bool std_unanchored = aut->cfg.match_kind == ACGPU_MATCH_STANDARD && aut->cfg.start_kind == ACGPU_START_UNANCHORED;
        if (aut->has_dfa) {
            acgpu_status ust = upload_dfa(aut->dfa, std_unanchored);
            if (ust) return ust;
        } else if (aut->cfg.engine != 1 && aut->cfg.start_kind == ACGPU_START_UNANCHORED && !aut->nnfa.pattern_lens.empty()) {
            // NFA-kind automaton (the reference's choice beyond 100 patterns, ahocorasick.rs:2213-2261): HBM has room
            // for the full DFA of the same noncontiguous NFA, so the device walks that instead of failure links --
            // identical results (same construction as DFA::build, rows filled on the device), one load per byte, and
            // the LDS engines become available.  acgpu_kind_of / the host tables still describe the reference's kind.
            const size_t alen = aut->nnfa.alphabet_len();
            size_t s2 = 0;
            while ((size_t(1) << s2) < alen) s2++;
            const uint64_t bytes = (uint64_t(aut->nnfa.states()) << s2) * 4;
            if (bytes <= (uint64_t(2) << 30)) {
                Dfa tmp;
                DfaRowFill fill = [](const NNfa& nn, const uint8_t* classes, size_t al, size_t st2, bool anchored, uint32_t* trans) {
                    return device_fill_dfa(nn, classes, al, st2, anchored, trans) == hipSuccess;
                };
                if (build_dfa(aut->nnfa, ACGPU_START_UNANCHORED, true, tmp, fill) == ACGPU_OK) {
                    acgpu_status ust = upload_dfa(tmp, std_unanchored && bytes <= (uint64_t(512) << 20));
                    if (ust) return ust;
                    ds->derived_dfa = true;
                }
            }
        }
        if (aut->has_cnfa) {
            {   // padded: the fast walk loads repr[sid + 2 + class] before it knows the state is dense (cnfa_walk.hip)
                std::vector<uint32_t> padded(aut->cnfa.repr);
                padded.resize(padded.size() + kCnfaReprPad, 0);
                HIP_TRY(ds->cnfa_repr.upload(padded));
            }
            if (aut->cfg.match_kind == ACGPU_MATCH_STANDARD && aut->cfg.start_kind != ACGPU_START_ANCHORED) {
                const hipError_t he = build_cnfa_hot(aut->cnfa, ds->cnfa_hot);
                if (he != hipSuccess) return hip_fail(he, "build_cnfa_hot");
                const hipError_t ht = build_cnfa_tri(aut->cnfa, ds->cnfa_tri);
                if (ht != hipSuccess) return hip_fail(ht, "build_cnfa_tri");
            }
            std::vector<uint8_t> cls(aut->cnfa.byte_classes, aut->cnfa.byte_classes + 256);
            HIP_TRY(ds->cnfa_cls.upload(cls));
            ds->da.has_cnfa = true;
            ds->da.cnfa.repr = ds->cnfa_repr.as<uint32_t>();
            ds->da.cnfa.plens = ds->plens.as<uint32_t>();
            ds->da.cnfa.classes = ds->cnfa_cls.as<uint8_t>();
            ds->da.cnfa.alphabet_len = uint32_t(aut->cnfa.alphabet_len);
            ds->da.cnfa.sp = {aut->cnfa.special.max_special_id, aut->cnfa.special.max_match_id,
                              aut->cnfa.special.start_unanchored_id, aut->cnfa.special.start_anchored_id};
        }
        return ACGPU_OK;
    };
    st = body();
    (void)hipSetDevice(prev);
    if (st) return st;
    aut->devs[device] = std::move(ds);
    return ACGPU_OK;
}

acgpu_status acgpu_find_overlapping(acgpu_automaton* aut, const acgpu_input* in, acgpu_match* out, size_t cap,
                                    size_t* n_out) {
    if (!in) return ACGPU_ERR_INVALID_ARGUMENT;
    return overlapping_entry(aut, in, in->span_start, in->span_end, out, cap, n_out, nullptr);
}
acgpu_status acgpu_find_overlapping_ex(acgpu_automaton* aut, const acgpu_input* in, acgpu_match* out, size_t cap,
                                       size_t* n_out, acgpu_profile* prof) {
    if (!in) return ACGPU_ERR_INVALID_ARGUMENT;
    return overlapping_entry(aut, in, in->span_start, in->span_end, out, cap, n_out, prof);
}
acgpu_status acgpu_find_overlapping_shard(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin,
                                          size_t shard_end, acgpu_match* out, size_t cap, size_t* n_out,
                                          acgpu_profile* prof) {
    return overlapping_entry(aut, in, shard_begin, shard_end, out, cap, n_out, prof);
}

acgpu_status acgpu_find_overlapping_enqueue_ex(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin,
                                               size_t shard_end, acgpu_match* out, size_t cap, uint64_t* totals,
                                               int32_t slot, uint32_t flags) {
    if (!aut || !totals || slot >= 64) return ACGPU_ERR_INVALID_ARGUMENT;
    acgpu_status st = check_input(in);
    if (st) return st;
    // the same argument checks, in the same order, as the synchronous form
    if ((st = enforce_anchored_consistency(aut->cfg.start_kind, in->anchored != 0))) return st;
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD) return ACGPU_ERR_UNSUPPORTED_OVERLAPPING;
    if (in->anchored) return ACGPU_ERR_INVALID_INPUT_ANCHORED;
    if ((st = check_start(aut, false))) return st;
    if (!(in->span_start <= shard_begin && shard_begin <= shard_end && shard_end <= in->span_end))
        return ACGPU_ERR_INVALID_ARGUMENT;
    if (aut->cfg.start_kind == ACGPU_START_BOTH && aut->occ)
        return acgpu_find_overlapping_enqueue_ex(aut->occ.get(), in, shard_begin, shard_end, out, cap, totals, slot, flags);
    DeviceState* ds = nullptr;
    if ((st = get_device_state(aut, &ds))) return st;
    if (!in->haystack_on_device || !(cap == 0 || out)) {
        g_last_error = "enqueue form: device haystack and device output required";
        return ACGPU_ERR_INVALID_ARGUMENT;
    }
    const size_t halo = aut->nnfa.max_pattern_len > 0 ? aut->nnfa.max_pattern_len - 1 : 0;
    if (halo > 0xFFFFFF00ull) return ACGPU_ERR_INVALID_ARGUMENT;
    // engine choice as in overlapping_impl (no routing target is needed here: see below)
    uint32_t eng = generic_engine(aut, ds);
    const int want = aut->cfg.engine;
    if (eng == ENG_DFA) {
        if ((want == 0 || want == 3) && ds->hot.pf_ready) eng = ENG_PF;
        else if (((want == 0 && aut->nnfa.min_pattern_len > 0) || want == 2) && ds->hot.lw_ready) eng = ENG_HOT;
    }
    if ((want == 2 && eng != ENG_HOT) || (want == 3 && eng != ENG_PF)) {
        g_last_error = "requested engine is unavailable for this automaton";
        return ACGPU_ERR_INVALID_ARGUMENT;
    }
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    DeviceState::AsyncCtx* ctx = ds->async_ctx(stream);
    Scratch* sc = &ctx->sc;
    const ScanGeom g = make_geom(aut, in, shard_begin, shard_end, in->haystack, halo);
    if (slot >= 0) {
        for (int k = 0; k < 2; k++) if (!ctx->ev[2 * slot + k]) HIP_TRY(hipEventCreate(&ctx->ev[2 * slot + k]));
        HIP_TRY(hipEventRecord(ctx->ev[2 * slot], stream));
    }
    if (in->span_start > in->span_end) {   // Input::is_done: no matches
        HIP_TRY(hipMemsetAsync(totals, 0, 2 * sizeof(uint64_t), stream));
        if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ev[2 * slot + 1], stream));
        return ACGPU_OK;
    }
    const bool events_form = eng == ENG_PF && aut->nnfa.max_pattern_len <= 0xFFFF && !(flags & ACGPU_ENQUEUE_CLASSIC);
    if (events_form) {
        // sparse results (up to ACGPU_ENQUEUE_MAX_EVENTS occurrences): filter scan -> all-pairs rank -> ordered records
        constexpr uint32_t kEvCap = ACGPU_ENQUEUE_MAX_EVENTS;
        static_assert(kEvCap == kEvAllPairs, "the enqueue form orders its events with the all-pairs rank");
        // (the event buffer is sized like the synchronous form's: beyond kEvCap events the bucket order pass takes over)
        const uint64_t cap_ev = std::min<uint64_t>(kSortMaxEvents, std::max<uint64_t>(uint64_t(1) << 16, (g.emit_hi - g.emit_lo) / 64));
        HIP_TRY(sc->events.ensure(size_t(cap_ev) * pf_event_bytes()));
        HIP_TRY(sc->evrank.ensure(size_t(kEvCap) * sizeof(uint32_t)));
        HIP_TRY(sc->evctr.ensure(kPfCtrWords * sizeof(unsigned long long)));
        if (!sc->ev_armed) {   // first call on this stream, or an earlier call failed between the scan and k_ev_write
            HIP_TRY(hipMemsetAsync(sc->evrank.p, 0, size_t(kEvCap) * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(sc->evctr.p, 0, kPfCtrWords * sizeof(unsigned long long), stream));
        }
        sc->ev_armed = false;
        unsigned long long* ctr = sc->evctr.as<unsigned long long>();
        uint32_t* rank = sc->evrank.as<uint32_t>();
        // the routing rule of the synchronous form applies here too: an abandoned scan reports totals[1] = UINT64_MAX,
        // which the caller treats like an event overflow ("repeat with the synchronous call": that one switches engine
        // and remembers).  While the automaton is remembered as "recently abandoned" and its alternative is the large-set
        // filter -- natural text against a dictionary -- the probe decides on the device: both filters are enqueued, gated
        // on the probe's word, one of them returns at once.
        PfRoute route;
        const uint32_t alt = pf_alternative(aut, ds, &route);
        const uint64_t span_bytes = g.emit_hi - g.emit_lo;
        const bool probe = alt == ENG_PF_LARGE && ds->route_hint.load(std::memory_order_relaxed) > 0 && span_bytes >= kProbeMinSpan &&
                           !pf_uses_large_set(ds->hot, route);
        if (probe) {
            if ((st = ensure_probe(sc, stream))) return st;
            uint32_t* flag = reinterpret_cast<uint32_t*>(sc->probe.as<uint8_t>() + 64);
            HIP_TRY(launch_pf_probe(ds->hot, g, route, flag, sc->probe.as<unsigned long long>(), stream));
            route.gate = flag; route.gate_val = 0;
        }
        if ((st = pf_route_prepare(sc, ds->hot, span_bytes, &route))) return st;
        HIP_TRY(launch_pf_any(ds->hot, g, nullptr, stream, sc->events.p, ctr, cap_ev, route));
        if (probe) {
            PfRoute other;
            other.force_pfx = true; other.gate = route.gate; other.gate_val = 1;
            if ((st = pf_route_prepare(sc, ds->hot, span_bytes, &other))) return st;
            HIP_TRY(launch_pf_any(ds->hot, g, nullptr, stream, sc->events.p, ctr, cap_ev, other));
        }
        if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ev[2 * slot + 1], stream));
        HIP_TRY(launch_pf_event_rank(sc->events.p, ctr, kEvCap, rank, totals, kEvCap / 2, stream));   // (grid hint only: grid-stride kernel)
        HIP_TRY(launch_pf_event_write(ds->hot, ds->da, sc->events.p, ctr, kEvCap, rank, totals, out ? cap : 0, out, stream));
        sc->ev_armed = true;
        // more occurrences than the all-pairs rank orders: the bucket order pass (one launch, empty unless needed) delivers
        // them and resets totals[1] to 0
        if (out && cap) {
            const uint64_t max_rec = std::min<uint64_t>(cap, uint64_t(1) << 26);
            if ((st = ensure_order_work(sc, event_order_work_bytes(cap_ev, max_rec, span_bytes), stream))) return st;
            HIP_TRY(launch_event_order_emit(ds->hot, ds->da, sc->events.p, totals, kEvCap, cap_ev, max_rec, shard_begin, span_bytes,
                                            sc->eswork.p, out, stream, totals));
        }
        return ACGPU_OK;
    }
    // every other engine, and dense results on request (ACGPU_ENQUEUE_CLASSIC): chunk counters -> scan -> fill, all
    // reading their sizes on the device -- no occurrence limit, no host round trip
    const uint64_t nb = (g.n_chunks + 255) / 256;
    HIP_TRY(sc->counts.ensure(g.n_chunks * sizeof(uint32_t)));
    HIP_TRY(sc->active.ensure(g.n_chunks * sizeof(uint64_t)));
    HIP_TRY(sc->aoff.ensure(g.n_chunks * sizeof(uint64_t)));
    HIP_TRY(sc->bsum.ensure(nb * sizeof(uint64_t)));
    HIP_TRY(sc->bact.ensure(nb * sizeof(uint32_t)));
    HIP_TRY(sc->totals.ensure(2 * sizeof(uint64_t)));
    ScanScratch ss;
    ss.counts = sc->counts.as<uint32_t>(); ss.offsets = nullptr; ss.active = sc->active.as<uint64_t>();
    ss.aoff = sc->aoff.as<uint64_t>(); ss.bsum = sc->bsum.as<uint64_t>(); ss.bact = sc->bact.as<uint32_t>();
    ss.totals = sc->totals.as<uint64_t>();
    TriEvents tev;
    if (tri_walk_selected(eng, ds)) {
        if ((st = cnfa_tri_events(sc, g, g.emit_hi - g.emit_lo, stream, &tev))) return st;
        if (tev.ev) { HIP_TRY(sc->offsets.ensure(g.n_chunks * sizeof(uint64_t))); ss.offsets = sc->offsets.as<uint64_t>(); }
    }
    if (eng == ENG_PF) {
        PfRoute pfr;
        if ((st = pf_route_prepare(sc, ds->hot, g.emit_hi - g.emit_lo, &pfr))) return st;
        HIP_TRY(launch_pf_any(ds->hot, g, ss.counts, stream, nullptr, nullptr, 0, pfr));
    } else if (eng == ENG_HOT) HIP_TRY(launch_hot_count(ds->hot, ds->da, g, ss.counts, stream));
    else HIP_TRY(launch_generic_count(eng, ds, g, ss.counts, stream, &tev));
    if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ev[2 * slot + 1], stream));
    HIP_TRY(launch_scan(ss, g.n_chunks, stream));
    if (cap > 0 && out) {
        const uint32_t fill_eng = generic_engine(aut, ds);
        if (tev.ev) {   // shallow-skip walks: records from the events; the re-walking fill is gated on their overflow flag
            if (eng == ENG_CNFA) HIP_TRY(launch_cnfa_tri_emit(ds->cnfa_tri, ds->da.cnfa.plens, g, tev, ss.offsets, ss.totals, cap, out, stream));
            else HIP_TRY(launch_dfa_tri_emit(ds->dfa_tri, ds->da, g, tev, ss.offsets, ss.totals, cap, out, stream));
            HIP_TRY(launch_walk_fill(eng, ds->da, g, ss.active, ss.totals, cap, 16384, ss.aoff, out, stream, tev.ctr + 1));
        } else if (fill_eng == ENG_DFA && aut->cfg.engine != 1 && hot_fill_supported(ds->hot, g))
            HIP_TRY(launch_hot_fill(ds->hot, ds->da, g, ss.active, ss.totals, cap, 16384, ss.aoff, out, stream));
        else
            HIP_TRY(launch_walk_fill(fill_eng, ds->da, g, ss.active, ss.totals, cap, 16384, ss.aoff, out, stream));
    }
    HIP_TRY(hipMemcpyAsync(totals, ss.totals, sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));   // records
    HIP_TRY(hipMemsetAsync(totals + 1, 0, sizeof(uint64_t), stream));                                  // no event list, no event limit
    return ACGPU_OK;
}

acgpu_status acgpu_find_overlapping_enqueue(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin,
                                            size_t shard_end, acgpu_match* out, size_t cap, uint64_t* totals,
                                            int32_t slot) {
    return acgpu_find_overlapping_enqueue_ex(aut, in, shard_begin, shard_end, out, cap, totals, slot, 0);
}

acgpu_status acgpu_enqueue_kernel_ms(acgpu_automaton* aut, void* stream, int32_t slot, float* ms) {
    if (!aut || !ms || slot < 0 || slot >= 64) return ACGPU_ERR_INVALID_ARGUMENT;
    if (aut->cfg.start_kind == ACGPU_START_BOTH && aut->occ) return acgpu_enqueue_kernel_ms(aut->occ.get(), stream, slot, ms);
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(aut, &ds);
    if (st) return st;
    DeviceState::AsyncCtx* ctx = ds->async_ctx(static_cast<hipStream_t>(stream));
    if (!ctx->ev[2 * slot] || !ctx->ev[2 * slot + 1]) return ACGPU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev[2 * slot], ctx->ev[2 * slot + 1]));
    return ACGPU_OK;
}

acgpu_status acgpu_find_iter_ex(acgpu_automaton* aut, const acgpu_input* in, acgpu_match* out, size_t cap,
                                size_t* n_out, acgpu_profile* prof) {
    if (!n_out) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = 0;
    if (prof) std::memset(prof, 0, sizeof *prof);
    acgpu_status st = check_nonoverlapping(aut, in);
    if (st) return st;
    if (in->span_start > in->span_end) return ACGPU_OK;
    // Input::earliest changes what a leftmost automaton reports (every step of FindIter is try_find on the caller's
    // Input, automaton.rs:864-883, :1266): the occurrence-selection rule does not model it, so the reference loop runs
    const bool earliest_matters = in->earliest && aut->cfg.match_kind != ACGPU_MATCH_STANDARD;
    if (aut->cfg.engine != 1 && !earliest_matters && parallel_find_eligible(aut, in)) {
        const bool force_windows = std::getenv("ACGPU_FIND_ITER_WINDOWS") != nullptr;   // test knob (read per call)
        g_too_dense = false;
        st = force_windows ? ACGPU_ERR_NOMEM : nonoverlapping_parallel(aut, in, aut->cfg.match_kind, out, cap, n_out, prof);
        if (st == ACGPU_ERR_NOMEM && !g_too_dense)   // the occurrence stream of the whole span does not fit: windows
            st = nonoverlapping_windowed(aut, in, aut->cfg.match_kind, out, cap, n_out);
        if (st == ACGPU_ERR_NOMEM && g_too_dense)    // tens of occurrences per byte: the serial loop is cheaper
            st = serial_impl(aut, in, false, out, cap, n_out, prof);
        return st;
    }
    return serial_impl(aut, in, false, out, cap, n_out, prof);
}
acgpu_status acgpu_find_iter(acgpu_automaton* aut, const acgpu_input* in, acgpu_match* out, size_t cap,
                             size_t* n_out) {
    return acgpu_find_iter_ex(aut, in, out, cap, n_out, nullptr);
}

// Automaton::try_replace_all_bytes / try_replace_all (src/automaton.rs:433-550) for the whole haystack.
acgpu_status acgpu_replace_all(acgpu_automaton* aut, const acgpu_input* in, const uint8_t* const* replace_with,
                               const size_t* replace_lens, size_t n_replace, uint32_t flags, uint8_t* out, size_t cap,
                               size_t* out_len) {
    if (!aut || !in || !out_len || (n_replace && (!replace_with || !replace_lens))) return ACGPU_ERR_INVALID_ARGUMENT;
    *out_len = 0;
    if (n_replace != aut->nnfa.pattern_lens.size()) {  // the reference asserts (src/automaton.rs:442-447)
        g_last_error = "replace_all requires a replacement for every pattern in the automaton";
        return ACGPU_ERR_INVALID_ARGUMENT;
    }
    acgpu_status st = check_nonoverlapping(aut, in);
    if (st) return st;
    if (in->anchored || in->earliest || in->span_start != 0 || in->span_end != in->haystack_len) {
        g_last_error = "replace_all searches the whole haystack unanchored (Input::new(haystack))";
        return ACGPU_ERR_INVALID_ARGUMENT;
    }
    DeviceState* ds = nullptr;
    if ((st = get_device_state(aut, &ds))) return st;
    ScratchLease sc(ds);
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    const uint64_t n = in->haystack_len;

    const uint8_t* dhay = in->haystack;
    if (!in->haystack_on_device) {
        HIP_TRY(sc->rhay.ensure(n + 32));
        if (n) HIP_TRY(hipMemcpyAsync(sc->rhay.p, in->haystack, n, hipMemcpyHostToDevice, stream));
        dhay = sc->rhay.as<uint8_t>();
    }
    // 1. the non-overlapping matches, left on the device
    acgpu_input fin = *in;
    fin.haystack = dhay; fin.haystack_on_device = 1; fin.out_on_device = 1;
    size_t m = 0, mcap = std::max<size_t>(size_t(1) << 16, n / 4096);
    for (;;) {
        HIP_TRY(sc->rmatch.ensure(mcap * sizeof(acgpu_match)));
        st = acgpu_find_iter_ex(aut, &fin, sc->rmatch.as<acgpu_match>(), mcap, &m, nullptr);
        if (st == ACGPU_ERR_BUFFER_TOO_SMALL && m > mcap) { mcap = m; continue; }
        if (st) return st;
        break;
    }
    // 2. replacement strings: concatenated bytes + offsets
    std::vector<uint64_t> roff(n_replace + 1, 0);
    for (size_t i = 0; i < n_replace; i++) roff[i + 1] = roff[i] + replace_lens[i];
    std::vector<uint8_t> rbytes(size_t(roff[n_replace]) + 16, 0);
    for (size_t i = 0; i < n_replace; i++)
        if (replace_lens[i]) std::memcpy(rbytes.data() + roff[i], replace_with[i], replace_lens[i]);
    HIP_TRY(sc->roff.upload(roff));
    HIP_TRY(sc->rtab.upload(rbytes));
    // 3. segment lengths -> output offsets -> total length
    HIP_TRY(sc->rwork.ensure(replace_scratch_bytes(m)));
    HIP_TRY(sc->totals.ensure(2 * sizeof(uint64_t)));
    uint64_t* d_total = sc->totals.as<uint64_t>();
    HIP_TRY(launch_replace_measure(sc->rmatch.as<acgpu_match>(), m, dhay, n, sc->roff.as<uint64_t>(),
                                   (flags & ACGPU_REPLACE_UTF8_BOUNDARIES) != 0, sc->rwork.p, d_total, stream));
    uint64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *out_len = size_t(total);
    if (total > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (total == 0) return ACGPU_OK;
    if (!out) return ACGPU_ERR_INVALID_ARGUMENT;
    // 4. the copy, straight into the caller's device buffer when it is 16-byte aligned
    uint8_t* dst = out;
    const bool direct = in->out_on_device && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    if (!direct) { HIP_TRY(sc->rout.ensure(total + 16)); dst = sc->rout.as<uint8_t>(); }
    HIP_TRY(launch_replace_copy(sc->rmatch.as<acgpu_match>(), m, dhay, n, sc->rtab.as<uint8_t>(),
                                sc->roff.as<uint64_t>(), sc->rwork.p, d_total, dst, total, stream));
    if (!direct)
        HIP_TRY(hipMemcpyAsync(out, dst, total, in->out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return ACGPU_OK;
}

// ---- stream search: AhoCorasick::try_stream_find_iter, src/ahocorasick.rs:1677-1683 -> StreamChunkIter,
// src/automaton.rs:1036-1244.  The reference reports a match the moment a match state is entered and restarts from
// the start state, i.e. the Standard find_iter of the concatenated stream; here every fed chunk is searched by all
// CUs with the last max_pattern_len-1 bytes of the stream as warm-up, and the selection chain carries `pos`.
acgpu_status acgpu_stream_begin(acgpu_automaton* aut, acgpu_stream** out) {
    if (!aut || !out) return ACGPU_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD) return ACGPU_ERR_UNSUPPORTED_STREAM;      // :1067-1069
    if (aut->nnfa.min_pattern_len == 0 && !aut->nnfa.pattern_lens.empty()) return ACGPU_ERR_UNSUPPORTED_EMPTY;  // :1082-1084
    acgpu_status st = enforce_anchored_consistency(aut->cfg.start_kind, false);                // start_state(Anchored::No)
    if (st) return st;
    auto* s = new (std::nothrow) acgpu_stream();
    if (!s) return ACGPU_ERR_NOMEM;
    s->aut = aut;
    *out = s;
    return ACGPU_OK;
}

void acgpu_stream_end(acgpu_stream* s) { delete s; }

namespace {
acgpu_status stream_feed_once(acgpu_stream* s, const uint8_t* bytes, size_t len, int32_t bytes_on_device,
                              void* hip_stream, size_t* n_matches);
}

acgpu_status acgpu_stream_feed(acgpu_stream* s, const uint8_t* bytes, size_t len, int32_t bytes_on_device,
                               void* hip_stream, size_t* n_matches) {
    struct GuardOff { bool prev = g_dense_guard; GuardOff() { g_dense_guard = false; } ~GuardOff() { g_dense_guard = prev; } } guard_off;
    // a large HOST chunk: its pieces are fed one after the other (the same stream search) while a helper thread copies
    // the later ones to the device (the reference refills its roll buffer behind the search, src/util/buffer.rs:113-123)
    if (s && n_matches && bytes && !bytes_on_device && len >= 2 * host_piece_bytes() && !s->aut->nnfa.pattern_lens.empty()) {
        DeviceState* ds = nullptr;
        acgpu_status pst = get_device_state(s->aut, &ds);
        if (pst) return pst;
        HIP_TRY(s->stage.ensure(len + 64));
        const size_t piece = host_piece_bytes();
        HostPipe pipe;
        HIP_TRY(pipe.start(ds->device, s->stage.as<uint8_t>(), bytes, len, piece));
        std::vector<acgpu_match> acc;
        for (size_t k = 0; k < pipe.n_pieces; k++) {
            HIP_TRY(pipe.wait(k, static_cast<hipStream_t>(hip_stream)));
            const size_t off = k * piece, nb = std::min(piece, len - off);
            size_t nk = 0;
            if ((pst = acgpu_stream_feed(s, s->stage.as<uint8_t>() + off, nb, 1, hip_stream, &nk))) return pst;
            acc.insert(acc.end(), s->last.begin(), s->last.end());
        }
        s->last.swap(acc);
        *n_matches = s->last.size();
        return ACGPU_OK;
    }
    const bool force_split = len > (size_t(64) << 10) && std::getenv("ACGPU_STREAM_SPLIT") != nullptr;   // test knob
    acgpu_status st = force_split ? ACGPU_ERR_NOMEM : stream_feed_once(s, bytes, len, bytes_on_device, hip_stream, n_matches);
    if (st == ACGPU_ERR_NOMEM && len > (size_t(64) << 10)) {
        // the occurrence stream of this chunk does not fit in device memory: feeding it as two halves is the same
        // stream search (state is only advanced by a feed that succeeds)
        const size_t h = len / 2;
        size_t n1 = 0, n2 = 0;
        if ((st = acgpu_stream_feed(s, bytes, h, bytes_on_device, hip_stream, &n1))) return st;
        std::vector<acgpu_match> acc;
        acc.swap(s->last);
        if ((st = acgpu_stream_feed(s, bytes + h, len - h, bytes_on_device, hip_stream, &n2))) return st;
        acc.insert(acc.end(), s->last.begin(), s->last.end());
        s->last.swap(acc);
        *n_matches = s->last.size();
    }
    return st;
}

namespace {
acgpu_status stream_feed_once(acgpu_stream* s, const uint8_t* bytes, size_t len, int32_t bytes_on_device,
                              void* hip_stream, size_t* n_matches) {
    if (!s || !n_matches || (len && !bytes)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_matches = 0;
    s->last.clear();
    if (len == 0 || s->aut->nnfa.pattern_lens.empty()) { s->total += len; return ACGPU_OK; }
    acgpu_automaton* aut = s->aut;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const size_t halo = s->halo.size();
    const size_t local = halo + len;
    HIP_TRY(s->buf.ensure(local + 32));
    uint8_t* d = s->buf.as<uint8_t>();
    if (halo) HIP_TRY(hipMemcpyAsync(d, s->halo.data(), halo, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d + halo, bytes, len, bytes_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    const uint64_t base = s->total - halo;   // absolute offset of d[0]
    acgpu_input in{};
    in.haystack = d; in.haystack_len = local; in.span_start = 0; in.span_end = local;
    in.haystack_on_device = 1; in.stream = hip_stream;
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(aut, &ds);
    if (st) return st;
    uint64_t n_sel = 0;
    {
        ScratchLease sc(ds);
        const size_t pos0 = s->pos > base ? size_t(s->pos - base) : 0;
        if ((st = nonoverlapping_core(aut, ds, sc.s.get(), &in, halo, local, pos0, ACGPU_MATCH_STANDARD, &n_sel, nullptr)))
            return st;
        s->last.resize(size_t(n_sel));
        if (n_sel) {
            HIP_TRY(hipMemcpyAsync(s->last.data(), sc->sel.p, n_sel * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
    }
    for (auto& m : s->last) { m.start += base; m.end += base; }
    if (n_sel) s->pos = s->last.back().end;
    // keep the last max_pattern_len-1 bytes as the next chunk's warm-up
    const size_t want = aut->nnfa.max_pattern_len ? aut->nnfa.max_pattern_len - 1 : 0;
    const size_t keep = std::min(want, local);
    std::vector<uint8_t> nh(keep);
    if (keep) {
        HIP_TRY(hipMemcpyAsync(nh.data(), d + (local - keep), keep, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    s->halo.swap(nh);
    s->total += len;
    *n_matches = size_t(n_sel);
    return ACGPU_OK;
}
}  // namespace

acgpu_status acgpu_stream_matches(const acgpu_stream* s, acgpu_match* out, size_t cap, size_t* n_out) {
    if (!s || !n_out) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = s->last.size();
    if (s->last.size() > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (!s->last.empty()) {
        if (!out) return ACGPU_ERR_INVALID_ARGUMENT;
        std::memcpy(out, s->last.data(), s->last.size() * sizeof(acgpu_match));
    }
    return ACGPU_OK;
}

namespace {

// First match of an eligible unanchored search, in parallel.  The span is scanned in growing windows; window k yields
// every occurrence with end <= b_k (earlier windows were empty), the selection rule picks its first match m, and m
// is final once every occurrence that could beat it is visible: always for Standard (first record of the stream),
// for the leftmost kinds when m.start + L <= b_k (an unseen occurrence ends after b_k, hence starts after b_k - L);
// otherwise the window is extended to m.start + L once.
acgpu_status find_parallel(acgpu_automaton* aut, const acgpu_input* in, int32_t* found, acgpu_match* m) {
    acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(occ, &ds);
    if (st) return st;
    ScratchLease sc(ds);
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    const uint64_t L = occ->nnfa.max_pattern_len;
    const int rule = aut->cfg.match_kind;
    uint64_t a = in->span_start, w = uint64_t(16) << 20;
    while (a < in->span_end) {
        uint64_t b = std::min<uint64_t>(in->span_end, a + w);
        const uint64_t lo = a > in->span_start + L ? a - L : in->span_start;
        for (int attempt = 0; attempt < 2; attempt++) {
            uint64_t n_sel = 0;
            st = nonoverlapping_core(occ, ds, sc.s.get(), in, size_t(lo), size_t(b), in->span_start, rule, &n_sel, nullptr);
            if (st == ACGPU_ERR_NOMEM && !g_too_dense && b - a > (uint64_t(64) << 10)) {   // occurrence stream of the window too large
                for (DevBuf* buf : {&sc->result, &sc->sel, &sc->selwork, &sc->events, &sc->eswork}) buf->release();
                w = std::max<uint64_t>((b - a) / 16, uint64_t(64) << 10);
                b = std::min<uint64_t>(in->span_end, a + w);
                attempt = -1;
                continue;
            }
            if (st) return st;
            if (n_sel == 0) break;
            acgpu_match first{};
            HIP_TRY(hipMemcpyAsync(&first, sc->sel.p, sizeof first, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            const bool final_ = rule == ACGPU_MATCH_STANDARD || b == in->span_end || first.start + L <= b;
            if (final_ || attempt == 1) { *m = first; *found = 1; return ACGPU_OK; }
            b = std::min<uint64_t>(in->span_end, first.start + L);
        }
        a = b;
        if (w < (uint64_t(4) << 30)) w *= 4;
    }
    return ACGPU_OK;
}

// is_match (earliest = true, only the boolean is observable): any occurrence in the span, windows as above, count only
acgpu_status is_match_parallel(acgpu_automaton* aut, const acgpu_input* in, int32_t* is_match) {
    acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
    acgpu_input oin = *in;
    oin.anchored = 0; oin.earliest = 0; oin.out_on_device = 0;
    uint64_t a = in->span_start, w = uint64_t(16) << 20;
    while (a < in->span_end) {
        const uint64_t b = std::min<uint64_t>(in->span_end, a + w);
        size_t n = 0;
        const acgpu_status st = overlapping_impl(occ, &oin, size_t(a), size_t(b), nullptr, 0, &n, nullptr);
        if (st != ACGPU_OK && st != ACGPU_ERR_BUFFER_TOO_SMALL) return st;
        if (n) { *is_match = 1; return ACGPU_OK; }
        a = b;
        if (w < (uint64_t(4) << 30)) w *= 4;
    }
    return ACGPU_OK;
}

}  // namespace

acgpu_status acgpu_find(acgpu_automaton* aut, const acgpu_input* in, int32_t* found, acgpu_match* m) {
    if (!found || !m || !in) return ACGPU_ERR_INVALID_ARGUMENT;
    *found = 0;
    acgpu_status st = check_nonoverlapping(aut, in);
    if (st) return st;
    if (in->span_start > in->span_end) return ACGPU_OK;
    // Standard automata always report the earliest match (src/automaton.rs:1259-1275), so `earliest` only changes
    // the answer for the leftmost kinds; those run the reference loop on one lane, like every input the occurrence
    // rule does not cover (anchored searches, empty patterns).
    const bool earliest_matters = in->earliest && aut->cfg.match_kind != ACGPU_MATCH_STANDARD;
    if (aut->cfg.engine != 1 && !earliest_matters && parallel_find_eligible(aut, in)) {
        g_too_dense = false;
        st = find_parallel(aut, in, found, m);
        if (!(st == ACGPU_ERR_NOMEM && g_too_dense)) return st;
        *found = 0;   // tens of occurrences per byte: the reference loop on one lane is cheaper (below)
    }
    acgpu_input host_out = *in;
    host_out.out_on_device = 0;
    size_t n = 0;
    st = serial_impl(aut, &host_out, true, m, 1, &n, nullptr);
    if (st == ACGPU_OK) *found = n ? 1 : 0;
    return st;
}

acgpu_status acgpu_is_match(acgpu_automaton* aut, const acgpu_input* in, int32_t* is_match) {
    if (!is_match || !in) return ACGPU_ERR_INVALID_ARGUMENT;
    *is_match = 0;
    acgpu_status st = check_nonoverlapping(aut, in);
    if (st) return st;
    if (in->span_start > in->span_end) return ACGPU_OK;
    if (aut->cfg.engine != 1 && parallel_find_eligible(aut, in)) return is_match_parallel(aut, in, is_match);
    acgpu_input e = *in;
    e.earliest = 1; e.out_on_device = 0;
    acgpu_match m;
    size_t n = 0;
    st = serial_impl(aut, &e, true, &m, 1, &n, nullptr);
    if (st == ACGPU_OK) *is_match = n ? 1 : 0;
    return st;
}

void acgpu_get_tables(const acgpu_automaton* a, acgpu_tables* t) {
    std::memset(t, 0, sizeof *t);
    t->nnfa_states = a->nnfa.states();
    t->nnfa_max_match_id = a->nnfa.special.max_match_id;
    t->nnfa_start_unanchored_id = a->nnfa.special.start_unanchored_id;
    t->nnfa_start_anchored_id = a->nnfa.special.start_anchored_id;
    std::memcpy(t->byte_classes, a->nnfa.byte_classes, 256);
    t->alphabet_len = a->nnfa.alphabet_len();
    t->nnfa_fail = a->nnfa.fail.data();
    t->nnfa_depth = a->nnfa.depth.data();
    t->nnfa_match_off = a->nnfa.moff.data();
    t->nnfa_match_pid = a->nnfa.mpid.data();
    t->pattern_lens = a->nnfa.pattern_lens.data();
    if (a->kind == ACGPU_KIND_DFA) {
        t->dfa_trans = a->dfa.trans.data(); t->dfa_trans_len = a->dfa.trans.size();
        t->dfa_state_len = a->dfa.state_len; t->dfa_stride2 = a->dfa.stride2;
        t->dfa_max_match_id = a->dfa.special.max_match_id;
        t->dfa_start_unanchored_id = a->dfa.special.start_unanchored_id;
        t->dfa_start_anchored_id = a->dfa.special.start_anchored_id;
        t->dfa_match_off = a->dfa.moff.data(); t->dfa_match_pid = a->dfa.mpid.data();
        t->dfa_num_match_states = a->dfa.num_match_states;
    }
    if (a->kind == ACGPU_KIND_CONTIGUOUS_NFA) {
        t->cnfa_repr = a->cnfa.repr.data(); t->cnfa_repr_len = a->cnfa.repr.size();
        t->cnfa_max_match_id = a->cnfa.special.max_match_id;
        t->cnfa_start_unanchored_id = a->cnfa.special.start_unanchored_id;
        t->cnfa_start_anchored_id = a->cnfa.special.start_anchored_id;
    }
}

acgpu_status acgpu_stream_read(const uint8_t* src, size_t len, int32_t iters, float* ms_best, void* stream) {
    if (!src || !ms_best || iters < 1 || len < 16) return ACGPU_ERR_INVALID_ARGUMENT;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sink), 16));
    acgpu_status st = ACGPU_OK;
    auto body = [&]() -> acgpu_status {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        float best = 0;
        for (int it = 0; it <= iters; it++) {   // (iteration 0 warms up)
            HIP_TRY(hipEventRecord(e0, s));
            HIP_TRY(launch_stream_read(src, len, sink, s));
            HIP_TRY(hipEventRecord(e1, s));
            HIP_TRY(hipEventSynchronize(e1));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (it && (best == 0 || ms < best)) best = ms;
        }
        *ms_best = best;
        return ACGPU_OK;
    };
    st = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    return st;
}

acgpu_status acgpu_gen_haystack(uint8_t* dst, uint64_t offset, size_t len, uint64_t seed, uint32_t lo, uint32_t span,
                                void* stream) {
    if (span == 0 || (len && !dst)) return ACGPU_ERR_INVALID_ARGUMENT;
    HIP_TRY(launch_gen_haystack(dst, offset, len, seed, lo, span, static_cast<hipStream_t>(stream)));
    return ACGPU_OK;
}

}  // extern "C"

uint32_t acgpu_default_chunk(const acgpu_automaton* aut, size_t span_len) { return default_chunk(aut, span_len); }
void acgpu_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
