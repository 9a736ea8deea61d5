// C ABI of libacgpu.so (include/acgpu.h): host orchestration of the device pipeline -- automaton construction and
// upload, and the overlapping search (capi_find.cpp: find_iter / find / is_match; capi_stream.cpp: replace_all and the
// stream search; test_hooks.cpp: the test hooks, not part of this library).
//
// No CPU search path exists here by design: every search entry point launches HIP
// kernels and fails with ACGPU_ERR_NO_DEVICE / ACGPU_ERR_HIP when that is impossible.
#include "capi_impl.hpp"

using namespace acgpu;
using namespace acgpu_capi;

namespace acgpu_capi {

thread_local std::string g_last_error;

acgpu_status hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();   // reset the thread's last-error slot: a later launch check must not see this failure
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) return ACGPU_ERR_NO_DEVICE;
    return e == hipErrorOutOfMemory ? ACGPU_ERR_NOMEM : ACGPU_ERR_HIP;
}

}  // namespace acgpu_capi

acgpu_automaton::acgpu_automaton() = default;
acgpu_automaton::~acgpu_automaton() = default;   // (here DeviceState is complete)

namespace acgpu_capi {


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
    DenseRule* dense = nullptr;           // ... and its density rule (never null in internal mode)
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
    // (internal mode -- the caller continues on the stream -- times the scan only: the other event records are not taken,
    // each is a barrier packet between two short launches)
    if (c.dev_result) { c.prof->ms_compact = 0; c.prof->ms_fill = 0; c.prof->ms_total = ms; return ACGPU_OK; }
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
// scratch of the bucket order pass (event_order.hip)
acgpu_status ensure_order_work(Scratch* sc, size_t bytes, hipStream_t) {
    HIP_TRY(sc->eswork.ensure(bytes));
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
    const bool legs = c.prof && !c.dev_result;   // the legs behind the scan are timed (ov_events_ms)
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
    // (the last scan on this scratch had more events than the all-pairs rank takes: its workgroups would all return at once --
    // a small grid; the kernel is grid-stride, so a wrong guess costs time on one call only)
    HIP_TRY(launch_pf_event_rank(sc->events.p, ctr, kEvAllPairs, rank, c.ss.totals, sc->rank_over ? 0u : sc->rank_hint, stream));
    if (legs) HIP_TRY(hipEventRecord(sc->ev[2], stream));
    if (c.to_caller) {   // device-resident output: the scatter is enqueued without a host round trip
        if (legs) HIP_TRY(hipEventRecord(sc->ev[3], stream));
        HIP_TRY(launch_pf_event_write(c.ds->hot, c.ds->da, sc->events.p, ctr, kEvAllPairs, rank, c.ss.totals, c.out ? c.cap : 0, c.out, stream));
        if (legs) HIP_TRY(hipEventRecord(sc->ev[4], stream));
        sc->ev_armed = true;
    }
    HIP_TRY(sc->ensure_pinned());
    HIP_TRY(hipMemcpyAsync(sc->pinned, c.ss.totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const uint64_t n_records = sc->pinned[0], n_events = sc->pinned[1];
    const bool abandoned = n_events == ~uint64_t(0);
    const bool all_pairs = n_events <= kEvAllPairs;
    acgpu_match* dout = nullptr;
    bool order_zeroed = false;
    if (!c.to_caller) {   // host / scratch output: size the buffer first, then scatter (always launched: it re-arms)
        const bool emit = all_pairs && n_records > 0 && (c.dev_result || (n_records <= c.cap && c.out));
        if (emit) { HIP_TRY(sc->result.ensure(n_records * sizeof(acgpu_match))); dout = sc->result.as<acgpu_match>(); }
        // the order pass that may follow wants its bucket counters zeroed: k_ev_write takes a small region along (one launch less)
        void* zero_p = nullptr;
        size_t zero_bytes = 0;
        if (!all_pairs && !abandoned && n_events <= cap_ev && n_records > 0) {
            zero_bytes = event_order_zero_bytes(n_events, n_records, c.span_bytes);
            if (zero_bytes <= (size_t(1) << 20)) {
                if (acgpu_status st = ensure_order_work(sc, event_order_work_bytes(n_events, n_records, c.span_bytes), stream)) return st;
                zero_p = sc->eswork.p; order_zeroed = true;
            } else zero_bytes = 0;
        }
        if (legs) HIP_TRY(hipEventRecord(sc->ev[3], stream));
        HIP_TRY(launch_pf_event_write(c.ds->hot, c.ds->da, sc->events.p, ctr, kEvAllPairs, rank, c.ss.totals, emit ? n_records : 0, dout, stream,
                                      zero_p, zero_bytes));
        if (legs) HIP_TRY(hipEventRecord(sc->ev[4], stream));
        sc->ev_armed = true;
        if (emit && !c.dev_result)
            HIP_TRY(hipMemcpyAsync(c.out, dout, n_records * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
        // (more events than the all-pairs rank orders: the order pass below follows on the same stream and nothing on the
        // host depends on this launch -- no round trip here; config 5's find_iter paid three per call, now two)
        if (all_pairs || abandoned) HIP_TRY(hipStreamSynchronize(stream));
    }
    if (abandoned) { *outcome = PfOutcome::Abandoned; return ACGPU_OK; }
    if (n_events > cap_ev) { *outcome = PfOutcome::TooManyEvents; return ACGPU_OK; }
    sc->rank_hint = uint32_t(std::min<uint64_t>(n_events, kEvAllPairs));
    sc->rank_over = n_events > kEvAllPairs;
    *c.n_out = size_t(n_records);
    if (c.dev_result) sc->events_served = true;   // (both forms below leave the records of the events in this scratch)
    if (all_pairs) {
        ov_profile(c, ENG_PF, n_records, n_events);
        acgpu_status st = ov_events_ms(c, true);
        if (st) return st;
        *result = ov_result(c, n_records, dout);
        return ACGPU_OK;
    }
    c.ds->dense_hint.store(16, std::memory_order_relaxed);
    // bucket order pass over the events this scan recorded (event_order.hip; the counters were re-armed by k_ev_write, the
    // counts are still in the device totals)
    acgpu_match* dst = nullptr;
    if (c.to_caller) { if (c.out && n_records <= c.cap) dst = c.out; }
    else if (n_records > 0 && (c.dev_result || (n_records <= c.cap && c.out))) {
        if (c.dev_result && c.dense->too_dense(n_records, c.span_bytes)) {
            // (k_ev_write, which re-arms this scratch's event counters, may still be in flight: the scratch goes back to the
            // pool when the caller gives up on this path, and another thread's scan must not start on half-armed counters)
            HIP_TRY(hipStreamSynchronize(stream));
            c.dense->hit = true; *result = ACGPU_ERR_NOMEM; return ACGPU_OK;
        }
        HIP_TRY(sc->result.ensure(n_records * sizeof(acgpu_match)));
        dst = sc->result.as<acgpu_match>();
    }
    // (the selection kernels of the parallel find_iter read the record count from the device totals: still there)
    if (legs) HIP_TRY(hipEventRecord(sc->ev[3], stream));
    if (dst && n_events) {
        if (acgpu_status st = ensure_order_work(sc, event_order_work_bytes(n_events, n_records, c.span_bytes), stream)) return st;
        HIP_TRY(launch_event_order_emit(c.ds->hot, c.ds->da, sc->events.p, c.ss.totals, kEvAllPairs, n_events, n_records,
                                        c.shard_begin, c.span_bytes, sc->eswork.p, dst, stream, nullptr, order_zeroed));
    }
    if (legs) HIP_TRY(hipEventRecord(sc->ev[4], stream));
    if (dst && !c.to_caller && !c.dev_result)
        HIP_TRY(hipMemcpyAsync(c.out, dst, n_records * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
    ov_profile(c, ENG_PF, n_records, n_events);
    if (c.dev_result) {
        // internal mode (find_iter's occurrence stream): the caller continues on this stream -- selection kernels, then its
        // own synchronisation -- so the order pass is not waited for here (one host round trip less per find_iter); only
        // the scan's time, complete since the counts were read, is reported
        if (c.prof) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, sc->ev[0], sc->ev[1]));
            c.prof->ms_scan = ms; c.prof->ms_total = ms;
        }
        *result = ov_result(c, n_records, dst);
        return ACGPU_OK;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    acgpu_status st = ov_events_ms(c, false);
    if (st) return st;
    *result = ov_result(c, n_records, dst);
    return ACGPU_OK;
}

// the transition-walk count kernel of `eng` (global tables; the contiguous NFA through its LDS-assisted form when available)
// does the contiguous-NFA walk of `ds` run the shallow-skip kernel (cnfa_tri.hip)?
bool cnfa_tri_selected(const DeviceState* ds) {
    // variants: walk_literal = the reference loop verbatim, walk_tri = 0: the LDS-row walk (cnfa_walk.hip)
    return ds->cnfa_tri.ready && !ds->var.walk_literal && ds->var.walk_tri;
}
// ... and does the DFA walk run its shallow-skip kernel (dfa_tri.hip)?
bool dfa_tri_selected(const DeviceState* ds) {
    return ds->dfa_tri.ready && ds->var.walk_tri;   // (variant walk_tri = 0: the global-table walk of kernels.hip)
}
bool tri_walk_selected(uint32_t eng, const DeviceState* ds) {
    return (eng == ENG_CNFA && cnfa_tri_selected(ds)) || (eng == ENG_DFA && dfa_tri_selected(ds));
}
// Event buffer for a scan by that kernel (zeroed counters enqueued on `stream`): the count pass then records every
// match state it enters, and k_cnfa_tri_emit writes the ordered records without walking the haystack again.
acgpu_status cnfa_tri_events(const DeviceState* ds, Scratch* sc, const ScanGeom& g, uint64_t span_bytes, hipStream_t stream, TriEvents* ev) {
    *ev = TriEvents();
    if (!ds->var.tri_events || g.n_chunks >= 0xFFFFFFFFull) return ACGPU_OK;   // (variant tri_events = 0: count -> scan -> re-walking fill)
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
    const bool literal = ds->var.walk_literal != 0;
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
        if (acgpu_status st = cnfa_tri_events(ds, sc, g, c.span_bytes, stream, &tev)) return st;
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
    // small automata: the fill whose walk AND match lists live in LDS (an explicitly requested transition walk keeps its own)
    const bool lw_fill = fill_eng == ENG_DFA && aut->cfg.engine != 1 && lw_fill_supported(ds->hot);
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
        if (lw_fill) return launch_lw_fill(ds->hot, g, c.ss.active, c.ss.totals, fcap, max_waves, c.ss.aoff, dst, stream);
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
    if (c.dev_result && c.dense->too_dense(n_records, c.span_bytes)) { c.dense->hit = true; return ACGPU_ERR_NOMEM; }
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

// LDS walk, one row per state, event form (device/lds_emit.hip): the count walk notes every dword that gained a record as
// a 16-byte event in the slab of its task, the scan runs over LANE-chunks (512 bytes: a lane knows its own rank), and the
// records come from the events with every lane busy -- no second walk over the haystack.  A task with more events than its
// slab holds (more than one per 16 haystack bytes: the call is bound by its record writes then) leaves the fill to k_lw_fill.
acgpu_status lw_event_pipeline(OvCtx& c, uint32_t lane_chunk) {
    Scratch* sc = c.sc;
    hipStream_t stream = c.stream;
    DeviceState* ds = c.ds;
    ScanGeom g = c.g;
    g.chunk = lane_chunk;
    g.grid0 = (g.emit_lo / g.chunk) * g.chunk;
    g.n_chunks = std::max<uint64_t>(1, (g.emit_hi - g.grid0 + g.chunk - 1) / g.chunk);
    const uint64_t nb = (g.n_chunks + 255) / 256;
    HIP_TRY(sc->counts.ensure(g.n_chunks * sizeof(uint32_t)));
    HIP_TRY(sc->offsets.ensure(g.n_chunks * sizeof(uint64_t)));
    HIP_TRY(sc->active.ensure(g.n_chunks * sizeof(uint64_t)));
    HIP_TRY(sc->aoff.ensure(g.n_chunks * sizeof(uint64_t)));
    HIP_TRY(sc->bsum.ensure(nb * sizeof(uint64_t)));
    HIP_TRY(sc->bact.ensure(nb * sizeof(uint32_t)));
    ScanScratch ss = c.ss;
    ss.counts = sc->counts.as<uint32_t>(); ss.offsets = sc->offsets.as<uint64_t>(); ss.active = sc->active.as<uint64_t>();
    ss.aoff = sc->aoff.as<uint64_t>(); ss.bsum = sc->bsum.as<uint64_t>(); ss.bact = sc->bact.as<uint32_t>();
    const LwEvSizes z = lw_events_sizes(g);
    HIP_TRY(sc->lwev.ensure(z.ev_bytes));
    HIP_TRY(sc->lwtn.ensure(z.task_n_bytes));
    if (!sc->lwovf.p) {
        HIP_TRY(sc->lwovf.ensure(64));
        HIP_TRY(hipMemsetAsync(sc->lwovf.p, 0, 64, stream));
        sc->lw_gen = 0;
    }
    uint32_t* ovf = sc->lwovf.as<uint32_t>();
    const uint32_t gen = ++sc->lw_gen ? sc->lw_gen : ++sc->lw_gen;   // (never 0: the word starts out zeroed)
    // (the totals and the overflow word reach the host from the scan's own kernel: no copy launches)
    HIP_TRY(sc->ensure_pinned());
    ss.host_totals = sc->pinned; ss.extra32 = ovf;
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[0], stream));
    HIP_TRY(launch_lw_count_ev(ds->hot, g, ss.counts, sc->lwev.p, sc->lwtn.as<uint32_t>(), ovf, gen, stream));
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[1], stream));
    HIP_TRY(launch_scan(ss, g.n_chunks, stream));
    const bool legs = c.prof && !c.dev_result;
    const bool queued = c.to_caller && c.cap > 0 && c.out;
    if (queued) {   // device-resident output: queued behind the scan, sizes read on the device
        HIP_TRY(launch_lw_ev_emit(ds->hot, g, sc->lwev.p, sc->lwtn.as<uint32_t>(), ovf, gen, ss.offsets, ss.totals, c.cap, c.out, stream));
        if (legs) HIP_TRY(hipEventRecord(sc->ev[4], stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    const uint64_t n_records = sc->pinned[0], n_active = sc->pinned[1];
    const bool overflow = uint32_t(sc->pinned[2]) == gen;
    *c.n_out = size_t(n_records);
    c.g = g;
    ov_profile(c, ENG_HOT, n_records, n_active);
    if (c.prof) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev[0], sc->ev[1])); c.prof->ms_scan = ms; c.prof->ms_total = ms;
        if (legs && queued) {   // (scan + emit: one event record less between two short launches)
            HIP_TRY(hipEventElapsedTime(&ms, sc->ev[1], sc->ev[4])); c.prof->ms_fill = ms;
            HIP_TRY(hipEventElapsedTime(&ms, sc->ev[0], sc->ev[4])); c.prof->ms_total = ms;
        }
    }
    if (c.dev_result) *c.dev_result = nullptr;
    if (!c.dev_result && n_records > c.cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (n_records == 0) return ACGPU_OK;
    if (c.to_caller) {
        if (overflow && c.out) {   // a slab overflowed: nothing was written, the chunk fill does it now
            HIP_TRY(launch_lw_fill(ds->hot, g, ss.active, ss.totals, c.cap, n_active, ss.aoff, c.out, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        return ACGPU_OK;
    }
    if (!c.out && !c.dev_result) return ACGPU_ERR_INVALID_ARGUMENT;
    if (c.dev_result && c.dense->too_dense(n_records, c.span_bytes)) { c.dense->hit = true; return ACGPU_ERR_NOMEM; }
    HIP_TRY(sc->result.ensure(n_records * sizeof(acgpu_match)));
    acgpu_match* dout = sc->result.as<acgpu_match>();
    if (overflow) HIP_TRY(launch_lw_fill(ds->hot, g, ss.active, ss.totals, n_records, n_active, ss.aoff, dout, stream));
    else HIP_TRY(launch_lw_ev_emit(ds->hot, g, sc->lwev.p, sc->lwtn.as<uint32_t>(), ovf, gen, ss.offsets, ss.totals, n_records, dout, stream));
    if (c.dev_result) { *c.dev_result = dout; return ACGPU_OK; }   // (the caller continues on this stream)
    HIP_TRY(hipMemcpyAsync(c.out, dout, n_records * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return ACGPU_OK;
}

// The engine a search may be handed to when the prefix filter abandons it (PfArgs::route_*), and the cost-model
// coefficients that go with it.  Only the automatic engine choice routes; an explicitly requested engine is kept.
EngineFacts engine_facts(const acgpu_automaton* aut, const DeviceState* ds) {
    EngineFacts f;
    f.has_dfa = ds->da.has_dfa; f.pf_ready = ds->hot.pf_ready; f.lw_ready = ds->hot.lw_ready; f.pfx_ready = ds->hot.pfx_ready;
    f.lw_full = ds->var.lw_first != 0 && lw_fill_supported(ds->hot);
    f.min_pattern_len = aut->nnfa.min_pattern_len; f.want = aut->cfg.engine; f.routing = ds->var.routing != 0;
    return f;
}
static_assert(kPlanDfaWalk == ENG_DFA && kPlanCnfaWalk == ENG_CNFA && kPlanLdsWalk == ENG_HOT && kPlanPrefixFilter == ENG_PF &&
              kPlanLargeSetFilter == ENG_PF_LARGE && kPlanProbeMinSpan == kProbeMinSpan, "host/engine_plan.hpp mirrors these");
uint32_t pf_alternative(const acgpu_automaton* aut, const DeviceState* ds, PfRoute* route) {
    *route = PfRoute();
    EngineFacts f = engine_facts(aut, ds);
    f.pf_ready = true;   // (asked by the prefix-filter paths only)
    const uint32_t alt = plan_engines(f).alternative;
    // the cost-model coefficients that go with the alternative (large-set filter: its level 3 is a second, throughput-oriented
    // pass, so inputs that drown the two-type filter's inline level 3 -- natural text against a dictionary -- cost it far less)
    if (alt == ENG_HOT) *route = pf_route_to_lds_walk(ds->hot);
    else if (alt == ENG_PF_LARGE) *route = kPfRouteToLargeSet();
    else if (alt == ENG_DFA) *route = kPfRouteToDfaWalk();
    return alt;
}


// Overlapping search of a split pattern set (acgpu_automaton::part): both parts in internal mode (ordered records left in
// their scratch), then ONE merge into the destination -- the caller's device buffer, the caller's scratch (internal mode of
// find_iter / replace_all / the stream search), or a staging buffer that is copied to the host.
acgpu_status overlapping_split(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end, acgpu_match* out,
                               size_t cap, size_t* n_out, acgpu_profile* prof, Scratch* ext, acgpu_match** dev_result, DenseRule* dense) {
    DeviceState* ds[2] = {nullptr, nullptr};
    acgpu_status st;
    for (int k = 0; k < 2; k++) if ((st = get_device_state(aut->part[k].get(), &ds[k]))) return st;
    ScratchLease l0(ds[0]), l1(ds[1]);
    Scratch* sc[2] = {l0.s.get(), l1.s.get()};
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    acgpu_input oin = *in;
    oin.out_on_device = 0;
    if (!in->haystack_on_device) {   // one copy of a host haystack for both parts
        const size_t halo = aut->nnfa.max_pattern_len > 0 ? aut->nnfa.max_pattern_len - 1 : 0;
        const size_t need_lo = std::max(in->span_start, shard_begin >= halo ? shard_begin - halo : size_t(0));
        const uint8_t* dhay = nullptr;
        if ((st = device_haystack(in, need_lo, shard_end, sc[0], stream, &dhay))) return st;
        oin.haystack = dhay; oin.haystack_on_device = 1;
    }
    DenseRule own_rule;
    DenseRule* rule = dense ? dense : &own_rule;
    if (!dev_result) rule->guard = false;   // (a caller that wants the records gets them, however many)
    size_t cnt[2] = {0, 0};
    acgpu_match* rec[2] = {nullptr, nullptr};
    acgpu_profile pp[2];
    for (int k = 0; k < 2; k++) {
        st = overlapping_impl(aut->part[k].get(), &oin, shard_begin, shard_end, nullptr, 0, &cnt[k], prof ? &pp[k] : nullptr, sc[k], &rec[k], rule);
        if (st) return st;   // (ACGPU_ERR_NOMEM with rule->hit: too dense -- the caller's alternative)
    }
    const uint64_t total = uint64_t(cnt[0]) + cnt[1];
    *n_out = size_t(total);
    if (prof) {
        *prof = pp[0];
        prof->ms_scan += pp[1].ms_scan; prof->ms_compact += pp[1].ms_compact; prof->ms_fill += pp[1].ms_fill; prof->ms_total += pp[1].ms_total;
        prof->n_matches = total; prof->routed |= pp[1].routed;
    }
    if (dev_result) {
        *dev_result = nullptr;
        if (!ext) return ACGPU_ERR_INVALID_ARGUMENT;   // (internal mode leaves the records in the CALLER's scratch)
        if (total && rule->too_dense(total, shard_end - shard_begin)) { rule->hit = true; return ACGPU_ERR_NOMEM; }
        Scratch* dst = ext;
        HIP_TRY(dst->totals.ensure(2 * sizeof(uint64_t)));
        if (total) HIP_TRY(dst->result.ensure(total * sizeof(acgpu_match)));
        acgpu_match* merged = total ? dst->result.as<acgpu_match>() : nullptr;
        HIP_TRY(launch_merge_records(rec[0], rec[1], cnt[0], cnt[1], merged, dst->totals.as<uint64_t>(), stream));
        HIP_TRY(hipStreamSynchronize(stream));   // (the parts' scratch goes back to its pool when this returns)
        *dev_result = merged;
        return ACGPU_OK;
    }
    if (total > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (total == 0) return ACGPU_OK;
    if (!out) return ACGPU_ERR_INVALID_ARGUMENT;
    if (in->out_on_device) {
        HIP_TRY(launch_merge_records(rec[0], rec[1], cnt[0], cnt[1], out, nullptr, stream));
    } else {
        HIP_TRY(sc[0]->sel.ensure(total * sizeof(acgpu_match)));
        HIP_TRY(launch_merge_records(rec[0], rec[1], cnt[0], cnt[1], sc[0]->sel.as<acgpu_match>(), nullptr, stream));
        HIP_TRY(hipMemcpyAsync(out, sc[0]->sel.p, total * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return ACGPU_OK;
}

// `ext` / `dev_result`: internal mode used by the parallel find_iter -- run on the caller's scratch and leave the
// ordered records in scratch->result (returned through *dev_result) instead of copying them anywhere.
acgpu_status overlapping_impl(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end,
                              acgpu_match* out, size_t cap, size_t* n_out, acgpu_profile* prof,
                              Scratch* ext, acgpu_match** dev_result, DenseRule* dense) {
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
        return overlapping_impl(aut->occ.get(), in, shard_begin, shard_end, out, cap, n_out, prof, ext, dev_result, dense);

    if (aut->part[0]) return overlapping_split(aut, in, shard_begin, shard_end, out, cap, n_out, prof, ext, dev_result, dense);

    DeviceState* ds = nullptr;
    if ((st = get_device_state(aut, &ds))) return st;
    std::unique_ptr<ScratchLease> lease;
    if (!ext) lease = std::make_unique<ScratchLease>(ds);
    OvCtx c;
    c.aut = aut; c.ds = ds; c.sc = ext ? ext : lease->s.get(); c.in = in;
    c.sc->events_served = false;
    c.stream = static_cast<hipStream_t>(in->stream);
    c.shard_begin = shard_begin; c.shard_end = shard_end; c.span_bytes = shard_end - shard_begin;
    c.out = out; c.cap = cap; c.n_out = n_out; c.prof = prof; c.dev_result = dev_result;
    DenseRule default_rule;   // internal-mode callers that pass none get the plain rule and nobody reads `hit`
    c.dense = dense ? dense : &default_rule;
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
    const int want = aut->cfg.engine;
    uint32_t eng = plan_engines(engine_facts(aut, ds)).first;   // (host/engine_plan.hpp)
    if (eng == 0) {
        g_last_error = "requested engine is unavailable for this automaton";
        return ACGPU_ERR_INVALID_ARGUMENT;
    }

    const bool no_events = ds->var.pf_classic != 0;   // variant: chunk counters + scan + fill
    // Occurrence-dense input under the large-set filter (dictionary/english/sorted.txt's 121 111 words of four bytes and more
    // over prose: 0.17 occurrences per byte): every occurrence costs the filter a level-3 walk and an event or a counter
    // atomic -- 31 GB/s for its scan, 3.6 GB/s for the whole call (scripts/split_probe.py) -- while the transition walk
    // counts at its usual rate.  Once a scan has overflowed the event list the automaton is remembered as dense and the
    // walk runs directly, until a result comes back sparse.
    const bool walk_ok = want == 0 && ds->da.has_dfa && tri_walk_selected(ENG_DFA, ds);
    if (eng == ENG_PF && walk_ok && ds->walk_hint.load(std::memory_order_relaxed) > 0) {
        st = classic_pipeline(c, ENG_DFA);
        if (st == ACGPU_OK || st == ACGPU_ERR_BUFFER_TOO_SMALL) {
            if (*n_out * 256 > c.span_bytes) ds->walk_hint.store(8, std::memory_order_relaxed);
            else ds->walk_hint.fetch_sub(1, std::memory_order_relaxed);
        }
        return st;
    }
    // Device-resident haystack and output, results dense lately: the enqueue-only machinery (probe or sticky choice, gated
    // filters, all-pairs rank AND the bucket order pass, all queued without a host decision) followed by ONE synchronisation,
    // instead of scan -> read the counts -> order pass -> synchronise (natural text, 1 GiB: 0.86 -> 0.79 ms per call).  A call
    // it does not deliver (abandoned scan, more events than the list holds, buffer too small for the order pass) falls
    // through to the regular path.
    if (eng == ENG_PF && aut->nnfa.max_pattern_len <= 0xFFFF && !no_events && c.to_caller && !ext && want == 0 && in->haystack_on_device &&
        out && cap > 0 && ds->dense_hint.load(std::memory_order_relaxed) > 0) {
      // (the stream's enqueue context is borrowed for the call; another host thread searching on the same stream at this
      // moment keeps the regular path and its pooled scratch)
      DeviceState::AsyncCtx* actx = ds->async_ctx(c.stream);
      std::unique_lock<std::mutex> borrowed(actx->busy, std::try_to_lock);
      if (borrowed.owns_lock()) {
        uint64_t* tot = c.ss.totals;
        const bool was_sticky = ds->probe_skip.load(std::memory_order_relaxed) > 0;
        bool probed = false;
        if ((st = enqueue_impl(aut, in, shard_begin, shard_end, out, cap, tot, 64, 0, &probed))) return st;
        HIP_TRY(sc->ensure_pinned());
        HIP_TRY(hipMemcpyAsync(sc->pinned, tot, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
        sc->pinned[2] = 0;
        if (probed)   // what the device-side probe of THIS call decided (a call without a probe leaves an older word there)
            HIP_TRY(hipMemcpyAsync(sc->pinned + 2, actx->sc.probe.as<uint8_t>() + 64, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        if (!was_sticky && probed) {   // four probes in a row for the large-set filter: the next 32 searches skip the probe
            if ((sc->pinned[2] & 0xFFFFFFFFull) != 0) {
                if (ds->probe_away_run.fetch_add(1, std::memory_order_relaxed) + 1 >= 4) {
                    ds->probe_away_run.store(0, std::memory_order_relaxed);
                    ds->probe_skip.store(32, std::memory_order_relaxed);
                }
            } else ds->probe_away_run.store(0, std::memory_order_relaxed);
        }
        const uint64_t t0 = sc->pinned[0], t1 = sc->pinned[1];
        if (t1 <= ACGPU_ENQUEUE_MAX_EVENTS && t0 <= cap) {   // delivered
            if (t1 == 0 && t0 > 0) ds->dense_hint.store(16, std::memory_order_relaxed);   // ... by the order pass: still dense
            *n_out = size_t(t0);
            ov_profile(c, ENG_PF, t0, t1);
            if (prof) {
                float ms = 0;
                if (actx->ev[128] && actx->ev[129] && hipEventElapsedTime(&ms, actx->ev[128], actx->ev[129]) == hipSuccess) { prof->ms_scan = ms; prof->ms_total = ms; }
                else (void)hipGetLastError();
            }
            return ACGPU_OK;
        }
      }
    }
    if (eng == ENG_PF && aut->nnfa.max_pattern_len <= 0xFFFF && !no_events) {
        PfRoute route;
        const uint32_t alt = pf_alternative(aut, ds, &route);
        PfOutcome outcome = PfOutcome::Done;
        acgpu_status result;
        bool probed_away = false;
        EnginePlan plan;
        plan.first = ENG_PF; plan.alternative = alt;
        const PfStart how = plan_pf_start(plan, ds->probe_skip.load(std::memory_order_relaxed), ds->route_hint.load(std::memory_order_relaxed),
                                          c.span_bytes, pf_uses_large_set(ds->hot, route));
        if (how == PfStart::TakeAlternative) {
            // the last four probes in a row chose the alternative (the large-set filter, or a transition walk: the reference's
            // match-dense small-set definitions call after call): the next 32 searches take it unasked -- the probe and its
            // host round trip were a third of a 256 MiB call
            ds->probe_skip.fetch_sub(1, std::memory_order_relaxed);
            probed_away = true;
        } else if (how == PfStart::Probe) {
            // recent scans of this automaton were abandoned: ask the probe first (256 samples of 8 KB through the filter)
            if ((st = ensure_probe(sc, c.stream))) return st;
            uint32_t* flag = reinterpret_cast<uint32_t*>(sc->probe.as<uint8_t>() + 64);
            HIP_TRY(launch_pf_probe(ds->hot, c.g, route, flag, sc->probe.as<unsigned long long>(), c.stream));
            HIP_TRY(sc->ensure_pinned());
            HIP_TRY(hipMemcpyAsync(sc->pinned, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
            HIP_TRY(hipStreamSynchronize(c.stream));
            probed_away = (sc->pinned[0] & 0xFFFFFFFFull) != 0;
            if (probed_away) {
                ds->route_hint.store(8, std::memory_order_relaxed);
                if (ds->probe_away_run.fetch_add(1, std::memory_order_relaxed) + 1 >= 4) {
                    ds->probe_away_run.store(0, std::memory_order_relaxed);
                    ds->probe_skip.store(32, std::memory_order_relaxed);
                }
            } else {
                ds->route_hint.fetch_sub(1, std::memory_order_relaxed);
                ds->probe_away_run.store(0, std::memory_order_relaxed);
            }
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
        // TooManyEvents: the chunk-counter form of the same filter below -- or, for the large-set filter, the walk
        PfRoute ran;
        ran.force_pfx = c.force_large_set;
        if (outcome == PfOutcome::TooManyEvents && walk_ok && pf_uses_large_set(ds->hot, ran)) {
            ds->walk_hint.store(8, std::memory_order_relaxed);
            c.routed = 1;
            eng = ENG_DFA;
        }
    }
    if (eng == ENG_HOT && ds->var.lw_events && aut->nnfa.min_pattern_len >= 1 && c.span_bytes < (uint64_t(15) << 30))
        if (const uint32_t lane_chunk = lw_events_chunk(ds->hot, uint32_t(halo))) return lw_event_pipeline(c, lane_chunk);
    return classic_pipeline(c, eng);
}

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

}  // namespace acgpu_capi

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

}  // extern "C"

namespace acgpu_capi {
// Split rule (acgpu_automaton::part): a dictionary of long patterns with a few short stragglers.  On natural text the
// large-set filter probes 8-byte prefixes at every other position when every pattern has nine bytes (0.61 ms per GiB of
// prose for the reference's words-5000), but ONE shorter word shortens the key for all of them -- 8 bytes 0.75, 7 0.86,
// 6 1.19, 5 1.97, 4 3.5 ms, and a 3-byte word sends the search to the global transition walk at 8 ms
// (profiles/r05_minlen_sweep.jsonl).  From a shortest pattern of six bytes down, and while the short ones are few, the set
// is searched as two: the price is the second pass over the haystack (~0.3 ms per GiB), also where the unsplit filter
// would have been fine (random text: 1.4x slower than unsplit).
constexpr size_t kSplitShortBelow = 9, kSplitMinShortest = 6, kSplitMaxShort = 64, kSplitMinLong = 1000;

acgpu_status build_impl(const acgpu_config* cfg_in, const uint8_t* const* patterns, const size_t* lens, size_t n,
                        const uint32_t* ids, size_t id_space, bool allow_split, acgpu_automaton** out);
}  // namespace acgpu_capi

extern "C" {

acgpu_status acgpu_build(const acgpu_config* cfg_in, const uint8_t* const* patterns, const size_t* lens, size_t n,
                         acgpu_automaton** out) {
    return build_impl(cfg_in, patterns, lens, n, nullptr, 0, true, out);
}

}  // extern "C"

acgpu_status acgpu_capi::build_impl(const acgpu_config* cfg_in, const uint8_t* const* patterns, const size_t* lens, size_t n,
                                    const uint32_t* ids, size_t id_space, bool allow_split, acgpu_automaton** out) {
    if (!out) return ACGPU_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    acgpu_config cfg;
    if (cfg_in) cfg = *cfg_in; else acgpu_config_init(&cfg);
    if (n && (!patterns || !lens)) return ACGPU_ERR_INVALID_ARGUMENT;
    if (cfg.match_kind < 0 || cfg.match_kind > 2 || cfg.start_kind < 0 || cfg.start_kind > 2 || cfg.kind < 0 || cfg.kind > 3 ||
        cfg.engine < ACGPU_ENGINE_AUTO || cfg.engine > ACGPU_ENGINE_PREFIX_FILTER)
        return ACGPU_ERR_INVALID_ARGUMENT;
    std::unique_ptr<acgpu_automaton> a;
    try {
        a = std::make_unique<acgpu_automaton>();
        a->cfg = cfg;
        // acgpu_engine -> the request as the pipelines test it: 0 auto, 1 transition walk (either table), 2 LDS walk, 3 prefix filter
        static const int kWant[5] = {0, 1, 1, 2, 3};
        a->cfg.engine = kWant[cfg.engine];
        BuildOptions o;
        o.match_kind = cfg.match_kind;
        o.ascii_case_insensitive = cfg.ascii_case_insensitive != 0;
        o.byte_classes = cfg.byte_classes != 0;
        o.start_kind = cfg.start_kind;
        o.pattern_ids = ids; o.id_space = id_space;
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
            const bool try_dfa = cfg.start_kind != ACGPU_START_BOTH && a->nnfa.n_patterns <= 100;
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
        const bool twin_for_both = cfg.match_kind == ACGPU_MATCH_STANDARD && cfg.start_kind == ACGPU_START_BOTH && a->cfg.engine != 1;
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
            acgpu_status ost = build_impl(&oc, patterns, lens, n, ids, id_space, allow_split, &o);
            if (ost == ACGPU_OK) a->occ.reset(o);
        }
        // split sets (see kSplit* above): Standard / unanchored, automatic engine choice, no empty pattern
        if (allow_split && !ids && cfg.match_kind == ACGPU_MATCH_STANDARD && cfg.start_kind == ACGPU_START_UNANCHORED &&
            a->cfg.engine == 0 && n > kSplitMinLong && a->nnfa.min_pattern_len >= 1 && a->nnfa.min_pattern_len <= kSplitMinShortest) {
            std::vector<const uint8_t*> pp[2];
            std::vector<size_t> ll[2];
            std::vector<uint32_t> ii[2];
            for (size_t i = 0; i < n; i++) {
                const int k = lens[i] < kSplitShortBelow ? 1 : 0;
                pp[k].push_back(patterns[i]); ll[k].push_back(lens[i]); ii[k].push_back(uint32_t(i));
            }
            if (!pp[1].empty() && pp[1].size() <= kSplitMaxShort && pp[0].size() >= kSplitMinLong) {
                acgpu_config pc = cfg;
                pc.byte_classes = 1; pc.dense_depth_set = 0; pc.gpu_dfa_fill = 0;
                acgpu_automaton* parts[2] = {nullptr, nullptr};
                bool ok = true;
                for (int k = 0; k < 2 && ok; k++) {
                    // (state count unknown before the build: the full automaton's bounds both parts')
                    pc.kind = a->nnfa.states() <= (size_t(1) << 20) ? ACGPU_KIND_DFA : ACGPU_KIND_CONTIGUOUS_NFA;
                    ok = build_impl(&pc, pp[k].data(), ll[k].data(), pp[k].size(), ii[k].data(), n, false, &parts[k]) == ACGPU_OK;
                }
                if (ok) { a->part[0].reset(parts[0]); a->part[1].reset(parts[1]); }
                else { delete parts[0]; delete parts[1]; }
            }
        }
    } catch (const std::bad_alloc&) {
        return ACGPU_ERR_NOMEM;
    }
    *out = a.release();
    return ACGPU_OK;
}

extern "C" {

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

// Engine variant of this automaton (host/variants.hpp lists the names): explicit, per automaton, before its first upload.
acgpu_status acgpu_set_variant(acgpu_automaton* aut, const char* name, int32_t value) {
    if (!aut || !name) return ACGPU_ERR_INVALID_ARGUMENT;
    {
        std::lock_guard<std::mutex> lk(aut->mu);
        if (!aut->devs.empty()) { g_last_error = "acgpu_set_variant: the automaton is already on a device"; return ACGPU_ERR_INVALID_ARGUMENT; }
        int32_t* f = aut->var.field(name);
        if (!f) { g_last_error = std::string("acgpu_set_variant: unknown variant ") + name; return ACGPU_ERR_INVALID_ARGUMENT; }
        *f = value;
    }
    // the automata searched on this one's behalf (the Standard twin of a leftmost automaton, the parts of a split set)
    for (acgpu_automaton* child : {aut->occ.get(), aut->part[0].get(), aut->part[1].get()})
        if (child) if (acgpu_status st = acgpu_set_variant(child, name, value)) return st;
    return ACGPU_OK;
}

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
    ds->adaptive = aut->cfg.deterministic_routing == 0;
    ds->var = aut->var;
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
                hipError_t e = build_hot_tables(aut->nnfa, d, aut->var, ds->hot);
                if (e != hipSuccess) return hip_fail(e, "build_hot_tables");
            }
            if (aut->cfg.match_kind == ACGPU_MATCH_STANDARD && aut->cfg.start_kind == ACGPU_START_UNANCHORED) {
                // the walk engine of this table: the shallow-skip form of the transition walk (single-start layout)
                // (an accelerator, not a requirement: if its second copy of the table does not fit, the plain walk serves)
                hipError_t e = build_dfa_tri(aut->nnfa, d, ds->da.dfa.moff, ds->dfa_tri);
                if (e != hipSuccess) { (void)hipGetLastError(); ds->dfa_tri.ready = false; }
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
                // accelerators of the contiguous-NFA walk: a failed allocation leaves the literal walk (k_cnfa_count)
                if (build_cnfa_hot(aut->cnfa, ds->cnfa_hot) != hipSuccess) { (void)hipGetLastError(); ds->cnfa_hot.ready = false; }
                if (build_cnfa_tri(aut->cnfa, ds->cnfa_tri) != hipSuccess) { (void)hipGetLastError(); ds->cnfa_tri.ready = false; }
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

// slot 64 is the library's own (the synchronous call that borrows a stream's enqueue context times itself there, leaving
// the caller's slots 0..63 alone); *probed = whether THIS call queued the device-side probe
}  // extern "C"
acgpu_status acgpu_capi::enqueue_impl(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end, acgpu_match* out,
                                      size_t cap, uint64_t* totals, int32_t slot, uint32_t flags, bool* probed, EnqueueGuess* guess) {
    if (probed) *probed = false;
    if (!aut || !totals || slot > 64) return ACGPU_ERR_INVALID_ARGUMENT;
    if (aut->part[0] && in && in->haystack_on_device) {
        // a split pattern set has no enqueue-only form (two pipelines and a merge sized from their counts): the call reports
        // "more occurrences than this form delivers", on which callers repeat with the synchronous call (acgpu.h)
        hipStream_t s0 = static_cast<hipStream_t>(in->stream);
        HIP_TRY(hipMemsetAsync(totals, 0, sizeof(uint64_t), s0));
        HIP_TRY(hipMemsetAsync(totals + 1, 0xFF, sizeof(uint64_t), s0));
        return ACGPU_OK;
    }
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
        return enqueue_impl(aut->occ.get(), in, shard_begin, shard_end, out, cap, totals, slot, flags, probed, guess);
    DeviceState* ds = nullptr;
    if ((st = get_device_state(aut, &ds))) return st;
    if (!in->haystack_on_device || !(cap == 0 || out)) {
        g_last_error = "enqueue form: device haystack and device output required";
        return ACGPU_ERR_INVALID_ARGUMENT;
    }
    const size_t halo = aut->nnfa.max_pattern_len > 0 ? aut->nnfa.max_pattern_len - 1 : 0;
    if (halo > 0xFFFFFF00ull) return ACGPU_ERR_INVALID_ARGUMENT;
    // engine choice as in overlapping_impl (no routing target is needed here: see below)
    uint32_t eng = plan_engines(engine_facts(aut, ds)).first;   // engine choice as in overlapping_impl (host/engine_plan.hpp)
    if (eng == 0) {
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
        // (after four synchronous probes in a row that chose the large-set filter the next searches take it unasked,
        // here as in overlapping_impl)
        const bool sticky = alt == ENG_PF_LARGE && ds->probe_skip.load(std::memory_order_relaxed) > 0 && span_bytes >= kProbeMinSpan &&
                            !pf_uses_large_set(ds->hot, route);
        if (sticky) { ds->probe_skip.fetch_sub(1, std::memory_order_relaxed); route = PfRoute(); route.force_pfx = true; }
        const bool probe = !sticky && alt == ENG_PF_LARGE && ds->route_hint.load(std::memory_order_relaxed) > 0 && span_bytes >= kProbeMinSpan &&
                           !pf_uses_large_set(ds->hot, route);
        if (probe) {
            if (probed) *probed = true;
            if ((st = ensure_probe(sc, stream))) return st;
            uint32_t* flag = reinterpret_cast<uint32_t*>(sc->probe.as<uint8_t>() + 64);
            HIP_TRY(launch_pf_probe(ds->hot, g, route, flag, sc->probe.as<unsigned long long>(), stream));
            route.gate = flag; route.gate_val = 0;
            if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ev[2 * slot], stream));   // (the slot times the scan, not the probe in front of it)
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
        // (a guessed result size: the order pass below is certain to be queued -- its scratch is sized first, so that
        // k_ev_write can zero its counters on the way)
        uint64_t g_events = 0, g_records = 0;
        void* zero_p = nullptr;
        size_t zero_bytes = 0;
        if (guess && out && cap) {
            g_events = std::min<uint64_t>(cap_ev, std::max<uint64_t>(guess->max_events, kEvCap + 1));
            g_records = cap;
            if ((st = ensure_order_work(sc, event_order_work_bytes(g_events, g_records, span_bytes), stream))) return st;
            zero_bytes = event_order_zero_bytes(g_events, g_records, span_bytes);
            if (zero_bytes <= (size_t(1) << 20)) zero_p = sc->eswork.p; else zero_bytes = 0;
        }
        HIP_TRY(launch_pf_event_rank(sc->events.p, ctr, kEvCap, rank, totals, guess && guess->over_all_pairs ? 0u : kEvCap / 2, stream));   // (grid hint only: grid-stride kernel)
        HIP_TRY(launch_pf_event_write(ds->hot, ds->da, sc->events.p, ctr, kEvCap, rank, totals, out ? cap : 0, out, stream, zero_p, zero_bytes));
        sc->ev_armed = true;
        if (g_events) {
            HIP_TRY(launch_event_order_emit(ds->hot, ds->da, sc->events.p, totals, kEvCap, g_events, g_records, shard_begin, span_bytes,
                                            sc->eswork.p, out, stream, nullptr, zero_p != nullptr));
            guess->served_events = g_events;
            return ACGPU_OK;
        }
        // more occurrences than the all-pairs rank orders: while the automaton's recent (synchronous) results were dense,
        // the bucket order pass is queued too -- launches that return at once unless needed -- and resets totals[1] to 0
        // when it delivered; otherwise the caller sees totals[1] > ACGPU_ENQUEUE_MAX_EVENTS and repeats synchronously
        if (out && cap && ds->dense_hint.load(std::memory_order_relaxed) > 0) {
            ds->dense_hint.fetch_sub(1, std::memory_order_relaxed);
            // (records the order pass is sized for: what cap_ev events can plausibly stand for -- a handful of patterns per
            // event -- and never more than the caller has room for; if its scratch cannot be had the pass is simply not
            // queued: the scan and the all-pairs path above are already enqueued, the caller sees totals[1] > MAX and
            // repeats synchronously)
            const uint64_t max_rec = std::min<uint64_t>({uint64_t(cap), uint64_t(1) << 26, 4 * cap_ev});
            if (ensure_order_work(sc, event_order_work_bytes(cap_ev, max_rec, span_bytes), stream) == ACGPU_OK)
                HIP_TRY(launch_event_order_emit(ds->hot, ds->da, sc->events.p, totals, kEvCap, cap_ev, max_rec, shard_begin, span_bytes,
                                                sc->eswork.p, out, stream, totals));
            else (void)hipGetLastError();
        }
        return ACGPU_OK;
    }
    // the LDS walk of a small automaton: records from the events of its count walk (lds_emit.hip), the chunk fill gated on
    // their overflow word -- everything reads its sizes on the device
    if (eng == ENG_HOT && ds->var.lw_events && aut->nnfa.min_pattern_len >= 1 && g.emit_hi - g.emit_lo < (uint64_t(15) << 30)) {
        if (const uint32_t lane_chunk = lw_events_chunk(ds->hot, uint32_t(halo))) {
            ScanGeom eg = g;
            eg.chunk = lane_chunk;
            eg.grid0 = (eg.emit_lo / eg.chunk) * eg.chunk;
            eg.n_chunks = std::max<uint64_t>(1, (eg.emit_hi - eg.grid0 + eg.chunk - 1) / eg.chunk);
            const uint64_t enb = (eg.n_chunks + 255) / 256;
            HIP_TRY(sc->counts.ensure(eg.n_chunks * sizeof(uint32_t)));
            HIP_TRY(sc->offsets.ensure(eg.n_chunks * sizeof(uint64_t)));
            HIP_TRY(sc->active.ensure(eg.n_chunks * sizeof(uint64_t)));
            HIP_TRY(sc->aoff.ensure(eg.n_chunks * sizeof(uint64_t)));
            HIP_TRY(sc->bsum.ensure(enb * sizeof(uint64_t)));
            HIP_TRY(sc->bact.ensure(enb * sizeof(uint32_t)));
            HIP_TRY(sc->totals.ensure(2 * sizeof(uint64_t)));
            ScanScratch es;
            es.counts = sc->counts.as<uint32_t>(); es.offsets = sc->offsets.as<uint64_t>(); es.active = sc->active.as<uint64_t>();
            es.aoff = sc->aoff.as<uint64_t>(); es.bsum = sc->bsum.as<uint64_t>(); es.bact = sc->bact.as<uint32_t>();
            es.totals = sc->totals.as<uint64_t>();
            const LwEvSizes z = lw_events_sizes(eg);
            HIP_TRY(sc->lwev.ensure(z.ev_bytes));
            HIP_TRY(sc->lwtn.ensure(z.task_n_bytes));
            if (!sc->lwovf.p) {
                HIP_TRY(sc->lwovf.ensure(64));
                HIP_TRY(hipMemsetAsync(sc->lwovf.p, 0, 64, stream));
                sc->lw_gen = 0;
            }
            uint32_t* ovf = sc->lwovf.as<uint32_t>();
            const uint32_t gen = ++sc->lw_gen ? sc->lw_gen : ++sc->lw_gen;
            HIP_TRY(launch_lw_count_ev(ds->hot, eg, es.counts, sc->lwev.p, sc->lwtn.as<uint32_t>(), ovf, gen, stream));
            if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ev[2 * slot + 1], stream));
            HIP_TRY(launch_scan(es, eg.n_chunks, stream));
            if (cap > 0 && out) {
                HIP_TRY(launch_lw_ev_emit(ds->hot, eg, sc->lwev.p, sc->lwtn.as<uint32_t>(), ovf, gen, es.offsets, es.totals, cap, out, stream));
                HIP_TRY(launch_lw_fill(ds->hot, eg, es.active, es.totals, cap, 16384, es.aoff, out, stream, ovf, gen));
            }
            HIP_TRY(hipMemcpyAsync(totals, es.totals, sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemsetAsync(totals + 1, 0, sizeof(uint64_t), stream));
            return ACGPU_OK;
        }
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
        if ((st = cnfa_tri_events(ds, sc, g, g.emit_hi - g.emit_lo, stream, &tev))) return st;
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
        } else if (fill_eng == ENG_DFA && aut->cfg.engine != 1 && lw_fill_supported(ds->hot))
            HIP_TRY(launch_lw_fill(ds->hot, g, ss.active, ss.totals, cap, 16384, ss.aoff, out, stream));
        else if (fill_eng == ENG_DFA && aut->cfg.engine != 1 && hot_fill_supported(ds->hot, g))
            HIP_TRY(launch_hot_fill(ds->hot, ds->da, g, ss.active, ss.totals, cap, 16384, ss.aoff, out, stream));
        else
            HIP_TRY(launch_walk_fill(fill_eng, ds->da, g, ss.active, ss.totals, cap, 16384, ss.aoff, out, stream));
    }
    HIP_TRY(hipMemcpyAsync(totals, ss.totals, sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));   // records
    HIP_TRY(hipMemsetAsync(totals + 1, 0, sizeof(uint64_t), stream));                                  // no event list, no event limit
    return ACGPU_OK;
}

extern "C" {
acgpu_status acgpu_find_overlapping_enqueue_ex(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin,
                                               size_t shard_end, acgpu_match* out, size_t cap, uint64_t* totals,
                                               int32_t slot, uint32_t flags) {
    if (slot >= 64) return ACGPU_ERR_INVALID_ARGUMENT;
    return enqueue_impl(aut, in, shard_begin, shard_end, out, cap, totals, slot, flags, nullptr);
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

