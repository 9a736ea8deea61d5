// C ABI of libacgpu.so, the overlapping search: the per-call context and the pipelines of the engines (prefix-filter events, count -> scan -> fill,
// the LDS walk's events), split sets, the routing between them (overlapping_impl), large host haystacks piece by piece.  See capi.cpp.
#include "capi_impl.hpp"

using namespace acgpu;
using namespace acgpu_capi;

namespace acgpu_capi {

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
        const bool emit = all_pairs && n_records > 0 && (c.dev_result ? n_records <= c.dense->max_records : (n_records <= c.cap && c.out));
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
    // (the order pass packs its prefixes: fewer than 2^32 records)
    if (n_events > cap_ev || (!all_pairs && n_records > 0xFFFFFFFFull)) { *outcome = PfOutcome::TooManyEvents; return ACGPU_OK; }
    sc->rank_hint = uint32_t(std::min<uint64_t>(n_events, kEvAllPairs));
    sc->rank_over = n_events > kEvAllPairs;
    *c.n_out = size_t(n_records);
    if (c.dev_result) sc->events_served = true;   // (both forms below leave the records of the events in this scratch)
    if (all_pairs) {
        ov_profile(c, ENG_PF, n_records, n_events);
        acgpu_status st = ov_events_ms(c, true);
        if (st) return st;
        *result = c.dev_result && n_records > c.dense->max_records ? ACGPU_ERR_BUFFER_TOO_SMALL : ov_result(c, n_records, dout);
        return ACGPU_OK;
    }
    c.ds->dense_hint.store(16, std::memory_order_relaxed);
    // bucket order pass over the events this scan recorded (event_order.hip; the counters were re-armed by k_ev_write, the
    // counts are still in the device totals)
    acgpu_match* dst = nullptr;
    if (c.to_caller) { if (c.out && n_records <= c.cap) dst = c.out; }
    else if (n_records > 0 && (c.dev_result || (n_records <= c.cap && c.out))) {
        if (c.dev_result && n_records > c.dense->max_records) {   // (counted, not materialised: see DenseRule)
            HIP_TRY(hipStreamSynchronize(stream));
            *result = ACGPU_ERR_BUFFER_TOO_SMALL; return ACGPU_OK;
        }
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
                                const TriEvents* tev) {
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
    if (c.dev_result && n_records > c.dense->max_records) return ACGPU_ERR_BUFFER_TOO_SMALL;
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
acgpu_status ensure_lw_events(Scratch* sc, const ScanGeom& g, hipStream_t stream, uint32_t* gen) {
    const LwEvSizes z = lw_events_sizes(g);
    HIP_TRY(sc->lwev.ensure(z.ev_bytes));
    HIP_TRY(sc->lwtn.ensure(z.task_n_bytes));
    if (!sc->lwovf.p) {
        HIP_TRY(sc->lwovf.ensure(64));
        HIP_TRY(hipMemsetAsync(sc->lwovf.p, 0, 64, stream));
        sc->lw_gen = 0;
    }
    if (++sc->lw_gen == 0) ++sc->lw_gen;   // (never 0: the word starts out zeroed)
    *gen = sc->lw_gen;
    return ACGPU_OK;
}

// the chunk fill behind an overflowed event form: chunks of four lane-chunks (2 KiB: a wavefront per 512 bytes spends its time
// on warm-ups and prefix sums -- 1.5 ms against 1.2 for 92 M records), record offsets from the lane-chunk scan
ScanGeom lw_fill_geom(const ScanGeom& g) {
    ScanGeom f = g;
    f.chunk = 4 * g.chunk;
    f.n_chunks = (g.n_chunks + 3) / 4;
    return f;
}

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
    uint32_t gen = 0;
    if (acgpu_status st = ensure_lw_events(sc, g, stream, &gen)) return st;
    uint32_t* ovf = sc->lwovf.as<uint32_t>();
    // count walk (+ events, + the tasks' record counts) -> ONE workgroup scans the tasks and hands the totals and the overflow
    // word to the host (page-locked) -> emit.  The lane-chunk scan of kernels.hip (three launches) only runs in front of the chunk fill.
    HIP_TRY(sc->ensure_pinned());
    uint32_t* task_n = sc->lwtn.as<uint32_t>();
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[0], stream));
    HIP_TRY(launch_lw_count_ev(ds->hot, g, ss.counts, sc->lwev.p, task_n, ovf, gen, stream));
    if (c.prof) HIP_TRY(hipEventRecord(sc->ev[1], stream));
    HIP_TRY(launch_lw_task_scan(g, task_n, ss.totals, sc->pinned, ovf, stream));
    const bool legs = c.prof && !c.dev_result;
    const bool queued = c.to_caller && c.cap > 0 && c.out;
    bool fill_queued = false;
    auto chunk_fill = [&](uint64_t fcap, uint64_t max_waves, acgpu_match* dst, const uint32_t* gate) -> acgpu_status {
        HIP_TRY(launch_scan(ss, g.n_chunks, stream));   // (record offsets of the lane-chunks)
        HIP_TRY(launch_lw_fill(ds->hot, lw_fill_geom(g), nullptr, ss.totals, fcap, max_waves, nullptr, dst, stream, gate, gen, ss.offsets, 4, g.n_chunks));
        return ACGPU_OK;
    };
    if (queued) {   // device-resident output: queued behind the scan, sizes read on the device
        HIP_TRY(launch_lw_ev_emit(ds->hot, g, sc->lwev.p, task_n, ovf, gen, ss.counts, ss.totals, c.cap, c.out, stream));
        // ... and the chunk fill, gated on the overflow word, while recent calls of this automaton overflowed (launches that
        // return at once otherwise; without the hint an overflow costs a host round trip before the fill)
        if (ds->lw_dense_hint.load() > 0) {
            if (acgpu_status st = chunk_fill(c.cap, 16384, c.out, ovf)) return st;
            fill_queued = true;
        }
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
        if (overflow && c.out && !fill_queued) {   // a slab overflowed: nothing was written, the chunk fill does it now
            if (acgpu_status st = chunk_fill(c.cap, (g.n_chunks + 3) / 4, c.out, nullptr)) return st;
            HIP_TRY(hipStreamSynchronize(stream));
        }
        if (overflow) ds->lw_dense_hint.store(8); else if (fill_queued) ds->lw_dense_hint.fetch_sub(1);
        return ACGPU_OK;
    }
    if (!c.out && !c.dev_result) return ACGPU_ERR_INVALID_ARGUMENT;
    if (c.dev_result && n_records > c.dense->max_records) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (c.dev_result && c.dense->too_dense(n_records, c.span_bytes)) { c.dense->hit = true; return ACGPU_ERR_NOMEM; }
    HIP_TRY(sc->result.ensure(n_records * sizeof(acgpu_match)));
    acgpu_match* dout = sc->result.as<acgpu_match>();
    if (overflow) { if (acgpu_status st = chunk_fill(n_records, (g.n_chunks + 3) / 4, dout, nullptr)) return st; }
    else HIP_TRY(launch_lw_ev_emit(ds->hot, g, sc->lwev.p, task_n, ovf, gen, ss.counts, ss.totals, n_records, dout, stream));
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
    // count only: nothing is materialised (the plain count of each part; BUFFER_TOO_SMALL is how a count without a buffer returns)
    auto count_part = [&](int k) -> acgpu_status {
        acgpu_input cin = oin;
        cin.out_on_device = 0;
        const acgpu_status cs = overlapping_impl(aut->part[k].get(), &cin, shard_begin, shard_end, nullptr, 0, &cnt[k], prof ? &pp[k] : nullptr, sc[k]);
        return cs == ACGPU_ERR_BUFFER_TOO_SMALL ? ACGPU_OK : cs;
    };
    bool too_small = false;
    if (!dev_result && cap == 0) {   // a count was asked for: both parts count only
        for (int k = 0; k < 2; k++) if ((st = count_part(k))) return st;
        too_small = true;
    } else {
        for (int k = 0; k < 2; k++) {
            // the records the caller has room for bound what a part materialises: beyond that the part is counted
            // (round 5 built both streams whatever their size: a 1-byte straggler over multi-GiB text took tens of GB)
            if (!dev_result) rule->max_records = k == 0 ? uint64_t(cap) : uint64_t(cap) - cnt[0];
            st = too_small ? count_part(k)
                           : overlapping_impl(aut->part[k].get(), &oin, shard_begin, shard_end, nullptr, 0, &cnt[k], prof ? &pp[k] : nullptr, sc[k], &rec[k], rule);
            if (st == ACGPU_ERR_BUFFER_TOO_SMALL && !dev_result) { too_small = true; st = ACGPU_OK; }
            if (st) return st;   // (ACGPU_ERR_NOMEM with rule->hit: too dense -- the caller's alternative)
        }
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
    if (total > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;   // (always so when a part was only counted)
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
        HIP_TRY(sc->ensure_pinned());
        EnqueueSync sync;
        sync.host_totals = sc->pinned;
        sync.seq = ++sc->fin_seq;
        if ((st = enqueue_impl(aut, in, shard_begin, shard_end, out, cap, tot, 64, 0, &probed, nullptr, &sync))) return st;
        if (sync.done) {
            // (fused order chain: its last kernel stored the totals in the page-locked words; the launch behind it, which
            // re-zeroes the bucket words for the next call, is not waited for)
            if (probed) {   // (what the device-side probe of THIS call decided: one more word, and the whole stream to wait for)
                HIP_TRY(hipMemcpyAsync(sc->pinned + 4, actx->sc.probe.as<uint8_t>() + 64, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
                HIP_TRY(hipStreamSynchronize(c.stream));
                if (!was_sticky) {   // four probes in a row for the large-set filter: the next 32 searches skip the probe
                    if ((sc->pinned[4] & 0xFFFFFFFFull) != 0) {
                        if (ds->probe_away_run.fetch_add(1, std::memory_order_relaxed) + 1 >= 4) {
                            ds->probe_away_run.store(0, std::memory_order_relaxed);
                            ds->probe_skip.store(32, std::memory_order_relaxed);
                        }
                    } else ds->probe_away_run.store(0, std::memory_order_relaxed);
                }
            } else {
                // the chain's last kernel stores the call's sequence number behind the totals: polled here (the event behind
                // that kernel costs the stream ~6 us more before the host hears of it); the event is the fallback
                const volatile uint64_t* seen = sc->pinned + 3;
                bool done = false;
                for (uint32_t spins = 0; spins < (1u << 22) && !(done = *seen == sync.seq); spins++) __builtin_ia32_pause();
                if (done) std::atomic_thread_fence(std::memory_order_acquire);
                else HIP_TRY(hipEventSynchronize(sync.done));
            }
            const uint64_t t0 = sc->pinned[0], t1 = sc->pinned[1], n_ev = sc->pinned[2];
            if (t1 == 0 && t0 <= cap) {
                if (t0 > 0) ds->dense_hint.store(16, std::memory_order_relaxed);
                *n_out = size_t(t0);
                ov_profile(c, ENG_PF, t0, n_ev);
                if (prof) {
                    float ms = 0;
                    if (actx->ev[128] && actx->ev[129] && hipEventElapsedTime(&ms, actx->ev[128], actx->ev[129]) == hipSuccess) { prof->ms_scan = ms; prof->ms_total = ms; }
                    else (void)hipGetLastError();
                }
                return ACGPU_OK;
            }
            HIP_TRY(hipStreamSynchronize(c.stream));   // not delivered: the regular path below
        } else {
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
        if (const uint32_t lane_chunk = lw_events_chunk(ds->hot, uint32_t(halo), c.span_bytes)) return lw_event_pipeline(c, lane_chunk);
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

}  // extern "C"
