// C ABI of libacgpu.so, the enqueue-only overlapping search (acgpu_find_overlapping_enqueue*): everything queued on the caller's stream, sizes read
// on the device, no host round trip.  See capi.cpp.
#include "capi_impl.hpp"

using namespace acgpu;
using namespace acgpu_capi;

// slot 64 is the library's own (the synchronous call that borrows a stream's enqueue context times itself there, leaving
// the caller's slots 0..63 alone); *probed = whether THIS call queued the device-side probe
acgpu_status acgpu_capi::enqueue_impl(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end, acgpu_match* out,
                                      size_t cap, uint64_t* totals, int32_t slot, uint32_t flags, bool* probed, EnqueueGuess* guess,
                                      EnqueueSync* sync) {
    if (probed) *probed = false;
    if (sync) sync->done = nullptr;
    if (!aut || !totals || slot > 64) return ACGPU_ERR_INVALID_ARGUMENT;
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
        return enqueue_impl(aut->occ.get(), in, shard_begin, shard_end, out, cap, totals, slot, flags, probed, guess, sync);
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
    if (aut->part[0]) {
        // a split pattern set has no enqueue-only form (two pipelines and a merge sized from their counts): the call reports
        // "more occurrences than this form delivers", on which callers repeat with the synchronous call (acgpu.h) -- behind the
        // argument checks above, with the slot's events recorded like any other call
        HIP_TRY(hipMemsetAsync(totals, 0, sizeof(uint64_t), stream));
        HIP_TRY(hipMemsetAsync(totals + 1, 0xFF, sizeof(uint64_t), stream));
        if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ev[2 * slot + 1], stream));
        return ACGPU_OK;
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
        // Fused order chain (event_order.hip) while the automaton's recent results were dense: the scan does the histogram, the
        // order pass serves ANY number of events, its last kernel reports the totals and re-arms the counters -- no
        // all-pairs kernels, no copy -- and the bucket words are re-zeroed BEHIND it (natural text, 1 GiB: 13 launches behind
        // the scan -> 4, 118 us -> ~50).  With the device-side probe both gated scans carry the histogram: one of them runs.
        if (ds->var.eo_fused && out && cap && (guess || ds->dense_hint.load(std::memory_order_relaxed) > 0)) {
            if (!guess) ds->dense_hint.fetch_sub(1, std::memory_order_relaxed);
            // (bounds: a guessed result size -- find_iter's occurrence stream, capi_find.cpp -- or what the event list holds)
            const uint64_t o_events = guess ? std::min<uint64_t>(cap_ev, std::max<uint64_t>(guess->max_events, kEvCap + 1)) : cap_ev;
            const uint64_t max_rec = guess ? uint64_t(cap) : std::min<uint64_t>({uint64_t(cap), uint64_t(1) << 26, 4 * cap_ev});
            if ((st = ensure_order_work(sc, event_order_work_bytes(o_events, max_rec, span_bytes), stream))) return st;
            const size_t zb = event_order_zero_bytes(o_events, max_rec, span_bytes);
            if (sc->eo_zero_p != sc->eswork.p || sc->eo_zero_bytes < zb) HIP_TRY(hipMemsetAsync(sc->eswork.p, 0, zb, stream));
            sc->eo_zero_p = nullptr;
            route.hist = event_order_hist(o_events, max_rec, shard_begin, span_bytes, sc->eswork.p);
            if ((st = pf_route_prepare(sc, ds->hot, span_bytes, &route))) return st;
            HIP_TRY(launch_pf_any(ds->hot, g, nullptr, stream, sc->events.p, ctr, cap_ev, route));
            if (probe) {
                PfRoute other;
                other.force_pfx = true; other.gate = route.gate; other.gate_val = 1; other.hist = route.hist;
                if ((st = pf_route_prepare(sc, ds->hot, span_bytes, &other))) return st;
                HIP_TRY(launch_pf_any(ds->hot, g, nullptr, stream, sc->events.p, ctr, cap_ev, other));
            }
            if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ev[2 * slot + 1], stream));
            EoFused fz;
            fz.ctr = ctr; fz.totals = totals; fz.host_totals = sync ? sync->host_totals : nullptr; fz.seq = sync ? sync->seq : 0;
            HIP_TRY(launch_event_order_emit(ds->hot, ds->da, sc->events.p, nullptr, 0, o_events, max_rec, shard_begin, span_bytes, sc->eswork.p, out,
                                            stream, nullptr, true, &fz));
            sc->ev_armed = true;   // (the chain's last kernel zeroes the counters; the rank words were not touched)
            if (guess) {   // (the caller queues its own kernels first, then the re-zeroing)
                guess->served_events = o_events; guess->rearm_p = sc->eswork.p; guess->rearm_bytes = zb;
                return ACGPU_OK;
            }
            if (sync && sync->host_totals) {
                if (!ctx->fin) HIP_TRY(hipEventCreateWithFlags(&ctx->fin, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(ctx->fin, stream));
                sync->done = ctx->fin;
            }
            HIP_TRY(launch_event_order_zero(sc->eswork.p, zb, stream));
            sc->eo_zero_p = sc->eswork.p; sc->eo_zero_bytes = zb;
            return ACGPU_OK;
        }
        sc->eo_zero_p = nullptr;   // (the chains below use the order pass's scratch their own way)
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
        if (const uint32_t lane_chunk = lw_events_chunk(ds->hot, uint32_t(halo), g.emit_hi - g.emit_lo)) {
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
            uint32_t gen = 0;
            if ((st = ensure_lw_events(sc, eg, stream, &gen))) return st;
            uint32_t* ovf = sc->lwovf.as<uint32_t>();
            uint32_t* task_n = sc->lwtn.as<uint32_t>();
            HIP_TRY(launch_lw_count_ev(ds->hot, eg, es.counts, sc->lwev.p, task_n, ovf, gen, stream));
            if (slot >= 0) HIP_TRY(hipEventRecord(ctx->ev[2 * slot + 1], stream));
            HIP_TRY(launch_lw_task_scan(eg, task_n, es.totals, nullptr, nullptr, stream));
            if (cap > 0 && out) {
                // the chunk fill (behind the lane-chunk scan it takes its offsets from: four launches, gated on the overflow word) is
                // queued while the automaton's recent results overflowed their slabs; otherwise an overflow is REPORTED
                // (totals[1] = UINT64_MAX: the caller repeats with the synchronous call, which remembers)
                const bool with_fill = ds->lw_dense_hint.load() > 0 || !ds->adaptive || (flags & ACGPU_ENQUEUE_CLASSIC) != 0;   // (classic: dense results asked for)
                HIP_TRY(launch_lw_ev_emit(ds->hot, eg, sc->lwev.p, task_n, ovf, gen, es.counts, es.totals, cap, out, stream, totals, !with_fill));
                if (with_fill) {
                    HIP_TRY(launch_scan(es, eg.n_chunks, stream));
                    ScanGeom fg = eg;   // (the chunk fill takes four lane-chunks at a time: capi_overlap.cpp, lw_fill_geom)
                    fg.chunk = 4 * eg.chunk; fg.n_chunks = (eg.n_chunks + 3) / 4;
                    HIP_TRY(launch_lw_fill(ds->hot, fg, nullptr, es.totals, cap, 16384, nullptr, out, stream, ovf, gen, es.offsets, 4, eg.n_chunks));
                }
                return ACGPU_OK;
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

}  // extern "C"
