// C ABI of libacgpu.so, non-overlapping searches: find_iter (parallel: occurrence stream + device selection; serial:
// the reference loops on one lane), find, is_match.  See capi.cpp.
#include "capi_impl.hpp"

using namespace acgpu;
using namespace acgpu_capi;

namespace acgpu_capi {

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

// One host round trip instead of two for an occurrence stream like the last one (config 5: 45 k occurrences in 8 GiB per
// call): scan, event order and selection are all queued, sized by a GUESS -- twice the last stream of this automaton
// (DeviceState::stream_hint) -- with every kernel reading the real counts on the device, and the host synchronises once.  A
// stream the guess does not hold (or a scan that abandoned itself) reports so in its totals, the selection kernels see an
// empty stream, *served stays false and the caller repeats the search the regular way; no guessing for the next 16
// searches then.  Results are those of the regular path: the same kernels in the same order.
constexpr uint64_t kGuessMaxStream = kSelectFewLimit / 2;   // streams guessed at: up to half of what the scan-less selection takes
acgpu_status nonoverlapping_guessed(acgpu_automaton* occ, DeviceState* ds, Scratch* sc, const acgpu_input* in,
                                    size_t shard_begin, size_t shard_end, size_t pos0, int rule_kind, uint64_t* n_sel,
                                    acgpu_profile* prof, acgpu_match* direct, size_t direct_cap, bool* went_direct, bool* served) {
    *served = false;
    const uint64_t last = uint64_t(ds->stream_hint.load());
    if (last == 0 || last > kGuessMaxStream || ds->stream_cool.load() > 0 || !in->haystack_on_device || occ->part[0] ||
        occ->cfg.engine != 0 || ds->var.pf_classic != 0 || occ->nnfa.max_pattern_len > 0xFFFF ||
        plan_engines(engine_facts(occ, ds)).first != ENG_PF)
        return ACGPU_OK;
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    // (the stream's enqueue context is borrowed, as in overlapping_impl; a second host thread on the same stream keeps the regular path)
    DeviceState::AsyncCtx* actx = ds->async_ctx(stream);
    std::unique_lock<std::mutex> borrowed(actx->busy, std::try_to_lock);
    if (!borrowed.owns_lock()) return ACGPU_OK;
    const uint64_t cap_rec = std::max<uint64_t>(2 * last, uint64_t(1) << 16);
    HIP_TRY(sc->result.ensure(cap_rec * sizeof(acgpu_match)));
    HIP_TRY(sc->totals.ensure(4 * sizeof(uint64_t)));
    uint64_t* d_tot = sc->totals.as<uint64_t>();   // [0] records, [1] events (0 once the order pass delivered), [2] selected
    acgpu_match* dS = sc->result.as<acgpu_match>();
    acgpu_input oin = *in;
    oin.anchored = 0; oin.earliest = 0; oin.out_on_device = 1;
    EnqueueGuess guess;
    guess.max_events = cap_rec;   // (an event stands for at least one record)
    guess.over_all_pairs = last > ACGPU_ENQUEUE_MAX_EVENTS;
    if (acgpu_status st = enqueue_impl(occ, &oin, shard_begin, shard_end, dS, cap_rec, d_tot, prof ? 64 : -1, 0, nullptr, &guess)) return st;
    if (guess.served_events == 0) {   // (the enqueue form took a branch without an order pass: let it drain, search the regular way)
        HIP_TRY(hipStreamSynchronize(stream));
        return ACGPU_OK;
    }
    const bool to_direct = direct && direct_cap >= cap_rec;
    if (!to_direct) HIP_TRY(sc->sel.ensure(cap_rec * sizeof(acgpu_match)));
    acgpu_match* sel_dst = to_direct ? direct : sc->sel.as<acgpu_match>();
    HIP_TRY(sc->selwork.ensure(select_scratch_bytes(cap_rec)));
    const uint64_t nblk = (cap_rec + 1023) / 1024;
    HIP_TRY(sc->counts.ensure(nblk * sizeof(uint32_t) + 16));
    ScanScratch ss;
    ss.counts = sc->counts.as<uint32_t>(); ss.totals = d_tot + 2;
    HIP_TRY(sc->ensure_pinned());
    SelectGate gate;
    gate.totals = d_tot; gate.max_events = guess.served_events; gate.max_records = cap_rec;
    gate.host = sc->pinned;   // (page-locked and device-visible: the last selection kernel reports there, no copy launch)
    sc->pinned[1] = ~uint64_t(0);
    HIP_TRY(launch_select_parallel(dS, cap_rec, d_tot, rule_kind, pos0, occ->nnfa.max_pattern_len, sc->selwork.p, ss, sel_dst,
                                   cap_rec, stream, gate));
    if (guess.rearm_bytes) {
        // fused order chain: its bucket words are re-zeroed behind the selection; the host waits for the selection only
        if (!actx->fin) HIP_TRY(hipEventCreateWithFlags(&actx->fin, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(actx->fin, stream));
        HIP_TRY(launch_event_order_zero(guess.rearm_p, guess.rearm_bytes, stream));
        actx->sc.eo_zero_p = guess.rearm_p; actx->sc.eo_zero_bytes = guess.rearm_bytes;
        HIP_TRY(hipEventSynchronize(actx->fin));
    } else HIP_TRY(hipStreamSynchronize(stream));
    const uint64_t records = sc->pinned[0], events = sc->pinned[1], selected = sc->pinned[2];
    if (events > guess.served_events || records > cap_rec) {   // not delivered: the regular path repeats the search
        ds->stream_hint.store(0);
        ds->stream_cool.store(16);
        return ACGPU_OK;
    }
    ds->stream_hint.store(records <= kGuessMaxStream ? int(records) : 0);
    *n_sel = selected;
    *served = true;
    if (went_direct) *went_direct = to_direct;
    if (prof) {
        std::memset(prof, 0, sizeof *prof);
        prof->engine_used = ENG_PF; prof->n_matches = selected;
        float ms = 0;
        if (actx->ev[128] && actx->ev[129] && hipEventElapsedTime(&ms, actx->ev[128], actx->ev[129]) == hipSuccess) { prof->ms_scan = ms; prof->ms_total = ms; }
        else (void)hipGetLastError();
    }
    return ACGPU_OK;
}

// Core of the parallel find_iter: occurrences whose end lies in (shard_begin, shard_end] (the whole span when the
// shard is the span), selection starting at position pos0.  The chosen records are left in sc->sel (device);
// *n_sel receives their number.
acgpu_status nonoverlapping_core(acgpu_automaton* occ, DeviceState* ds, Scratch* sc, const acgpu_input* in,
                                 size_t shard_begin, size_t shard_end, size_t pos0, int rule_kind, uint64_t* n_sel,
                                 acgpu_profile* prof, DenseRule* dense, acgpu_match* direct, size_t direct_cap, bool* went_direct) {
    *n_sel = 0;
    if (went_direct) *went_direct = false;
    {
        bool served = false;
        if (acgpu_status st = nonoverlapping_guessed(occ, ds, sc, in, shard_begin, shard_end, pos0, rule_kind, n_sel, prof, direct, direct_cap,
                                                     went_direct, &served))
            return st;
        if (served) return ACGPU_OK;
    }
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    acgpu_input oin = *in;
    oin.anchored = 0; oin.earliest = 0; oin.out_on_device = 0;
    size_t m_total = 0;
    acgpu_match* dS = nullptr;
    acgpu_status st = overlapping_impl(occ, &oin, shard_begin, shard_end, nullptr, 0, &m_total, prof, sc, &dS, dense);
    if (st) return st;
    // (what the next search of this automaton may guess at -- nonoverlapping_guessed; streams that did not come from the prefix
    // filter's events are not guessed at: that form queues the filter)
    if (ds->stream_cool.load() > 0) ds->stream_cool.fetch_sub(1);
    ds->stream_hint.store(sc->events_served && m_total > 0 && m_total <= kGuessMaxStream ? int(m_total) : 0);
    if (m_total == 0) return ACGPU_OK;
    // selection on the device (select.hip): succ pointers for all occurrences in parallel, block-wise orbit, ordered
    // compaction; the single-lane form only for streams beyond the u32 index range
    // the caller's device buffer takes the selection directly when it holds the whole stream (the selection is a subset):
    // no second copy, one synchronisation less
    const bool to_direct = direct && direct_cap >= m_total && m_total < 0xFFFFFFF0ull;
    if (!to_direct) HIP_TRY(sc->sel.ensure(m_total * sizeof(acgpu_match)));
    acgpu_match* sel_dst = to_direct ? direct : sc->sel.as<acgpu_match>();
    if (went_direct) *went_direct = to_direct;
    uint64_t* d_tot = sc->totals.as<uint64_t>();  // [0] = number of stream records (written by the scan)
    if (m_total < 0xFFFFFFF0ull) {
        HIP_TRY(sc->selwork.ensure(select_scratch_bytes(m_total)));
        HIP_TRY(sc->seltot.ensure(2 * sizeof(uint64_t)));
        const bool few = m_total <= kSelectFewLimit;   // (no scan launches: the stream length stays in d_tot until the last kernel)
        if (!few) HIP_TRY(hipMemcpyAsync(sc->seltot.p, d_tot, sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));  // n_in survives the scan below
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
        HIP_TRY(launch_select_parallel(dS, m_total, few ? d_tot : sc->seltot.as<uint64_t>(), rule_kind, pos0,
                                       occ->nnfa.max_pattern_len, sc->selwork.p, ss, sel_dst, m_total,
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
                                     size_t cap, size_t* n_out, acgpu_profile* prof, DenseRule* dense) {
    *n_out = 0;
    acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(occ, &ds);
    if (st) return st;
    ScratchLease sc(ds);
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    uint64_t n_sel = 0;
    bool direct = false;
    if ((st = nonoverlapping_core(occ, ds, sc.s.get(), in, in->span_start, in->span_end, in->span_start, rule_kind,
                                  &n_sel, prof, dense, in->out_on_device ? out : nullptr, in->out_on_device ? cap : 0, &direct)))
        return st;
    *n_out = size_t(n_sel);
    if (n_sel > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (n_sel == 0 || direct) return ACGPU_OK;
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
                                     size_t* n_out, DenseRule* dense) {
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
        st = nonoverlapping_core(occ, ds, sc.s.get(), in, size_t(pos), size_t(b), size_t(pos), rule, &n_sel, nullptr, dense);
        if (st == ACGPU_ERR_NOMEM && !dense->hit && w > w_min) { trim(); w = std::max<uint64_t>(w / 8, w_min); grow = false; continue; }
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

// Leftmost find_iter through the per-start candidate table (device/start_select.hip) -- for inputs on which the occurrence
// stream is dense.  Eligible: a leftmost match kind, the conditions of the parallel form (unanchored, no empty pattern),
// the trie tables of the prefix filters on the device, no pattern longer than a block of the table.
bool start_table_eligible(const acgpu_automaton* aut, const acgpu_input* in) {
    if (aut->cfg.match_kind == ACGPU_MATCH_STANDARD || !parallel_find_eligible(aut, in)) return false;
    if (aut->cfg.engine != ACGPU_ENGINE_AUTO) return false;   // (an explicitly requested engine is kept, or refused, by the occurrence scan)
    const acgpu_automaton* o = aut->occ ? aut->occ.get() : aut;
    return aut->var.start_table && o->nnfa.max_pattern_len >= 1 && o->nnfa.max_pattern_len <= kSsBlock;
}

bool start_table_servable(const DeviceState* ds) {
    const HotTables& h = ds->hot;
    return h.pf_ready && h.atab && h.own_pid && (ds->da.has_dfa || ds->da.has_cnfa);
}

acgpu_status nonoverlapping_start_table(acgpu_automaton* aut, const acgpu_input* in, int rule, acgpu_match* out, size_t cap,
                                        size_t* n_out, acgpu_profile* prof, bool* served) {
    *n_out = 0;
    *served = false;
    acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(occ, &ds);
    if (st) return st;
    const HotTables& h = ds->hot;
    if (!start_table_servable(ds)) return ACGPU_OK;   // (not served: the caller falls back)
    *served = true;
    ScratchLease sc(ds);
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    if (prof && (st = ensure_events(sc.s.get()))) return st;
    const uint8_t* dhay = nullptr;
    if ((st = device_haystack(in, in->span_start, in->span_end, sc.s.get(), stream, &dhay))) return st;
    SsTables t;
    t.atab = h.atab; t.acls = h.acls; t.own_pid = h.own_pid;
    t.plens = ds->da.has_dfa ? ds->da.dfa.plens : ds->da.cnfa.plens;
    t.ashift = h.ashift; t.root = h.start; t.L = uint32_t(occ->nnfa.max_pattern_len); t.n_states = h.n_states;
    uint64_t window = aut->var.ss_window_kib > 0 ? uint64_t(aut->var.ss_window_kib) << 10 : uint64_t(256) << 20;   // (variant: window size)
    window = std::max<uint64_t>(kSsBlock, window / kSsBlock * kSsBlock);
    const uint64_t span = in->span_end - in->span_start;
    const uint64_t max_win = std::min(window, std::max<uint64_t>(span, 1));
    const uint64_t nblk = (max_win + kSsBlock - 1) / kSsBlock, nb = (nblk + 255) / 256;
    HIP_TRY(sc->selwork.ensure(start_select_work_bytes(max_win, t.L)));
    HIP_TRY(sc->counts.ensure(nblk * sizeof(uint32_t) + 16));
    HIP_TRY(sc->offsets.ensure(nblk * sizeof(uint64_t)));
    HIP_TRY(sc->active.ensure(nblk * sizeof(uint64_t)));
    HIP_TRY(sc->aoff.ensure(nblk * sizeof(uint64_t)));
    HIP_TRY(sc->bsum.ensure((nb + 1) * sizeof(uint64_t)));
    HIP_TRY(sc->bact.ensure((nb + 1) * sizeof(uint32_t)));
    HIP_TRY(sc->totals.ensure(2 * sizeof(uint64_t)));
    HIP_TRY(sc->ensure_pinned());
    ScanScratch ss;
    ss.counts = sc->counts.as<uint32_t>(); ss.offsets = sc->offsets.as<uint64_t>(); ss.active = sc->active.as<uint64_t>();
    ss.aoff = sc->aoff.as<uint64_t>(); ss.bsum = sc->bsum.as<uint64_t>(); ss.bact = sc->bact.as<uint32_t>();
    ss.totals = sc->totals.as<uint64_t>();
    uint64_t total = 0;
    if (prof) HIP_TRY(hipEventRecord(sc->ev[0], stream));
    for (uint64_t lo = in->span_start; lo < in->span_end; lo += window) {
        const uint64_t n = std::min<uint64_t>(window, in->span_end - lo);
        HIP_TRY(launch_start_select(t, dhay, in->span_end, lo, n, rule == ACGPU_MATCH_LEFTMOST_LONGEST ? 1 : 0, sc->selwork.p,
                                    lo == in->span_start, ss, stream));
        HIP_TRY(hipMemcpyAsync(sc->pinned, ss.totals, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        const uint64_t m = sc->pinned[0];
        if (m && out && total + m <= cap) {
            if (in->out_on_device) {
                HIP_TRY(launch_start_select_emit(t, lo, n, sc->selwork.p, ss, total, cap, out, stream));
            } else {
                HIP_TRY(sc->sel.ensure(m * sizeof(acgpu_match)));
                HIP_TRY(launch_start_select_emit(t, lo, n, sc->selwork.p, ss, 0, m, sc->sel.as<acgpu_match>(), stream));
                HIP_TRY(hipMemcpyAsync(out + total, sc->sel.p, m * sizeof(acgpu_match), hipMemcpyDeviceToHost, stream));
            }
        }
        total += m;
    }
    if (prof) HIP_TRY(hipEventRecord(sc->ev[1], stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (prof) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev[0], sc->ev[1]));
        prof->ms_scan = ms; prof->ms_total = ms;
        prof->bytes_scanned = span; prof->n_matches = total; prof->engine_used = ENG_PF;
    }
    // dense lately?  (keeps the next calls on this path; a sparse result sends them back to the filters)
    ds->ss_hint.store(total > std::max<uint64_t>(uint64_t(1) << 12, span / 8) ? 8 : 0, std::memory_order_relaxed);
    *n_out = size_t(total);
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

}  // namespace acgpu_capi

extern "C" {

acgpu_status acgpu_find_iter_ex(acgpu_automaton* aut, const acgpu_input* in, acgpu_match* out, size_t cap,
                                size_t* n_out, acgpu_profile* prof) {
    if (!n_out) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = 0;
    if (prof) std::memset(prof, 0, sizeof *prof);
    acgpu_status st = check_nonoverlapping(aut, in);
    if (st) return st;
    if (in->span_start > in->span_end) return ACGPU_OK;
    // Input::earliest changes what a leftmost automaton reports (every step of FindIter is try_find on the caller's Input,
    // automaton.rs:864-883, :1266: the search returns at the FIRST match state it enters).  Up to that state the leftmost
    // automaton is the Standard one -- its construction only differs behind match states (noncontiguous.rs:1296-1346: failure
    // links of match states go to DEAD; leftmost-first leaves out patterns that have an earlier pattern as a prefix, which
    // end behind that pattern) -- and the pattern reported is the first own pattern along the failure chain in both: for
    // sets without an empty pattern, `earliest` on a leftmost automaton IS the Standard iteration of the same patterns
    // (tests/test_oracle_naive.py checks it on the oracle), i.e. the Standard rule over the twin's occurrence stream.
    // Round 5 ran the reference loop on one lane for it.
    const bool earliest_matters = in->earliest && aut->cfg.match_kind != ACGPU_MATCH_STANDARD;
    const int rule = earliest_matters ? int(ACGPU_MATCH_STANDARD) : aut->cfg.match_kind;
    if (aut->cfg.engine != 1 && parallel_find_eligible(aut, in)) {
        const bool force_windows = aut->var.find_iter_windows != 0;       // (variants: the forms tests force)
        const bool force_table = aut->var.find_iter_start_table != 0 && !earliest_matters;
        if (!force_windows && !force_table && aut->var.find_iter_disjoint) {
            // Pattern sets whose occurrences can neither overlap nor share an end (LwHostTables::disjoint -- one-byte sets: the
            // reference's memchr / jetscii / teddy1 definitions): the iteration of every match kind takes every occurrence, so
            // find_iter IS the overlapping search of the Standard twin -- nothing to select, nothing to materialise twice
            acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
            DeviceState* ods = nullptr;
            if ((st = get_device_state(occ, &ods))) return st;
            if (!occ->part[0] && ods->hot.lw_ready && ods->hot.lw.flavour == kLwFull && ods->hot.lw.disjoint &&
                plan_engines(engine_facts(occ, ods)).first == ENG_HOT) {
                acgpu_input oin = *in;
                oin.anchored = 0; oin.earliest = 0;
                return overlapping_impl(occ, &oin, in->span_start, in->span_end, out, cap, n_out, prof);
            }
        }
        bool table_ok = !force_windows && !earliest_matters && start_table_eligible(aut, in);   // (the table holds leftmost candidates)
        bool served = false;
        DeviceState* ds = nullptr;
        if (table_ok) {
            acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
            if ((st = get_device_state(occ, &ds))) return st;
            // the table must be servable on this device (trie tables of the prefix filters: absent for contiguous-NFA-only
            // uploads, beyond 131 072 patterns or 2^20 states) BEFORE the density threshold is lowered for its sake: otherwise
            // a moderately dense input would be declared "too dense" for the occurrence stream and fall to the one-lane loop
            table_ok = start_table_servable(ds);
        }
        if (table_ok) {   // recent calls met dense input (or the test knob): straight to the per-start table
            if (force_table || ds->ss_hint.load(std::memory_order_relaxed) > 0) {
                st = nonoverlapping_start_table(aut, in, rule, out, cap, n_out, prof, &served);
                if (served) return st;
            }
        }
        DenseRule dense;
        // (with the table at hand a stream of more than one occurrence per 4 bytes counts as dense: the table costs 3.9 ms per
        // 256 MiB whatever the input, the stream -- events of the LDS walk or of the filters, then the selection from its breaks
        // -- 35-60 ns per occurrence; round 5's threshold of 1 / 64 dated from a serial selection)
        dense.div = table_ok ? 4 : 0;
        st = force_windows ? ACGPU_ERR_NOMEM : nonoverlapping_parallel(aut, in, rule, out, cap, n_out, prof, &dense);
        if (st == ACGPU_ERR_NOMEM && !dense.hit) {   // the occurrence stream of the whole span does not fit: windows
            dense.div = 0;
            st = nonoverlapping_windowed(aut, in, rule, out, cap, n_out, &dense);
        }
        if (st == ACGPU_ERR_NOMEM && dense.hit && table_ok) {   // dense: select from the per-start table instead
            st = nonoverlapping_start_table(aut, in, rule, out, cap, n_out, prof, &served);
            if (served) return st;
            st = ACGPU_ERR_NOMEM;
        }
        if (st == ACGPU_ERR_NOMEM && dense.hit)    // tens of occurrences per byte: the serial loop is cheaper
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
namespace {

// First match of an eligible unanchored search, in parallel.  The span is scanned in growing windows; window k yields
// every occurrence with end <= b_k (earlier windows were empty), the selection rule picks its first match m, and m
// is final once every occurrence that could beat it is visible: always for Standard (first record of the stream),
// for the leftmost kinds when m.start + L <= b_k (an unseen occurrence ends after b_k, hence starts after b_k - L);
// otherwise the window is extended to m.start + L once.
acgpu_status find_parallel(acgpu_automaton* aut, const acgpu_input* in, int rule, int32_t* found, acgpu_match* m, DenseRule* dense) {
    acgpu_automaton* occ = aut->occ ? aut->occ.get() : aut;
    DeviceState* ds = nullptr;
    acgpu_status st = get_device_state(occ, &ds);
    if (st) return st;
    ScratchLease sc(ds);
    hipStream_t stream = static_cast<hipStream_t>(in->stream);
    const uint64_t L = occ->nnfa.max_pattern_len;
    uint64_t a = in->span_start, w = uint64_t(16) << 20;
    while (a < in->span_end) {
        uint64_t b = std::min<uint64_t>(in->span_end, a + w);
        const uint64_t lo = a > in->span_start + L ? a - L : in->span_start;
        for (int attempt = 0; attempt < 2; attempt++) {
            uint64_t n_sel = 0;
            st = nonoverlapping_core(occ, ds, sc.s.get(), in, size_t(lo), size_t(b), in->span_start, rule, &n_sel, nullptr, dense);
            if (st == ACGPU_ERR_NOMEM && !dense->hit && b - a > (uint64_t(64) << 10)) {   // occurrence stream of the window too large
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
    // Standard automata always report the earliest match (src/automaton.rs:1259-1275); `earliest` on a leftmost automaton is
    // the Standard rule over the same patterns (see acgpu_find_iter_ex).  Inputs the occurrence rule does not cover (anchored
    // searches, empty patterns) run the reference loop on one lane.
    const bool earliest_matters = in->earliest && aut->cfg.match_kind != ACGPU_MATCH_STANDARD;
    if (aut->cfg.engine != 1 && parallel_find_eligible(aut, in)) {
        DenseRule dense;
        st = find_parallel(aut, in, earliest_matters ? int(ACGPU_MATCH_STANDARD) : aut->cfg.match_kind, found, m, &dense);
        if (!(st == ACGPU_ERR_NOMEM && dense.hit)) return st;
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

}  // extern "C"
