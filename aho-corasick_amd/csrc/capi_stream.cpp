// C ABI of libacgpu.so: replace_all and the stream search (StreamChunkIter, src/automaton.rs:1036-1244).  See capi.cpp.
#include "capi_impl.hpp"

using namespace acgpu;
using namespace acgpu_capi;

namespace {
acgpu_status stream_feed_once(acgpu_stream* s, const uint8_t* bytes, size_t len, int32_t bytes_on_device, void* hip_stream, size_t* n_matches);
}  // namespace

namespace {
// what a feed advances (acgpu_stream): saved in front of a feed that is split into pieces, restored if a later piece fails
struct StreamState { std::vector<uint8_t> halo; std::vector<acgpu_match> last; uint64_t total, pos; };
StreamState save_stream_state(const acgpu_stream* s) { return StreamState{s->halo, s->last, s->total, s->pos}; }
void restore_stream_state(acgpu_stream* s, const StreamState& st) { s->halo = st.halo; s->last = st.last; s->total = st.total; s->pos = st.pos; }
}  // namespace

extern "C" {

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
        // a feed either succeeds as a whole or leaves the stream where it was: the pieces already consumed are rolled back
        const StreamState saved = save_stream_state(s);
        for (size_t k = 0; k < pipe.n_pieces; k++) {
            const size_t off = k * piece, nb = std::min(piece, len - off);
            size_t nk = 0;
            if (hipError_t he = pipe.wait(k, static_cast<hipStream_t>(hip_stream)); he != hipSuccess) { restore_stream_state(s, saved); return hip_fail(he, "HostPipe::wait"); }
            if ((pst = acgpu_stream_feed(s, s->stage.as<uint8_t>() + off, nb, 1, hip_stream, &nk))) { restore_stream_state(s, saved); return pst; }
            acc.insert(acc.end(), s->last.begin(), s->last.end());
        }
        s->last.swap(acc);
        *n_matches = s->last.size();
        return ACGPU_OK;
    }
    const bool force_split = len > (size_t(64) << 10) && s->aut->var.stream_split != 0;   // (variant)
    acgpu_status st = force_split ? ACGPU_ERR_NOMEM : stream_feed_once(s, bytes, len, bytes_on_device, hip_stream, n_matches);
    if (st == ACGPU_ERR_NOMEM && len > (size_t(64) << 10)) {
        // the occurrence stream of this chunk does not fit in device memory: feeding it as two halves is the same
        // stream search (state is only advanced by a feed that succeeds)
        const size_t h = len / 2;
        size_t n1 = 0, n2 = 0;
        const StreamState saved = save_stream_state(s);
        if ((st = acgpu_stream_feed(s, bytes, h, bytes_on_device, hip_stream, &n1))) return st;
        std::vector<acgpu_match> acc;
        acc.swap(s->last);
        if ((st = acgpu_stream_feed(s, bytes + h, len - h, bytes_on_device, hip_stream, &n2))) { restore_stream_state(s, saved); return st; }   // (the first half is rolled back)
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
        DenseRule never_dense;       // the stream search has no serial alternative to hand a dense chunk to
        never_dense.guard = false;
        if ((st = nonoverlapping_core(aut, ds, sc.s.get(), &in, halo, local, pos0, ACGPU_MATCH_STANDARD, &n_sel, nullptr, &never_dense)))
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

}  // extern "C"
