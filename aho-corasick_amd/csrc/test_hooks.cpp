// Test hooks of libacgpu (include/acgpu_test.h): NOT search paths and not part of the product library -- they are built
// into libacgpu_testhooks.so (make testhooks; it links libacgpu.so) and into the host-ASan flavour.  Each builds the
// host-side tables of one device engine and replays that engine's decisions on the CPU, so that table construction and
// step logic are checked against the oracle without a GPU (tests/test_*_tables.py, tests/test_select_rule.py).
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "acgpu.h"
#include "acgpu_test.h"
#include "capi_internal.hpp"
#include "device/cnfa_tri_step.hpp"
#include "device/dfa_tri_step.hpp"
#include "device/engines.hpp"
#include "device/select.hpp"
#include "host/automaton.hpp"
#include "host/cnfa_tables.hpp"
#include "host/cnfa_tri_tables.hpp"
#include "host/dfa_tri_tables.hpp"
#include "host/engine_plan.hpp"
#include "host/lw_tables.hpp"
#include "host/pf_tables.hpp"

using namespace acgpu;

namespace {
// The shallow-skip walks (device/cnfa_tri_step.hpp, device/dfa_tri_step.hpp) lane by lane over the chunk grid of a search of
// the whole haystack (chunk size from the automaton's configuration): warm-up, ownership and piece bounds as in
// k_tri_walk.  make(ci): a walker with its tables bound; records(state, f): the pattern ids of an event's state.
// Returns the match count; *hash = FNV-1a over the (pattern, start, end) of the records the events stand for, in output
// order (0: the events do not tile the output).
template <class Walk, class Make, class Records>
uint64_t run_tri_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint32_t n_used, uint32_t apair,
                      uint32_t start_mlen, Make make, Records records, uint64_t* hash_out) {
    ScanGeom g{};
    g.hay16 = haystack; g.base_mis = 0; g.cold_floor = 0; g.emit_lo = 0; g.emit_hi = len;
    g.chunk = acgpu_default_chunk(aut, len);
    g.halo = uint32_t(aut->nnfa.max_pattern_len > 0 ? aut->nnfa.max_pattern_len - 1 : 0);
    g.grid0 = 0;
    g.n_chunks = std::max<uint64_t>(1, (g.emit_hi + g.chunk - 1) / g.chunk);
    g.emit_start_matches = 1;
    uint64_t total = 0;
    std::vector<TriEvent> events(std::min<size_t>(len + 2, size_t(1) << 26));   // (at most one event per position)
    std::vector<uint64_t> offsets(g.n_chunks, 0);
    unsigned long long n_events = 0;
    for (uint64_t ci = 0; ci < g.n_chunks; ci++) {
        const ChunkRange r = chunk_range(g, ci);
        uint8_t lane_buf[16];
        Walk f = make();
        f.s_buf = lane_buf;
        f.ua = f.ub = f.na = f.nb = n_used;
        f.ev_buf = events.data(); f.ev_ctr = &n_events; f.ev_max_segs = uint32_t(events.size()); f.ci = uint32_t(ci);
        if (ci == 0 && start_mlen) {
            f.note_event(0x80000000u | (n_used * apair + n_used), 0, start_mlen);
            f.flush_events(-1);
        }
        const uint64_t p0 = r.w & ~uint64_t(63);
        const int32_t w_rel = int32_t(r.w - p0), lo_rel = int32_t(r.lo - p0), hi_rel = int32_t(r.hi - p0);
        auto clamp16 = [](int32_t x) -> uint32_t { return uint32_t(x < 0 ? 0 : (x > 16 ? 16 : x)); };
        for (int32_t pv = 0; pv < hi_rel; pv += 16) {
            uint32_t wds[4] = {0, 0, 0, 0};
            for (int i = 0; i < 16; i++)
                if (p0 + pv + i < len) wds[i >> 2] |= uint32_t(haystack[p0 + pv + i]) << (8 * (i & 3));
            const uint32_t lo_i = clamp16(w_rel - pv), hi_i = clamp16(hi_rel - pv), own_from = clamp16(lo_rel - pv);
            const uint32_t act16 = ((1u << hi_i) - 1u) & ~((1u << lo_i) - 1u);
            if (f.sm) {
                if (act16 == 0xFFFFu) f.template piece_scan<true, true>(wds, act16);
                else f.template piece_scan<false, true>(wds, act16);
            } else {
                if (act16 == 0xFFFFu) f.template piece_scan<true, false>(wds, act16);
                else f.template piece_scan<false, false>(wds, act16);
            }
            f.piece_walk(hi_i, own_from, pv - int32_t(int64_t(ci * uint64_t(g.chunk)) - int64_t(p0)));
        }
        offsets[ci] = total;
        total += f.cnt;
    }
    uint64_t hash = 0xCBF29CE484222325ull;
    std::vector<std::pair<uint64_t, uint32_t>> order;   // (output slot, event)
    for (uint64_t e = 0; e < n_events; e++) order.emplace_back(offsets[events[e].ci] + events[e].pre, uint32_t(e));
    std::sort(order.begin(), order.end());
    uint64_t slot = 0;
    bool dense = true;
    auto mix = [&](uint64_t w) { for (int i = 0; i < 8; i++) hash = (hash ^ ((w >> (8 * i)) & 0xFF)) * 0x100000001B3ull; };
    std::vector<uint32_t> pids;
    for (const auto& oe : order) {
        const TriEvent& e = events[oe.second];
        const uint64_t end = uint64_t(e.ci) * g.chunk + uint64_t(int64_t(int32_t(e.rel))) + 1;
        records(e.state, pids);
        if (oe.first != slot) dense = false;
        for (uint32_t pid : pids) { mix(pid); mix(end - aut->nnfa.pattern_lens[pid]); mix(end); }
        slot += pids.size();
    }
    if (!dense || slot != total) hash = 0;   // the events must tile the output exactly
    *hash_out = hash;
    return total;
}
}  // namespace

extern "C" {

// Test hook (NOT a search path): runs the selection rule of device/select.hpp on a host-resident ordered
// occurrence stream, so that the rule itself can be checked against the oracle without a GPU.
acgpu_status acgpu_test_select_host(const acgpu_match* stream, size_t n, int32_t match_kind, size_t span_start,
                                    size_t max_pattern_len, acgpu_match* out, size_t cap, size_t* n_out) {
    if (!n_out || (n && !stream)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = size_t(select_nonoverlapping(stream, n, match_kind, span_start, max_pattern_len,
                                          [&](uint64_t k, const acgpu_match& mm) { if (k < cap && out) out[k] = mm; }));
    return *n_out > cap ? ACGPU_ERR_BUFFER_TOO_SMALL : ACGPU_OK;
}

// Test hook (NOT a search path): the LDS-walk engine's tables built on the host and walked by the CPU emulation of the
// kernel's step rules (host/lw_tables.cpp).
acgpu_status acgpu_test_lw_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint64_t* n_matches,
                                uint64_t* info) {
    if (!aut || !n_matches || !info || (len && !haystack)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_matches = 0;
    const int force_flavour = int(info[0]) - 1, force_cls = int(info[1]) - 1;   // inputs: 0 = the engine's own choice
    std::memset(info, 0, 8 * sizeof(uint64_t));
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD || aut->cfg.start_kind != ACGPU_START_UNANCHORED || !aut->has_dfa)
        return ACGPU_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> order, sid2hid;
    uint32_t first_match = 0;
    hid_order(aut->nnfa, order, sid2hid, first_match);
    LwHostTables t;
    if (!build_lw_host(aut->nnfa, aut->dfa, order, sid2hid, first_match, t, force_flavour, force_cls)) return ACGPU_OK;   // info[0] == 0: not eligible
    uint64_t redo = 0;
    *n_matches = lw_emulate_count(t, haystack, len, &redo);
    info[0] = 1; info[1] = t.image.size() * 4; info[2] = t.n_dense; info[3] = t.n_multi; info[4] = t.classes;
    info[5] = t.n_states; info[6] = redo;
    info[7] = (t.wide() ? 1 : 0) | (t.flavour == kLwFull ? 2 : 0) | (t.computed_cls ? 4 : 0) | (t.monotone ? 8 : 0) | (t.disjoint ? 16 : 0) |
              (uint64_t(lw_estimate_redo(t) * 1e6) << 8);
    return ACGPU_OK;
}

// Test hook (NOT a search path): the records of the one-row-per-state form of the LDS walk (match lists in the LDS image),
// produced on the host the way k_lw_fill produces them.  *n_out = number of records (0 and ACGPU_OK with *served = 0 when
// the automaton has no such form).
acgpu_status acgpu_test_lw_records_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, acgpu_match* out, size_t cap,
                                        size_t* n_out, int32_t* served) {
    if (!aut || !n_out || !served || (len && !haystack)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = 0; *served = 0;
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD || aut->cfg.start_kind != ACGPU_START_UNANCHORED || !aut->has_dfa)
        return ACGPU_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> order, sid2hid;
    uint32_t first_match = 0;
    hid_order(aut->nnfa, order, sid2hid, first_match);
    LwHostTables t;
    if (!build_lw_host(aut->nnfa, aut->dfa, order, sid2hid, first_match, t, kLwFull, -1)) return ACGPU_OK;
    std::vector<acgpu_match> rec;
    if (!lw_emulate_records(t, haystack, len, rec)) return ACGPU_OK;
    *served = 1;
    *n_out = rec.size();
    if (rec.size() > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (!rec.empty()) std::memcpy(out, rec.data(), rec.size() * sizeof(acgpu_match));
    return ACGPU_OK;
}

// Test hook (NOT a search path): the same records through the EVENT form of the LDS walk (device/lds_emit.hip) modelled on the
// host: lane-chunks of `chunk` bytes, events, scan, re-walk of the events in reverse order (host/lw_tables.cpp).
acgpu_status acgpu_test_lw_event_records_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint32_t chunk,
                                              acgpu_match* out, size_t cap, size_t* n_out, int32_t* served) {
    if (!aut || !n_out || !served || (len && !haystack)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = 0; *served = 0;
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD || aut->cfg.start_kind != ACGPU_START_UNANCHORED || !aut->has_dfa)
        return ACGPU_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> order, sid2hid;
    uint32_t first_match = 0;
    hid_order(aut->nnfa, order, sid2hid, first_match);
    LwHostTables t;
    if (!build_lw_host(aut->nnfa, aut->dfa, order, sid2hid, first_match, t, kLwFull, -1)) return ACGPU_OK;
    std::vector<acgpu_match> rec;
    const uint32_t halo = uint32_t(aut->nnfa.max_pattern_len > 0 ? aut->nnfa.max_pattern_len - 1 : 0);
    if (!lw_emulate_event_records(t, haystack, len, chunk, halo, rec)) return ACGPU_OK;
    *served = 1;
    *n_out = rec.size();
    if (rec.size() > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (!rec.empty()) std::memcpy(out, rec.data(), rec.size() * sizeof(acgpu_match));
    return ACGPU_OK;
}

// Test hook (NOT a search path): the routing rules of host/engine_plan.hpp, which capi_overlap.cpp and capi_enqueue.cpp apply, on explicit facts.
// facts[0..7] = {has_dfa, pf_ready, lw_ready, pfx_ready, min_pattern_len, want, routing, lw_full}; hints[0..1] = {probe_skip, route_hint};
// out[0..2] = {first engine (0: the request cannot be honoured), alternative, how a prefix-filter scan starts (PfStart)}.
acgpu_status acgpu_test_engine_plan(const uint64_t* facts, const int32_t* hints, uint64_t span_bytes, int32_t first_kernel_is_large_set,
                                    uint32_t* out) {
    if (!facts || !hints || !out) return ACGPU_ERR_INVALID_ARGUMENT;
    EngineFacts f;
    f.has_dfa = facts[0] != 0; f.pf_ready = facts[1] != 0; f.lw_ready = facts[2] != 0; f.pfx_ready = facts[3] != 0;
    f.min_pattern_len = size_t(facts[4]); f.want = int(facts[5]); f.routing = facts[6] != 0; f.lw_full = facts[7] != 0;
    const EnginePlan p = plan_engines(f);
    out[0] = p.first; out[1] = p.alternative;
    out[2] = uint32_t(plan_pf_start(p, hints[0], hints[1], span_bytes, first_kernel_is_large_set != 0));
    return ACGPU_OK;
}

uint32_t acgpu_test_event_order_shift(uint64_t max_events, uint64_t max_records, uint64_t span_bytes) {
    return event_order_shift(max_events, max_records, span_bytes);
}

// Test hook (NOT a search path): the prefix filters' tables built on the host and their decisions replayed on the CPU
// (host/pf_tables.cpp).
acgpu_status acgpu_test_pf_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, int32_t kernel,
                                uint64_t* n_matches, uint64_t* info) {
    if (!aut || !n_matches || !info || (len && !haystack) || kernel < 0 || kernel > 4) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_matches = 0;
    std::memset(info, 0, 8 * sizeof(uint64_t));
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD || aut->cfg.start_kind != ACGPU_START_UNANCHORED)
        return ACGPU_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> order, sid2hid;
    uint32_t first_match = 0;
    hid_order(aut->nnfa, order, sid2hid, first_match);
    PfHostTables t;
    if (!build_pf_host(aut->nnfa, order, sid2hid, t, aut->var.pfx_tails, aut->var.pfx_key8_x2 != 0, aut->var.pfx_short != 0 && aut->var.pfx_key8 != 0)) return ACGPU_OK;   // info[0] == 0: not served by the filters
    info[0] = 1; info[1] = t.pfx_ok ? 1 : 0; info[4] = t.pfx_map8.empty() ? 4 : t.pfx_depth; info[5] = t.n_patterns;
    info[6] = (t.exact2 ? 1 : 0) | (t.fold ? 2 : 0); info[7] = t.use3 ? 1 : 0;
    uint64_t sv[4] = {0, 0, 0, 0};   // level-1 survivors, level-2 hits, starts the exact-prefix bit table lets through, hits decided by a chain tail
    const uint64_t n = pf_emulate_count(t, sid2hid[aut->nnfa.special.start_unanchored_id], haystack, len, kernel, sv);
    if (n == ~uint64_t(0)) { info[1] = 0; return ACGPU_OK; }
    *n_matches = n;
    info[2] = sv[0]; info[3] = sv[1];
    info[6] |= sv[3] << 8;                       // level-2 hits decided by a chain tail instead of a walk
    info[7] |= uint64_t(t.pfx_tail_nodes) << 8;  // prefix nodes that have a tail record
    return ACGPU_OK;
}

// Test hook (NOT a search path): the contiguous-NFA walk kernel's tables built on the host and its step replayed on the
// CPU (host/cnfa_tables.cpp).
acgpu_status acgpu_test_cnfa_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint64_t* n_matches,
                                  uint64_t* info) {
    if (!aut || !n_matches || !info || (len && !haystack)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_matches = 0;
    std::memset(info, 0, 8 * sizeof(uint64_t));
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD || aut->cfg.start_kind == ACGPU_START_ANCHORED || !aut->has_cnfa)
        return ACGPU_ERR_INVALID_ARGUMENT;
    CnfaHotHost t;
    if (!build_cnfa_hot_host(aut->cnfa, t)) return ACGPU_OK;   // info[0] == 0: the kernel does not serve this automaton
    info[0] = 1; info[1] = t.n_slots; info[2] = t.dense_outside ? 1 : 0; info[3] = t.sorted_sparse ? 1 : 0;
    info[4] = t.slot_matches ? 1 : 0;
    for (size_t i = 0; i < aut->cnfa.repr.size(); i++) if (t.repr_t[i] & kCnfaSlotTag) info[5]++;   // patched words
    *n_matches = cnfa_emulate_count(t, aut->cnfa, haystack, len);
    return ACGPU_OK;
}

// Test hook (NOT a search path): the contiguous-NFA shallow-skip walk (device/cnfa_tri.hip), its tables built on the host and
// the kernel's own per-piece code run on the CPU.
acgpu_status acgpu_test_cnfa_tri_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint64_t* n_matches,
                                      uint64_t* info) {
    if (!aut || !n_matches || !info || (len && !haystack)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_matches = 0;
    std::memset(info, 0, 8 * sizeof(uint64_t));
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD || aut->cfg.start_kind == ACGPU_START_ANCHORED || !aut->has_cnfa)
        return ACGPU_ERR_INVALID_ARGUMENT;
    CnfaTriHost t;
    if (!build_cnfa_tri_host(aut->cnfa, t)) return ACGPU_OK;   // info[0] == 0: the kernel does not serve this automaton
    info[0] = 1; info[1] = t.n_used; info[2] = t.bw; info[3] = t.granule; info[4] = t.shallow_matches ? 1 : 0;
    info[5] = t.lds_bytes;
    uint64_t steps[3] = {0, 0, 0};
    const uint64_t model = cnfa_tri_emulate_count(t, aut->cnfa, haystack, len, steps);
    info[6] = steps[0] + steps[1] + steps[2];
    uint32_t gshift = 0;
    while ((1u << gshift) < t.granule) gshift++;
    auto make = [&]() {
        TriWalk f;
        f.s_bits = t.bits.data(); f.s_base = t.base.data(); f.s_uc = t.uc.data(); f.s_inv = t.inv.data(); f.s_mc2 = t.mc2.data();
        f.A = t.apair; f.bw = t.bw; f.gshift = gshift; f.U = t.n_used; f.sm = t.shallow_matches ? 1u : 0u;
        f.n_child = uint32_t(t.child.size());
        f.child = t.child.data(); f.repr3 = t.repr3.data(); f.alen = uint32_t(aut->cnfa.alphabet_len);
        f.max_match = aut->cnfa.special.max_match_id; f.repr_words = uint32_t(t.repr3.size());
        return f;
    };
    auto records = [&](uint32_t state, std::vector<uint32_t>& pids) {   // k_cnfa_tri_emit (contiguous.rs:611-633)
        pids.clear();
        const uint32_t st = (state & 0x80000000u) ? t.st2[state & 0x7FFFFFFFu] : state;
        const uint32_t kind = t.repr3[st] & 0xFFu;
        const uint32_t base = st + (kind == 0xFFu ? 2 + uint32_t(aut->cnfa.alphabet_len) : 2 + ((kind + 3) >> 2) + kind);
        const uint32_t packed = t.repr3[base];
        if (packed & (1u << 31)) pids.push_back(packed & 0x7FFFFFFFu);
        else for (uint32_t k = 0; k < packed; k++) pids.push_back(t.repr3[base + 1 + k]);
    };
    const uint64_t total = run_tri_host<TriWalk>(aut, haystack, len, t.n_used, t.apair, t.start_mlen, make, records, &info[7]);
    *n_matches = total == model ? total : ~uint64_t(0);   // the two must agree; the tests compare with the oracle
    return ACGPU_OK;
}

// Test hook (NOT a search path): the DFA shallow-skip walk (device/dfa_tri.hip), its tables built on the host (from the
// automaton's DFA, or for NFA kinds from the DFA of the same noncontiguous NFA, as the upload derives it) and the
// kernel's own per-piece code run on the CPU.
acgpu_status acgpu_test_dfa_tri_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint64_t* n_matches,
                                     uint64_t* info) {
    if (!aut || !n_matches || !info || (len && !haystack)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_matches = 0;
    std::memset(info, 0, 8 * sizeof(uint64_t));
    if (aut->cfg.match_kind != ACGPU_MATCH_STANDARD || aut->cfg.start_kind != ACGPU_START_UNANCHORED) return ACGPU_ERR_INVALID_ARGUMENT;
    Dfa tmp;
    const Dfa* d = &aut->dfa;
    if (!aut->has_dfa) {
        if (build_dfa(aut->nnfa, ACGPU_START_UNANCHORED, true, tmp) != ACGPU_OK) return ACGPU_OK;
        d = &tmp;
    }
    DfaTriHost t;
    if (!build_dfa_tri_host(aut->nnfa, *d, t)) return ACGPU_OK;   // info[0] == 0: the kernel does not serve this automaton
    info[0] = 1; info[1] = t.n_used; info[2] = t.bw; info[3] = t.granule; info[4] = t.shallow_matches ? 1 : 0;
    info[5] = t.lds_bytes;
    // the model: the reference loop over the DFA (dfa.rs:218-226 inside automaton.rs:1491-1534), and how many of its
    // steps start below depth 2 (what the kernel pays a gather for)
    uint64_t model = 0, deep_steps = 0;
    {
        const uint32_t s2 = uint32_t(d->stride2);
        auto mlen = [&](uint32_t sid) -> uint32_t {
            if (sid == 0 || sid > d->special.max_match_id) return 0;
            const uint32_t o = (sid >> s2) - 2;
            return d->moff[o + 1] - d->moff[o];
        };
        uint32_t sid = d->special.start_unanchored_id;
        model += mlen(sid);
        for (size_t at = 0; at < len; at++) {
            if (aut->nnfa.depth[sid >> s2] > 2) deep_steps++;
            sid = d->trans[sid + d->byte_classes[haystack[at]]];
            model += mlen(sid);
        }
    }
    info[6] = deep_steps;
    uint32_t gshift = 0;
    while ((1u << gshift) < t.granule) gshift++;
    auto make = [&]() {
        DfaTriWalk f;
        f.s_bits = t.bits.data(); f.s_base = t.base.data(); f.s_uc = t.uc.data(); f.s_inv = t.inv.data(); f.s_mc2 = t.mc2.data();
        f.A = t.apair; f.bw = t.bw; f.gshift = gshift; f.U = t.n_used; f.sm = t.shallow_matches ? 1u : 0u;
        f.n_child = uint32_t(t.child.size());
        f.child = t.child.data(); f.trans3 = t.trans3.data(); f.moff = d->moff.data(); f.stride2 = uint32_t(d->stride2);
        f.max_match = d->special.max_match_id; f.trans_words = uint32_t(t.trans3.size());
        return f;
    };
    auto records = [&](uint32_t state, std::vector<uint32_t>& pids) {   // k_dfa_tri_emit (dfa.rs:275-286)
        pids.clear();
        const uint32_t sid = (state & 0x80000000u) ? t.st2[state & 0x7FFFFFFFu] : state;
        const uint32_t o = (sid >> d->stride2) - 2;
        for (uint32_t k = d->moff[o]; k < d->moff[o + 1]; k++) pids.push_back(d->mpid[k]);
    };
    const uint64_t total = run_tri_host<DfaTriWalk>(aut, haystack, len, t.n_used, t.apair, t.start_mlen, make, records, &info[7]);
    *n_matches = total == model ? total : ~uint64_t(0);
    return ACGPU_OK;
}
}  // extern "C"
