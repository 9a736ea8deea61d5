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
            // (one or two distinct stragglers of 3..8 bytes: the large-set filter compares them in its producers' registers
            // beside the long patterns' tables -- short mode, host/pf_tables.hpp -- and the set stays whole)
            bool inline_shorts = !cfg.ascii_case_insensitive && !pp[1].empty();
            {
                std::vector<std::string> distinct;
                for (size_t i = 0; i < pp[1].size() && inline_shorts; i++) {
                    if (ll[1][i] < 3) { inline_shorts = false; break; }
                    std::string w(reinterpret_cast<const char*>(pp[1][i]), ll[1][i]);
                    if (std::find(distinct.begin(), distinct.end(), w) == distinct.end()) distinct.push_back(w);
                    if (distinct.size() > 2) inline_shorts = false;
                }
            }
            if (!inline_shorts && !pp[1].empty() && pp[1].size() <= kSplitMaxShort && pp[0].size() >= kSplitMinLong) {
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
    // the automata searched on this one's behalf (the Standard twin of a leftmost automaton, the parts of a split set) take
    // the same value: all of them are checked first, then all are set -- a refusal leaves every one of them as it was
    std::vector<acgpu_automaton*> all;
    std::vector<acgpu_automaton*> todo{aut};
    while (!todo.empty()) {
        acgpu_automaton* a = todo.back(); todo.pop_back();
        all.push_back(a);
        for (acgpu_automaton* child : {a->occ.get(), a->part[0].get(), a->part[1].get()}) if (child) todo.push_back(child);
    }
    for (acgpu_automaton* a : all) {
        std::lock_guard<std::mutex> lk(a->mu);
        if (!a->devs.empty()) { g_last_error = "acgpu_set_variant: the automaton is already on a device"; return ACGPU_ERR_INVALID_ARGUMENT; }
        if (!a->var.field(name)) { g_last_error = std::string("acgpu_set_variant: unknown variant ") + name; return ACGPU_ERR_INVALID_ARGUMENT; }
    }
    for (acgpu_automaton* a : all) {
        std::lock_guard<std::mutex> lk(a->mu);
        *a->var.field(name) = value;
    }
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
    // (the automaton the caller built, as the reference counts it -- plus the parts of a split set, which are searched in its place)
    size_t parts = 0;
    for (const acgpu_automaton* p : {a->part[0].get(), a->part[1].get()}) if (p) parts += acgpu_memory_usage(p);
    const size_t pl = a->nnfa.pattern_lens.size() * 4 + parts;
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

}  // extern "C"
extern "C" {

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

