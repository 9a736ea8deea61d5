// Fast-path tables of the Standard/unanchored full DFA (hid renumbering, 256-wide u16 table, prefix-filter tables)
// and the LDS-row FILL kernel of the count -> scan -> fill pipeline.  The count kernels live in lds_walk.hip (LDS
// transition walk) and pf_scan.hip (prefix filter).
// Match states are recognised by one compare (hid >= first_match) exactly like the reference's
// `sid <= max_special_id` trick (src/dfa.rs:229-241), just with the order reversed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "../host/lw_tables.hpp"
#include "../host/pf_tables.hpp"
#include "hot.hpp"
#include "launch_util.hpp"
#include "tile_walk.hpp"

namespace acgpu {

namespace {

constexpr uint32_t kMaxHotRows = 192;  // 192 x 512 B = 96 KiB of LDS

// ------------------------------------------------------------------------------------- fill
// Same job as k_walk_fill (kernels.hip) for automata that have hot tables: one wavefront per non-empty chunk, the
// chunk and its warm-up staged in LDS, lane l walks sub-range l once and remembers its first match events.  The walk
// itself never leaves LDS on the common path: byte -> class -> class-compressed u16 row of the start state / the
// distance-1 states (rebuilt per workgroup from the 256-wide table); deeper states read the global table.  A step
// is three dependent LDS reads (~0.3 us) instead of an L2 round trip (~1.25 us measured), and the match records are
// produced from the reference's own match lists (hid -> DFA state id), so the order inside one `end` is the state's
// match-list order exactly as in k_walk_fill.
constexpr int kHfWaves = 16;
constexpr uint32_t kHfStage = 2048 + 512;   // staged bytes per wavefront (chunk + warm-up + alignment slack)
constexpr int kHfEvents = 2;

__global__ __launch_bounds__(kHfWaves * 64) void k_hot_fill(DfaEng eng, const uint16_t* __restrict__ tab,
                                                            const uint32_t* __restrict__ hid2sid, uint32_t n_hot,
                                                            uint32_t first_match, uint32_t start, ScanGeom g,
                                                            const uint64_t* __restrict__ active,
                                                            const uint64_t* __restrict__ totals, uint64_t cap,
                                                            const uint64_t* __restrict__ aoff,
                                                            acgpu_match* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint8_t s_cls[256];
    __shared__ uint8_t s_rep[256];
    const uint32_t ncls = 1u << eng.d.stride2;                                   // row stride of the compressed table
    uint16_t* s_ctab = reinterpret_cast<uint16_t*>(smem);                        // [n_hot][ncls]
    uint8_t* s_hay_all = smem + ((size_t(n_hot) * ncls * 2 + 15) & ~size_t(15));  // kHfWaves * kHfStage
    const uint64_t n_active = totals[1];
    if (totals[0] > cap || uint64_t(blockIdx.x) * kHfWaves >= n_active) return;
    if (threadIdx.x < 256) {
        const uint8_t c = eng.cls[threadIdx.x];
        s_cls[threadIdx.x] = c;
        s_rep[c] = uint8_t(threadIdx.x);   // any byte of the class: they share every transition
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_hot * ncls; i += kHfWaves * 64) {
        const uint32_t h = i >> eng.d.stride2, c = i & (ncls - 1);
        s_ctab[i] = tab[(h << 8) | s_rep[c]];   // classes beyond alphabet_len are never looked up
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* s_hay = s_hay_all + size_t(wave) * kHfStage;
    auto write = [&](uint32_t sid, uint64_t end, acgpu_match* dst) -> uint32_t {
        const uint32_t n = eng.match_len(sid);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t pid = eng.match_pattern(sid, i);
            acgpu_match m; m.pattern = pid; m._pad = 0; m.end = end; m.start = end - eng.pattern_len(pid);
            dst[i] = m;
        }
        return n;
    };
    for (uint64_t a = uint64_t(blockIdx.x) * kHfWaves + wave; a < n_active; a += uint64_t(gridDim.x) * kHfWaves) {
        const uint64_t ci = active[a];
        const ChunkRange r = chunk_range(g, ci);
        const uint64_t w16 = r.w & ~uint64_t(15);
        for (uint64_t o = uint64_t(lane) * 16; w16 + o < r.hi; o += 64 * 16) {   // host guarantees r.hi - w16 <= kHfStage
            ACGPU_HAY_CHECK(g, w16 + o, 16);
            *reinterpret_cast<uint4*>(s_hay + o) = *reinterpret_cast<const uint4*>(g.hay16 + w16 + o);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // split [r.lo, r.hi) into 64 sub-ranges of `sub` bytes (the last ones may be empty)
        const uint64_t len = r.hi - r.lo;
        const uint64_t sub = (len + 63) / 64;
        uint64_t lo = r.lo + uint64_t(lane) * sub, hi = lo + sub;
        if (lo > r.hi) lo = r.hi;
        if (hi > r.hi) hi = r.hi;
        uint64_t w = lo >= g.halo ? lo - g.halo : 0;
        if (w < g.cold_floor) w = g.cold_floor;
        const bool sm = ci == 0 && lane == 0 && g.emit_start_matches && start >= first_match;
        uint64_t ev_end[kHfEvents] = {};
        uint32_t ev_hid[kHfEvents] = {}, nev = 0, c = 0;
        auto walk = [&](bool emit_now, acgpu_match* dst) {
            uint32_t n_out = 0, k_ev = 0;
            auto event = [&](uint32_t h, uint64_t end) {
                if (emit_now) { n_out += write(hid2sid[h], end, dst + n_out); return; }
#pragma unroll
                for (int k = 0; k < kHfEvents; k++) if (k_ev == uint32_t(k)) { ev_end[k] = end; ev_hid[k] = h; }
                k_ev++;
                n_out += eng.match_len(hid2sid[h]);
            };
            if (sm) event(start, g.cold_floor - g.base_mis);
            uint32_t h = start;
            if (hi > lo)
                for (uint64_t v = w; v < hi; v++) {
                    const uint32_t byte = s_hay[v - w16];
                    h = h < n_hot ? s_ctab[(h << eng.d.stride2) | s_cls[byte]] : tab[(h << 8) | byte];
                    if (h >= first_match && v >= lo) event(h, v + 1 - g.base_mis);
                }
            nev = k_ev;
            return n_out;
        };
        c = walk(false, nullptr);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (c) {
            acgpu_match* dst = out + aoff[a] + (incl - c);
            if (nev <= uint32_t(kHfEvents)) {
#pragma unroll
                for (int k = 0; k < kHfEvents; k++)
                    if (uint32_t(k) < nev) dst += write(hid2sid[ev_hid[k]], ev_end[k], dst);
            } else {
                (void)walk(true, dst);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // s_hay is reused by this wave's next chunk
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

hipError_t build_hot_tables(const NNfa& n, const Dfa& d, const Variants& var, HotTables& out) {
    out.ready = false;
    out.var = var;
    std::vector<uint32_t> order, sid2hid;   // hid -> nnfa sid and back (host/lw_tables.cpp)
    uint32_t first_match = 0;
    hid_order(n, order, sid2hid, first_match);
    const uint32_t su = n.special.start_unanchored_id;
    const size_t nh = order.size();
    const bool small = nh <= 65535;   // the 256-wide u16 table of the LDS-row engines needs 16-bit state ids
    if (nh > kPfMaxStates) return hipSuccess;
    // hot rows: start state + non-match states at distance 1 (stored depth 0), capped by the LDS budget
    uint32_t n_hot = 1;  // DEAD row is row 0 (all zeros); it is never looked up, but keeps indices simple
    for (size_t h = 1; h < first_match; h++) {
        const uint32_t s = order[h];
        const bool shallow = (s == su) || n.depth[s] == 0;
        if (!shallow || n_hot >= kMaxHotRows) break;
        n_hot++;
    }
    std::vector<uint16_t> tab(small ? nh * 256 : 0, 0);
    std::vector<uint32_t> hid2sid(nh, 0);
    for (size_t h = 1; h < nh; h++) {
        const uint32_t s = order[h];
        hid2sid[h] = s << d.stride2;
        if (!small) continue;
        const uint32_t* row = &d.trans[size_t(s) << d.stride2];
        for (int b = 0; b < 256; b++) tab[h * 256 + b] = uint16_t(sid2hid[row[d.byte_classes[b]] >> d.stride2]);
    }
    hipError_t e;
    if (small) {
        if ((e = hipMalloc(reinterpret_cast<void**>(&out.tab), tab.size() * sizeof(uint16_t))) != hipSuccess) return e;
        if ((e = hipMemcpy(out.tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.hid2sid), hid2sid.size() * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipMemcpy(out.hid2sid, hid2sid.data(), hid2sid.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.n_states = uint32_t(nh);
    out.first_match = first_match;
    out.n_hot = n_hot;
    out.start = sid2hid[su];
    out.ready = small;
    if ((e = build_lw_tables(n, d, order, sid2hid, first_match, out)) != hipSuccess) return e;

    // ---- prefix-filter tables (pf_scan.hip, pfx_scan.hip): built on the host (host/pf_tables.cpp), uploaded here
    out.pf_ready = false;
    PfHostTables t;
    if (!build_pf_host(n, order, sid2hid, t, var.pfx_tails, var.pfx_key8_x2 != 0, var.pfx_short != 0 && var.pfx_key8 != 0)) return hipSuccess;
    auto up = [&](auto** dst, const auto& v) -> hipError_t {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (v.empty()) return hipSuccess;
        if (hipError_t er = hipMalloc(reinterpret_cast<void**>(dst), v.size() * sizeof(T)); er != hipSuccess) return er;
        return hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    };
    out.pf_exact2 = t.exact2;
    out.pf_fold = t.fold;
    if ((e = up(&out.pf_bits3, t.bits3)) != hipSuccess) return e;
    out.pf_bits3_log2 = t.bits3_log2;
    if ((e = up(&out.pf_bits2, t.bits2)) != hipSuccess) return e;
    if ((e = up(&out.pf_bits, t.bits)) != hipSuccess) return e;
    out.pf_bits_bytes = t.bits_bytes;
    if ((e = up(&out.atab, t.atab)) != hipSuccess) return e;
    if ((e = up(&out.own_cnt, t.own)) != hipSuccess) return e;
    if ((e = up(&out.own_pid, t.own_pid)) != hipSuccess) return e;
    if ((e = up(&out.acls, t.acls)) != hipSuccess) return e;
    out.ashift = t.ashift;
    out.n_patterns = t.n_patterns;
    if (t.pfx_ok) {
        if ((e = up(&out.pfx_map, t.pfx_map)) != hipSuccess) return e;
        out.pfx_map_log2 = t.pfx_map_log2;
        out.pfx_prefixes = t.pfx_prefixes;
        if (!t.pfx_map8.empty()) {
            if ((e = up(&out.pfx_map8, t.pfx_map8)) != hipSuccess) return e;
            out.pfx_map8_log2 = t.pfx_map8_log2;
            out.pfx_depth = t.pfx_depth;
            if (!t.pfx_tails.empty() && (e = up(&out.pfx_tails, t.pfx_tails)) != hipSuccess) return e;
        }
        if ((e = up(&out.pfx_bits, t.xbits)) != hipSuccess) return e;
        if (!t.xbits8.empty() && (e = up(&out.pfx_bits8, t.xbits8)) != hipSuccess) return e;
        if (!t.xbits8x2.empty() && (e = up(&out.pfx_bits8x2, t.xbits8x2)) != hipSuccess) return e;
        out.pfx_short_n = t.short_n;
        for (uint32_t i = 0; i < t.short_n; i++) {
            out.pfx_short_lo[i] = t.short_lo[i]; out.pfx_short_hi[i] = t.short_hi[i];
            out.pfx_short_len[i] = t.short_len[i]; out.pfx_short_node[i] = t.short_node[i];
        }
        out.pfx4_complete = t.pfx4_complete;
        out.pfx_ready = true;
    }
    out.pf_ready = true;
    return hipSuccess;
}

bool hot_fill_supported(const HotTables& h, const ScanGeom& g) {
    if (!h.ready) return false;
    return uint64_t(g.chunk) + g.halo + 32 <= kHfStage;   // chunk + warm-up + 16-byte alignment of both ends
}

// max_waves: upper bound of the number of non-empty chunks the grid should cover in one pass (grid-stride beyond)
hipError_t launch_hot_fill(const HotTables& h, const DevAutomaton& a, const ScanGeom& g, const uint64_t* active,
                           const uint64_t* totals, uint64_t cap, uint64_t max_waves, const uint64_t* aoff,
                           acgpu_match* out, hipStream_t s) {
    uint64_t waves = max_waves < g.n_chunks ? max_waves : g.n_chunks;
    uint64_t blocks = (waves + kHfWaves - 1) / kHfWaves;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    DfaEng eng; eng.d = a.dfa; eng.cls = a.dfa.classes;
    const size_t smem = ((size_t(h.n_hot) * (size_t(1) << a.dfa.stride2) * 2 + 15) & ~size_t(15)) + size_t(kHfWaves) * kHfStage;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_hot_fill), 160 * 1024 - 1024); e != hipSuccess) return e;
    k_hot_fill<<<dim3(uint32_t(blocks)), dim3(kHfWaves * 64), smem, s>>>(eng, h.tab, h.hid2sid, h.n_hot, h.first_match,
                                                                         h.start, g, active, totals, cap, aoff, out);
    return hipGetLastError();
}

}  // namespace acgpu
