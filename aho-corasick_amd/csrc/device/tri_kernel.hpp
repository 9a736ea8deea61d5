// The scan kernel both shallow-skip walks share (k_tri_walk<Dev, Walk>: cnfa_tri.hip instantiates it for the
// contiguous-NFA failure-link walk, dfa_tri.hip for the DFA transition walk).  Dev: the device tables -- the common
// members (bits, base, uc, inv, mc2, pairs, apair, bw, gshift, n_used, shallow_matches, start_mlen, n_child) and
// setup(Walk&) for the walk's own.  One haystack lane-chunk per wavefront lane, read in whole 128-byte lines, walked in
// 16-byte pieces (tri_common.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "engines.hpp"
#include "tri_common.hpp"

namespace acgpu {

#ifndef TRI_BLOCK
#define TRI_BLOCK 1024
#endif
constexpr int kTriBlock = TRI_BLOCK;

// Event buffer of one scan (count pass -> emit kernel).
struct TriEvents {
    TriEvent* ev = nullptr;              // [max_segs * kTriSeg]
    uint32_t* seg_fill = nullptr;        // [max_segs]
    unsigned long long* ctr = nullptr;   // [0] segments handed out, [1] overflow flag (zeroed before the count pass)
    uint32_t max_segs = 0;
};
inline uint32_t tri_event_segments(uint64_t span_bytes) {   // one event per 64 haystack bytes, 64 Ki to 12 Mi events
    const uint64_t ev = span_bytes / 64 < (uint64_t(1) << 16) ? (uint64_t(1) << 16) : (span_bytes / 64 > (uint64_t(12) << 20) ? (uint64_t(12) << 20) : span_bytes / 64);
    return uint32_t(ev / kTriSeg);
}

__device__ __forceinline__ unsigned long long* tri_guard(const ScanGeom& g) {
#ifdef ACGPU_GUARD
    return g.guard;
#else
    return nullptr;
#endif
}

template <class Dev, class Walk>
__global__ __launch_bounds__(kTriBlock, 1) void k_tri_walk(Dev t, ScanGeom g, uint32_t* __restrict__ counts, TriEvents evs, uint32_t one_lane) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* s_lane = smem;                                                   // [kTriBlock][16]: the piece at hand, per lane
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(smem + kTriBlock * 16);
    uint16_t* s_base = reinterpret_cast<uint16_t*>(s_bits + size_t(t.pairs) * t.bw);
    uint8_t* s_mc2 = reinterpret_cast<uint8_t*>(s_base + t.pairs);
    uint8_t* s_uc = s_mc2 + (t.shallow_matches ? t.pairs : 0);
    uint8_t* s_inv = s_uc + 256;
    for (uint32_t i = threadIdx.x; i < t.pairs * t.bw; i += kTriBlock) s_bits[i] = t.bits[i];
    for (uint32_t i = threadIdx.x; i < t.pairs; i += kTriBlock) s_base[i] = t.base[i];
    if (t.shallow_matches) for (uint32_t i = threadIdx.x; i < t.pairs; i += kTriBlock) s_mc2[i] = t.mc2[i];
    for (uint32_t i = threadIdx.x; i < 256; i += kTriBlock) { s_uc[i] = t.uc[i]; s_inv[i] = t.inv[i]; }
    __syncthreads();

    // (debug knob one_lane: only lane 0 of every wavefront walks a chunk -- the wave-level votes then see one lane)
    const uint64_t ci = one_lane ? (uint64_t(blockIdx.x) * kTriBlock + threadIdx.x) >> 6 : uint64_t(blockIdx.x) * kTriBlock + threadIdx.x;
    const bool valid = ci < g.n_chunks && (!one_lane || (threadIdx.x & 63) == 0);
    ChunkRange r{0, 0, 0};
    if (valid) r = chunk_range(g, ci);
    Walk f;
    f.s_bits = s_bits; f.s_base = s_base; f.s_uc = s_uc; f.s_inv = s_inv; f.s_mc2 = s_mc2; f.s_buf = s_lane + threadIdx.x * 16;
    f.A = t.apair; f.bw = t.bw; f.gshift = t.gshift; f.U = t.n_used; f.sm = t.shallow_matches; f.n_child = t.n_child;
    f.guard = tri_guard(g);
    f.ua = f.ub = f.na = f.nb = t.n_used;
    t.setup(f);
    f.ev_buf = evs.ev; f.ev_seg_fill = evs.seg_fill; f.ev_ctr = evs.ctr; f.ev_max_segs = evs.max_segs; f.ci = uint32_t(ci);
    if (valid && ci == 0 && g.emit_start_matches && t.start_mlen) {   // the empty pattern at the start of the search
        f.note_event(0x80000000u | (t.n_used * t.apair + t.n_used), 0, t.start_mlen);
    }
    // positions relative to the 128-byte line the lane's walk starts in: wave-uniform offsets, per-lane bounds
    const uint64_t p0 = r.w & ~uint64_t(127);
    const int32_t w_rel = int32_t(r.w - p0), lo_rel = int32_t(r.lo - p0), hi_rel = valid ? int32_t(r.hi - p0) : 0;
    // events carry positions relative to the chunk's grid origin; the start-of-search event sits one byte in front of it
    const int32_t org_rel = int32_t(int64_t(g.grid0 + ci * uint64_t(g.chunk)) - int64_t(p0));
    f.flush_events(int32_t(int64_t(g.cold_floor) - 1 - int64_t(g.grid0)));
    for (int32_t s0 = 0; ACGPU_TRI_ANY(s0 < hi_rel); s0 += 128) {
        // one whole cache line in registers: its eight 16-byte loads are issued back to back, so the line is fetched once
        // (with 64-byte sectors every line was requested twice, microseconds apart, and came over the fabric twice:
        // profiles/r03_tri_walk_pmc.json, 19.0 GB of reads for an 8.6 GB haystack)
        auto piece = [&](int32_t q) -> uint4 {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int32_t pv = s0 + 16 * q;
            if (pv + 16 > w_rel && pv < hi_rel) {
                ACGPU_HAY_CHECK(g, p0 + pv, 16);
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                const v4u x = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(g.hay16 + p0 + pv));
                v = make_uint4(x.x, x.y, x.z, x.w);
            }
            return v;
        };
        // (named registers: an array goes to scratch memory)
        const uint4 c0 = piece(0), c1 = piece(1), c2 = piece(2), c3 = piece(3), c4 = piece(4), c5 = piece(5), c6 = piece(6), c7 = piece(7);
#pragma unroll 1
        for (int32_t q = 0; q < 8; q++) {   // (one copy of the piece code in the instruction stream)
            auto sel = [&](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7) {
                const uint32_t lo = (q & 2) ? ((q & 1) ? a3 : a2) : ((q & 1) ? a1 : a0);
                const uint32_t hi = (q & 2) ? ((q & 1) ? a7 : a6) : ((q & 1) ? a5 : a4);
                return (q & 4) ? hi : lo;
            };
            const uint32_t wds[4] = {sel(c0.x, c1.x, c2.x, c3.x, c4.x, c5.x, c6.x, c7.x), sel(c0.y, c1.y, c2.y, c3.y, c4.y, c5.y, c6.y, c7.y),
                                     sel(c0.z, c1.z, c2.z, c3.z, c4.z, c5.z, c6.z, c7.z), sel(c0.w, c1.w, c2.w, c3.w, c4.w, c5.w, c6.w, c7.w)};
            const int32_t pv = s0 + 16 * q;
            auto clamp16 = [](int32_t x) -> uint32_t { return uint32_t(x < 0 ? 0 : (x > 16 ? 16 : x)); };
            const uint32_t lo_i = clamp16(w_rel - pv), hi_i = clamp16(hi_rel - pv), own_from = clamp16(lo_rel - pv);
            const uint32_t act16 = ((1u << hi_i) - 1u) & ~((1u << lo_i) - 1u);
            if (t.shallow_matches) {
                if (!ACGPU_TRI_ANY(act16 != 0xFFFFu)) f.template piece_scan<true, true>(wds, act16);
                else f.template piece_scan<false, true>(wds, act16);
            } else {
                if (!ACGPU_TRI_ANY(act16 != 0xFFFFu)) f.template piece_scan<true, false>(wds, act16);
                else f.template piece_scan<false, false>(wds, act16);
            }
            f.piece_walk(hi_i, own_from, pv - org_rel);
        }
    }
    f.finish_events();
    if (valid) counts[ci] = f.cnt;
}

}  // namespace acgpu
