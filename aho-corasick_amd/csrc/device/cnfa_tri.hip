// Contiguous-NFA failure-link walk with an exact skip of the depth <= 2 regime (k_cnfa_tri): the reference's
// contiguous::NFA::next_state (src/nfa/contiguous.rs:186-247) over the reference's own `repr` words, one haystack
// lane-chunk per wavefront lane.
//
// What bounds the literal walk on gfx950 (profiles/r03_cnfa_v2_pmc.json: 1.14 L1->L2 requests per haystack byte, waves
// waiting 73 % of their cycles, the L1's miss queue full 81 % of the time): every step is a dependent gather, and a
// CU sustains only ~0.26 of them per cycle.  89 % of those steps start in a state of depth 2, fail to depth 1 and land
// in another state of depth 2 -- and need no table at all: while its depth is <= 2 the state is the longest suffix of
// the text that is a trie node of depth <= 2, a function of the last two bytes, and the next byte c leaves that regime
// exactly when the trigram (a, b, c) is a trie node (host/cnfa_tri_tables.cpp).  So
//   * LDS holds one bit per (pair of classes, class) -- 108 KiB at 95 x 95 x 95 -- and a step whose bit is clear is
//     finished after that one LDS read (the state is implied by the bytes; the records of matches of <= 2 bytes, if the
//     set has any, come from a per-pair table);
//   * a set bit is the transition depth 2 -> 3: ONE 16-byte gather from a table of the depth-3 nodes (indexed by the
//     pair's base + the rank of the bit) brings the state together with the head of its record {header, fail, first
//     data word}, so the following byte is decided from registers: 99 % of the time "no transition, fail" -- and a fail
//     word that names a state of depth <= 2 (tagged at upload) ends the excursion with no further access;
//   * below depth 3 the walk is the literal one: state record (16 bytes, one gather), class compare, failure link.
// 0.127 gathers per byte instead of 1.14 at 100 000 patterns (counted by the CPU model of this walk,
// tests/test_cnfa_tables.py).  Lanes are byte-synchronous; a byte costs the wavefront one gather round (some lane
// always needs one) issued for all needing lanes together.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "cnfa_tri.hpp"
#include "cnfa_tri_step.hpp"
#include "launch_util.hpp"

namespace acgpu {

namespace {

#ifndef TRI_BLOCK
#define TRI_BLOCK 1024
#endif
constexpr int kTriBlock = TRI_BLOCK;

__device__ __forceinline__ unsigned long long* tri_guard(const ScanGeom& g) {
#ifdef ACGPU_GUARD
    return g.guard;
#else
    return nullptr;
#endif
}

__global__ __launch_bounds__(kTriBlock, 1) void k_cnfa_tri(CnfaTriDev t, ScanGeom g, uint32_t* __restrict__ counts, TriEvents evs, uint32_t one_lane) {
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
    TriWalk f{s_bits, s_base, s_uc, s_inv, s_mc2, s_lane + threadIdx.x * 16, t.child, t.repr3, t.apair, t.bw, t.gshift, t.n_used,
              t.alen, t.max_match_id, t.shallow_matches, t.repr_words, t.n_child, tri_guard(g),
              MD_SHALLOW, 0u, 0u, 0u, 0u, 0u, 0u, t.n_used, t.n_used, t.n_used, t.n_used, 0u, 0u, 0u, 0u, 0u,
              evs.ev, evs.seg_fill, evs.ctr, evs.max_segs, uint32_t(ci), 0xFFFFFFFFu, kTriSeg, 0u, 0u, 0u, 0u};
    if (valid && ci == 0 && g.emit_start_matches && t.start_mlen) {   // the empty pattern at the start of the search
        f.ev_has = 1; f.ev_state = 0x80000000u | (t.n_used * t.apair + t.n_used); f.ev_idx = 0; f.ev_pre = 0;
        f.cnt += t.start_mlen;
    }
    // positions relative to the 64-byte sector the lane's walk starts in: wave-uniform offsets, per-lane bounds
    const uint64_t p0 = r.w & ~uint64_t(63);
    const int32_t w_rel = int32_t(r.w - p0), lo_rel = int32_t(r.lo - p0), hi_rel = valid ? int32_t(r.hi - p0) : 0;
    // events carry positions relative to the chunk's grid origin; the start-of-search event sits one byte in front of it
    const int32_t org_rel = int32_t(int64_t(g.grid0 + ci * uint64_t(g.chunk)) - int64_t(p0));
    f.flush_events(int32_t(int64_t(g.cold_floor) - 1 - int64_t(g.grid0)));
    for (int32_t s0 = 0; ACGPU_TRI_ANY(s0 < hi_rel); s0 += 64) {
        // the sector in registers (a 128-byte line is requested twice, back to back halves; nothing else of it is kept)
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
        const uint4 c0 = piece(0), c1 = piece(1), c2 = piece(2), c3 = piece(3);   // (named registers: an array goes to scratch memory)
#pragma unroll 1
        for (int32_t q = 0; q < 4; q++) {   // (one copy of the piece code in the instruction stream)
            const uint4 dq = q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3));
            const uint32_t wds[4] = {dq.x, dq.y, dq.z, dq.w};
            const int32_t pv = s0 + 16 * q;
            auto clamp16 = [](int32_t x) -> uint32_t { return uint32_t(x < 0 ? 0 : (x > 16 ? 16 : x)); };
            const uint32_t lo_i = clamp16(w_rel - pv), hi_i = clamp16(hi_rel - pv), own_from = clamp16(lo_rel - pv);
            const uint32_t act16 = ((1u << hi_i) - 1u) & ~((1u << lo_i) - 1u);
            if (!ACGPU_TRI_ANY(act16 != 0xFFFFu)) f.piece_scan<true>(wds, act16);
            else f.piece_scan<false>(wds, act16);
            f.piece_walk(hi_i, own_from, pv - org_rel);
        }
    }
    f.finish_events();
    if (valid) counts[ci] = f.cnt;
}

// One thread per event slot: the records of the event's state at out[offsets[ci] + pre ...] (contiguous.rs:611-633).
__global__ __launch_bounds__(256) void k_cnfa_tri_emit(CnfaTriDev t, const uint32_t* __restrict__ plens, ScanGeom g, TriEvents evs,
                                                       const uint64_t* __restrict__ offsets, const uint64_t* __restrict__ totals,
                                                       uint64_t cap, acgpu_match* __restrict__ out) {
    if (evs.ctr[1] != 0 || totals[0] > cap) return;
    const uint64_t nseg = evs.ctr[0] < evs.max_segs ? evs.ctr[0] : evs.max_segs;
    for (uint64_t s = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); s < nseg; s += uint64_t(gridDim.x) * 4) {
        const uint32_t i = threadIdx.x & 63;
        if (i >= evs.seg_fill[s]) continue;
        const TriEvent e = evs.ev[s * kTriSeg + i];
        const uint32_t st = (e.state & 0x80000000u) ? t.st2[e.state & 0x7FFFFFFFu] : e.state;
        const uint32_t kind = t.repr3[st] & 0xFFu;
        const uint32_t base = st + (kind == 0xFFu ? 2 + t.alen : 2 + ((kind + 3) >> 2) + kind);
        const uint32_t packed = t.repr3[base];
        // (rel = -1 as unsigned: the empty pattern at the start of the search)
        const uint64_t end = g.grid0 + uint64_t(e.ci) * g.chunk + uint64_t(int64_t(int32_t(e.rel))) + 1 - g.base_mis;
        acgpu_match* dst = out + offsets[e.ci] + e.pre;
        const uint32_t n = (packed & (1u << 31)) ? 1u : packed;
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t pid = (packed & (1u << 31)) ? (packed & 0x7FFFFFFFu) : t.repr3[base + 1 + k];
            acgpu_match m; m.pattern = pid; m._pad = 0; m.end = end; m.start = end - plens[pid];
            dst[k] = m;
        }
    }
}

}  // namespace

hipError_t build_cnfa_tri(const CNfa& c, CnfaTriTables& out) {
    out.ready = false;
    CnfaTriHost t;
    if (!build_cnfa_tri_host(c, t)) return hipSuccess;
    hipError_t e;
    if ((e = out.b_bits.upload(t.bits)) != hipSuccess) return e;
    if ((e = out.b_base.upload(t.base)) != hipSuccess) return e;
    if ((e = out.b_uc.upload(t.uc)) != hipSuccess) return e;
    if ((e = out.b_inv.upload(t.inv)) != hipSuccess) return e;
    if ((e = out.b_child.upload(t.child)) != hipSuccess) return e;
    if ((e = out.b_repr3.upload(t.repr3)) != hipSuccess) return e;
    if (t.shallow_matches) {
        if ((e = out.b_mc2.upload(t.mc2)) != hipSuccess) return e;
        if ((e = out.b_st2.upload(t.st2)) != hipSuccess) return e;
    }
    CnfaTriDev& d = out.dev;
    d.bits = out.b_bits.as<uint32_t>(); d.base = out.b_base.as<uint16_t>(); d.uc = out.b_uc.as<uint8_t>(); d.inv = out.b_inv.as<uint8_t>();
    d.mc2 = out.b_mc2.as<uint8_t>(); d.st2 = out.b_st2.as<uint32_t>();
    d.child = out.b_child.as<TriChild>(); d.repr3 = out.b_repr3.as<uint32_t>();
    d.pairs = t.apair * t.apair; d.apair = t.apair; d.bw = t.bw; d.n_used = t.n_used;
    d.gshift = 0;
    while ((1u << d.gshift) < t.granule) d.gshift++;
    d.shallow_matches = t.shallow_matches ? 1u : 0u;
    d.start_mlen = t.start_mlen;
    d.alen = uint32_t(c.alphabet_len);
    d.max_match_id = c.special.max_match_id;
    d.repr_words = uint32_t(t.repr3.size());
    d.n_child = uint32_t(t.child.size());
    out.lds_bytes = t.lds_bytes;
    out.ready = true;
    return hipSuccess;
}

hipError_t launch_cnfa_tri_count(const CnfaTriTables& h, const ScanGeom& g, uint32_t* counts, const TriEvents* evs, hipStream_t s) {
    if (!h.ready) return hipErrorInvalidValue;
    static const bool one_lane = std::getenv("ACGPU_TRI_ONE_LANE") != nullptr;   // debug knob
    const uint64_t blocks = one_lane ? (g.n_chunks + 15) / 16 : (g.n_chunks + kTriBlock - 1) / kTriBlock;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_cnfa_tri), int(kTriLdsBudget)); e != hipSuccess) return e;
    k_cnfa_tri<<<dim3(uint32_t(blocks)), dim3(kTriBlock), h.lds_bytes, s>>>(h.dev, g, counts, evs ? *evs : TriEvents(), one_lane ? 1u : 0u);
    return hipGetLastError();
}

}  // namespace acgpu

namespace acgpu {
hipError_t launch_cnfa_tri_emit(const CnfaTriTables& h, const uint32_t* plens, const ScanGeom& g, const TriEvents& evs,
                                const uint64_t* offsets, const uint64_t* totals, uint64_t cap, acgpu_match* out, hipStream_t s) {
    if (!h.ready || !evs.ev) return hipErrorInvalidValue;
    const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(evs.max_segs) + 3) / 4, uint64_t(device_cus()) * 32));
    k_cnfa_tri_emit<<<dim3(blocks), dim3(256), 0, s>>>(h.dev, plens, g, evs, offsets, totals, cap, out);
    return hipGetLastError();
}
}  // namespace acgpu
