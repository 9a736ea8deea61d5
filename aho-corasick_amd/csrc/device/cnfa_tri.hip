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
#include "tri_kernel.hpp"
#include "launch_util.hpp"

namespace acgpu {

namespace {

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
    constexpr bool one_lane = false;   // (debug form of the kernel: one lane per wavefront walks)
    const uint64_t blocks = one_lane ? (g.n_chunks + 15) / 16 : (g.n_chunks + kTriBlock - 1) / kTriBlock;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_tri_walk<CnfaTriDev, TriWalk>), int(kTriLdsBudget)); e != hipSuccess) return e;
    k_tri_walk<CnfaTriDev, TriWalk><<<dim3(uint32_t(blocks)), dim3(kTriBlock), h.lds_bytes, s>>>(h.dev, g, counts, evs ? *evs : TriEvents(), one_lane ? 1u : 0u);
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
