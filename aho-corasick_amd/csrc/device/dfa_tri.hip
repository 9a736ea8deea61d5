// DFA transition walk with an exact skip of the depth <= 2 regime (k_tri_walk<DfaTriDev, DfaTriWalk>): the reference's
// per-byte primitive sid = trans[sid + classes[byte]] (src/dfa.rs:218-226) inside the overlapping loop
// (src/automaton.rs:1491-1534), one haystack lane-chunk per wavefront lane.
//
// The global-table walk (k_walk_count<DfaEng>, kernels.hip) pays one dependent L2 gather per haystack byte: 6.4 % of the
// HBM peak on the headline workload since round 1.  Almost all of those gathers only confirm that the state stays near
// the root: with 1 000 random patterns 99.9 % of the steps start and end in a state of depth <= 2.  While the depth is
// <= 2 the state is a function of the last two bytes, and the next byte leaves that regime exactly when the trigram is
// a trie node -- one bit per (pair of classes, class) in LDS (host/dfa_tri_tables.cpp, host/cnfa_tri_tables.cpp).  So a
// lane scans 16 bytes branch-free into a candidate mask (tri_common.hpp), jumps from candidate to candidate, and only
// below depth 2 takes the reference's step through the table in global memory -- until a transition lands on a target
// the table's device copy tags as "depth <= 2".  Matches are recorded as events by the count pass (records of matches
// of <= 2 bytes from a per-pair table), so nothing is walked twice.  Serves the single-start unanchored layout; the
// two-start layout of StartKind::Both keeps k_walk_count (its unanchored searches run on the single-start twin anyway).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "dfa_tri.hpp"
#include "launch_util.hpp"

namespace acgpu {

namespace {

// One thread per event slot: the records of the event's state at out[offsets[ci] + pre ...] (dfa.rs:275-286).
__global__ __launch_bounds__(256) void k_dfa_tri_emit(DfaTriDev t, DfaEng eng, ScanGeom g, TriEvents evs,
                                                      const uint64_t* __restrict__ offsets, const uint64_t* __restrict__ totals,
                                                      uint64_t cap, acgpu_match* __restrict__ out) {
    if (evs.ctr[1] != 0 || totals[0] > cap) return;
    const uint64_t nseg = evs.ctr[0] < evs.max_segs ? evs.ctr[0] : evs.max_segs;
    for (uint64_t s = uint64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); s < nseg; s += uint64_t(gridDim.x) * 4) {
        const uint32_t i = threadIdx.x & 63;
        if (i >= evs.seg_fill[s]) continue;
        const TriEvent e = evs.ev[s * kTriSeg + i];
        const uint32_t sid = (e.state & 0x80000000u) ? t.st2[e.state & 0x7FFFFFFFu] : e.state;
        // (rel = -1 as unsigned: the empty pattern at the start of the search)
        const uint64_t end = g.grid0 + uint64_t(e.ci) * g.chunk + uint64_t(int64_t(int32_t(e.rel))) + 1 - g.base_mis;
        acgpu_match* dst = out + offsets[e.ci] + e.pre;
        const uint32_t n = eng.match_len(sid);
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t pid = eng.match_pattern(sid, k);
            acgpu_match m; m.pattern = pid; m._pad = 0; m.end = end; m.start = end - eng.pattern_len(pid);
            dst[k] = m;
        }
    }
}

}  // namespace

hipError_t build_dfa_tri(const NNfa& n, const Dfa& d, const uint32_t* dev_moff, DfaTriTables& out) {
    out.ready = false;
    DfaTriHost t;
    if (!build_dfa_tri_host(n, d, t)) return hipSuccess;
    hipError_t e;
    if ((e = out.b_bits.upload(t.bits)) != hipSuccess) return e;
    if ((e = out.b_base.upload(t.base)) != hipSuccess) return e;
    if ((e = out.b_uc.upload(t.uc)) != hipSuccess) return e;
    if ((e = out.b_inv.upload(t.inv)) != hipSuccess) return e;
    if ((e = out.b_child.upload(t.child)) != hipSuccess) return e;
    if ((e = out.b_trans3.upload(t.trans3)) != hipSuccess) return e;
    if (t.shallow_matches) {
        if ((e = out.b_mc2.upload(t.mc2)) != hipSuccess) return e;
        if ((e = out.b_st2.upload(t.st2)) != hipSuccess) return e;
    }
    DfaTriDev& dv = out.dev;
    dv.bits = out.b_bits.as<uint32_t>(); dv.base = out.b_base.as<uint16_t>(); dv.uc = out.b_uc.as<uint8_t>(); dv.inv = out.b_inv.as<uint8_t>();
    dv.mc2 = out.b_mc2.as<uint8_t>(); dv.st2 = out.b_st2.as<uint32_t>();
    dv.child = out.b_child.as<uint32_t>(); dv.trans3 = out.b_trans3.as<uint32_t>(); dv.moff = dev_moff;
    dv.pairs = t.apair * t.apair; dv.apair = t.apair; dv.bw = t.bw; dv.n_used = t.n_used;
    dv.gshift = 0;
    while ((1u << dv.gshift) < t.granule) dv.gshift++;
    dv.shallow_matches = t.shallow_matches ? 1u : 0u;
    dv.start_mlen = t.start_mlen;
    dv.stride2 = uint32_t(d.stride2);
    dv.max_match_id = d.special.max_match_id;
    dv.trans_words = uint32_t(t.trans3.size());
    dv.n_child = uint32_t(t.child.size());
    out.lds_bytes = t.lds_bytes;
    out.ready = true;
    return hipSuccess;
}

hipError_t launch_dfa_tri_count(const DfaTriTables& h, const ScanGeom& g, uint32_t* counts, const TriEvents* evs, hipStream_t s) {
    if (!h.ready) return hipErrorInvalidValue;
    const uint64_t blocks = (g.n_chunks + kTriBlock - 1) / kTriBlock;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_tri_walk<DfaTriDev, DfaTriWalk>), int(kTriLdsBudget)); e != hipSuccess) return e;
    k_tri_walk<DfaTriDev, DfaTriWalk><<<dim3(uint32_t(blocks)), dim3(kTriBlock), h.lds_bytes, s>>>(h.dev, g, counts, evs ? *evs : TriEvents(), 0u);
    return hipGetLastError();
}

hipError_t launch_dfa_tri_emit(const DfaTriTables& h, const DevAutomaton& a, const ScanGeom& g, const TriEvents& evs,
                               const uint64_t* offsets, const uint64_t* totals, uint64_t cap, acgpu_match* out, hipStream_t s) {
    if (!h.ready || !evs.ev) return hipErrorInvalidValue;
    DfaEng eng; eng.d = a.dfa; eng.cls = a.dfa.classes;
    const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(evs.max_segs) + 3) / 4, uint64_t(device_cus()) * 32));
    k_dfa_tri_emit<<<dim3(blocks), dim3(256), 0, s>>>(h.dev, eng, g, evs, offsets, totals, cap, out);
    return hipGetLastError();
}

}  // namespace acgpu
