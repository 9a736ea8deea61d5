// GPU-side fill of the DFA transition table.  The reference fills one row per noncontiguous-NFA
// state by asking the NFA for next_state(byte) of every class (src/dfa.rs:544-607, :801-835); because
//     row(s) = row(fail(s)) overridden by the explicit transitions of s
// and fail(s) is strictly shallower than s, all states of one trie depth are independent given the previous depth:
// one launch per depth, one thread per (state, class).  Produces the premultiplied table of DFA::build for
// StartKind::Unanchored / Anchored (state index == nNFA id, src/dfa.rs:553-560); the table is built in HBM and copied
// back once so that the host-side consumers (hot tables, introspection) see the same words the CPU path produces.
#include <hip/hip_runtime.h>

#include <vector>

#include "../host/automaton.hpp"
#include "dfa_fill.hpp"

namespace acgpu {

namespace {

struct FillArgs {
    const uint32_t* order;   // states of this depth (nNFA ids)
    uint32_t count;
    const uint32_t* toff;
    const uint8_t* tbyte;
    const uint32_t* tnext;
    const uint32_t* fail;
    const uint8_t* cls;      // [256]
    const uint8_t* rep;      // [256] 1 = first byte of its class (dfa.rs:801-835 asks once per class)
    uint32_t* trans;
    uint32_t s2, alen, anchored;
};

__global__ __launch_bounds__(256) void k_dfa_fill_level(FillArgs a) {
    const uint64_t idx = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    const uint64_t si = idx >> a.s2;
    const uint32_t k = uint32_t(idx) & ((1u << a.s2) - 1);
    if (si >= a.count || k >= a.alen) return;
    const uint32_t s = a.order[si];
    const uint32_t f = a.fail[s];
    // noncontiguous::NFA::next_state (noncontiguous.rs:601-626): follow fail links until a transition exists;
    // the fail target's row already holds that answer.  Anchored rows stop at the first failure (DEAD).
    uint32_t v = (!a.anchored && f != kDead) ? a.trans[(size_t(f) << a.s2) + k] : kDead;
    for (uint32_t t = a.toff[s]; t < a.toff[s + 1]; t++) {
        const uint8_t b = a.tbyte[t];
        if (!a.rep[b] || a.cls[b] != k) continue;
        const uint32_t nx = a.tnext[t];
        if (nx == kFail) { if (a.anchored || f == kDead) v = kDead; }
        else v = nx << a.s2;
    }
    a.trans[(size_t(s) << a.s2) + k] = v;
}

template <class T> hipError_t up(const T* src, size_t n, T** dst) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(n * sizeof(T), 16));
    if (e != hipSuccess) return e;
    return n ? hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
}

}  // namespace

hipError_t device_fill_dfa(const NNfa& n, const uint8_t* classes, size_t alen, size_t s2, bool anchored,
                           uint32_t* host_trans) {
    const size_t N = n.states();
    uint8_t rep[256];
    for (int b = 0; b < 256; b++) rep[b] = (b == 0) || classes[b] != classes[b - 1];
    // depth levels of the breadth-first order: the two start states first, then by stored depth (= distance - 1)
    std::vector<uint32_t> level_begin;
    for (size_t i = 0; i < n.bfs.size(); i++) {
        const bool new_level = i == 0 || i == 2 || (i > 2 && n.depth[n.bfs[i]] != n.depth[n.bfs[i - 1]]);
        if (new_level) level_begin.push_back(uint32_t(i));
    }
    level_begin.push_back(uint32_t(n.bfs.size()));

    uint32_t *d_order = nullptr, *d_toff = nullptr, *d_tnext = nullptr, *d_fail = nullptr, *d_trans = nullptr;
    uint8_t *d_tbyte = nullptr, *d_cls = nullptr, *d_rep = nullptr;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t r) {
        for (void* p : {static_cast<void*>(d_order), static_cast<void*>(d_toff), static_cast<void*>(d_tnext),
                        static_cast<void*>(d_fail), static_cast<void*>(d_trans), static_cast<void*>(d_tbyte),
                        static_cast<void*>(d_cls), static_cast<void*>(d_rep)})
            if (p) (void)hipFree(p);
        return r;
    };
    if ((e = up(n.bfs.data(), n.bfs.size(), &d_order)) != hipSuccess) return done(e);
    if ((e = up(n.toff.data(), n.toff.size(), &d_toff)) != hipSuccess) return done(e);
    if ((e = up(n.tbyte.data(), n.tbyte.size(), &d_tbyte)) != hipSuccess) return done(e);
    if ((e = up(n.tnext.data(), n.tnext.size(), &d_tnext)) != hipSuccess) return done(e);
    if ((e = up(n.fail.data(), n.fail.size(), &d_fail)) != hipSuccess) return done(e);
    if ((e = up(classes, 256, &d_cls)) != hipSuccess) return done(e);
    if ((e = up(rep, 256, &d_rep)) != hipSuccess) return done(e);
    const size_t words = N << s2;
    if ((e = hipMalloc(reinterpret_cast<void**>(&d_trans), std::max<size_t>(words * 4, 16))) != hipSuccess) return done(e);
    if ((e = hipMemset(d_trans, 0, words * 4)) != hipSuccess) return done(e);   // DEAD and FAIL rows stay all-DEAD
    for (size_t l = 0; l + 1 < level_begin.size(); l++) {
        const uint32_t lo = level_begin[l], cnt = level_begin[l + 1] - lo;
        if (!cnt) continue;
        FillArgs a{d_order + lo, cnt, d_toff, d_tbyte, d_tnext, d_fail, d_cls, d_rep, d_trans,
                   uint32_t(s2), uint32_t(alen), anchored ? 1u : 0u};
        const uint64_t threads = uint64_t(cnt) << s2;
        k_dfa_fill_level<<<dim3(uint32_t((threads + 255) / 256)), dim3(256)>>>(a);
        if ((e = hipGetLastError()) != hipSuccess) return done(e);
    }
    e = hipMemcpy(host_trans, d_trans, words * 4, hipMemcpyDeviceToHost);
    return done(e);
}

}  // namespace acgpu
