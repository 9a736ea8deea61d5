// What the two shallow-skip walks share (k_cnfa_tri: contiguous-NFA failure-link walk, cnfa_tri_step.hpp; k_dfa_tri: DFA
// transition walk, dfa_tri_step.hpp): the tables of the depth <= 2 regime, the branch-free scan of a 16-byte piece, the
// jump to the next candidate, and the match events.  Shared with the host: the test hooks run THIS code lane by lane on
// the CPU (tests/test_cnfa_tri_tables.py, tests/test_dfa_tri_tables.py).
#pragma once
#include <stdint.h>

#include "../host/cnfa_tri_tables.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ACGPU_TRI_FN __host__ __device__ __forceinline__
#else
#define ACGPU_TRI_FN inline
#endif
// Lane flags are 32-bit values and loop conditions wave-uniform scalars.  (Round 3 blamed a miscount on `bool` lane masks;
// round 4 rebuilt the kernels with `bool` flags and plain __any() conditions -- exact over 18 runs,
// profiles/r04_tri_bool_flavour.jsonl -- so the experiment flavour is gone; the shape below is simply the faster one.)
typedef uint32_t tri_flag;
#if defined(__HIP_DEVICE_COMPILE__)
#define ACGPU_TRI_ANY(x) (__builtin_amdgcn_readfirstlane(int(__ballot(x) != 0)) != 0)   // wave-uniform, and known to be
#define ACGPU_TRI_MUL24(a, b) __umul24(a, b)   // full-rate 24-bit multiply: every index here is far below 2^24
#else
#define ACGPU_TRI_ANY(x) (x)   // one lane at a time on the host
#define ACGPU_TRI_MUL24(a, b) ((a) * (b))
#endif

// Bounds-checked debug flavour (make guard): every table access of the walk is checked; a violation is counted
// (acgpu_guard_violations), the first few are printed, and the access is redirected to word 0.
#if defined(ACGPU_GUARD) && defined(__HIP_DEVICE_COMPILE__)
#define ACGPU_TRI_BOUND(idx, limit, what)                                                                              \
    do {                                                                                                               \
        if ((idx) >= (limit)) {                                                                                        \
            if (guard && atomicAdd(guard, 1ull) < 8)                                                                   \
                printf("k_cnfa_tri: %s index %u >= %u (md %u o %u head %08x fail %08x pr %u)\n", what, unsigned(idx),  \
                       unsigned(limit), md, o, head, fail, pr);                                                        \
            (idx) = 0;                                                                                                 \
        }                                                                                                              \
    } while (0)
#else
#define ACGPU_TRI_BOUND(idx, limit, what) ((void)0)
#endif

namespace acgpu {

typedef uint32_t tri_u32x4 __attribute__((ext_vector_type(4), aligned(4)));

// One lane's view of the depth <= 2 regime.  The haystack is consumed in 16-byte pieces:
//   piece_scan    branch-free, the same for every lane: compact classes of the 16 bytes (kept in this lane's 16 bytes of
//                 LDS for the walk) and the candidate mask: bit i = "if the automaton's state has depth <= 2 in front
//                 of byte i, byte i leaves that regime (the trigram is a trie node) or ends a match of <= 2 bytes";
//   shallow_jump  the lane's next candidate at or behind `pos` (ctz), with the pair of classes in front of it.
// All lane flags are 32-bit values, not bool: carried around the loops as lane masks they came out wrong on the
// device (identical source; one lane per wavefront: right, 64 lanes: 1e-4 of the counts off).
struct TriLane {
    const uint32_t* s_bits = nullptr;
    const uint16_t* s_base = nullptr;
    const uint8_t* s_uc = nullptr;      // [256] byte -> compact class (U: the byte labels no trie edge)
    const uint8_t* s_inv = nullptr;     // [256] compact class -> the automaton's class
    const uint8_t* s_mc2 = nullptr;
    uint8_t* s_buf = nullptr;           // this lane's 16 bytes of LDS: the compact classes of the piece at hand
    uint32_t A = 0, bw = 0, gshift = 0, U = 0;
    tri_flag sm = 0;                    // wave-uniform: some state of depth <= 2 is a match state
    uint32_t n_child = 0;               // table size (bounds-checked flavour only)
    unsigned long long* guard = nullptr;
    uint32_t cnt = 0;
    uint32_t ua = 0, ub = 0;            // compact classes of the two bytes in front of the piece at hand
    uint32_t na = 0, nb = 0;            // ... of its last two bytes (piece_scan)
    uint32_t cand = 0, pos = 0;
    uint32_t pr = 0;                    // (diagnostics of the bounds-checked flavour)
    // match events (optional: ev_buf == nullptr counts only): one per match state entered at an owned position --
    // {chunk, records of the chunk in front of it, state, position} -- appended to wave-private segments of kTriSeg
    // events, so that the emit kernel can write the ordered records without walking anything again
    TriEvent* ev_buf = nullptr;
    uint32_t* ev_seg_fill = nullptr;
    unsigned long long* ev_ctr = nullptr;   // [0] segments handed out, [1] overflow flag
    uint32_t ev_max_segs = 0, ci = 0;
    static constexpr uint32_t kTriSegDead = 0xFFFFFFFEu;
    uint32_t wseg = 0xFFFFFFFFu, wused = kTriSeg;   // wave-uniform: the wavefront's current segment and its fill
    tri_flag ev_has = 0;
    uint32_t ev_state = 0, ev_idx = 0, ev_pre = 0;

    ACGPU_TRI_FN void note_event(uint32_t state, uint32_t idx, uint32_t records) {
        ev_has = 1; ev_state = state; ev_idx = idx; ev_pre = cnt;
        cnt += records;
    }
    // End of a trip (wave-uniform control flow): the lanes that counted a match append their event.
    ACGPU_TRI_FN void flush_events(int32_t rel0) {
        if (!ev_buf) { ev_has = 0; return; }
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned long long mask = __ballot(ev_has != 0);
        if (mask == 0) return;
        if (wseg == kTriSegDead) { ev_has = 0; return; }   // the buffer overflowed earlier: the caller's fallback takes over
        const uint32_t n = uint32_t(__popcll(mask));
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (wused + n > kTriSeg) {   // (also the first event of the wavefront: wused starts at kTriSeg)
            uint32_t ns = 0;
            if (lane == 0) {
                if (wseg < ev_max_segs) ev_seg_fill[wseg] = wused;
                ns = uint32_t(atomicAdd(ev_ctr, 1ull));
                if (ns >= ev_max_segs) ev_ctr[1] = 1ull;   // more events than the buffer holds (ONE store per wavefront:
            }                                              // every lane storing to this word serialised the whole scan)
            wseg = uint32_t(__builtin_amdgcn_readfirstlane(int(ns)));
            wused = 0;
            if (wseg >= ev_max_segs) { wseg = kTriSegDead; ev_has = 0; return; }
        }
        if (ev_has) ev_buf[size_t(wseg) * kTriSeg + wused + rank] = TriEvent{ci, ev_pre, ev_state, uint32_t(rel0 + int32_t(ev_idx))};
        wused += n;
        ev_has = 0;
#else
        if (ev_has && *ev_ctr < ev_max_segs) ev_buf[(*ev_ctr)++] = TriEvent{ci, ev_pre, ev_state, uint32_t(rel0 + int32_t(ev_idx))};   // (host: ev_max_segs = capacity in events)
        ev_has = 0;
#endif
    }
    // End of the lane's walk: the wavefront's last segment gets its fill recorded.
    ACGPU_TRI_FN void finish_events() {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (ev_buf && lane == 0 && wseg < ev_max_segs) ev_seg_fill[wseg] = wused;
#endif
    }
    // wds: the 16 bytes; act16 bit i: byte i lies inside the lane's range [walk start, chunk end)
    // (SM: the automaton has matches of <= 2 bytes -- a template parameter, not a test of `sm` per byte: straight-line
    // code lets the compiler issue the 32 LDS reads of a piece ahead of their uses)
    template <bool ALL_ACTIVE, bool SM>
    ACGPU_TRI_FN void piece_scan(const uint32_t (&wds)[4], uint32_t act16) {
#if defined(__HIP_DEVICE_COMPILE__)
        // Device form, instruction by instruction (round 4).  Left to itself the compiler turned the pair index
        // `b * A + uc` into v_mad_u64_u32 (quarter rate, once per byte: it knows both factors are small and drops the 24-bit
        // multiply) and the bit-word address into multiply + shift + add3: ~12 VALU per byte, one of them four times as long.
        // Here: pair of the NEXT byte = v_mad_u32_u24(b, A, uc); word address = v_mad_u32_u24(pair, 4 bw, (uc >> 5) * 4);
        // bit = v_bfe_u32(word, uc, 1) (the hardware takes the low five bits of uc); mask = v_lshl_or_b32.
        auto mad24 = [](uint32_t x, uint32_t y, uint32_t z) {
            uint32_t r;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
            return r;
        };
        const uint32_t bw4 = bw * 4u;
        const uint8_t* bits8 = reinterpret_cast<const uint8_t*>(s_bits);
        uint32_t pair = mad24(ua, A, ub), b = ub, m = 0;   // pair of the two classes in front of the byte at hand
        uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t uc = s_uc[__builtin_amdgcn_ubfe(wds[i >> 2], 8 * (i & 3), 8)];
            if (!ALL_ACTIVE) uc = ((act16 >> i) & 1u) ? uc : U;
            const uint32_t w = *reinterpret_cast<const uint32_t*>(bits8 + mad24(pair, bw4, (uc >> 3) & 28u));
            uint32_t bit = __builtin_amdgcn_ubfe(w, uc & 31u, 1);
            const uint32_t next_pair = mad24(b, A, uc);
            if (SM) bit |= s_mc2[next_pair] != 0 ? 1u : 0u;
            m |= bit << i;
            pk[i >> 2] |= uc << (8 * (i & 3));
            na = b;
            b = uc;
            pair = next_pair;
        }
        nb = b;
        cand = ALL_ACTIVE ? m : (m & act16);
        *reinterpret_cast<uint4*>(s_buf) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
#else
        uint32_t ta = ACGPU_TRI_MUL24(ua, A), b = ub, m = 0;
        uint32_t pk[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) {
            const uint32_t byte = (wds[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            uint32_t uc = s_uc[byte];
            if (!ALL_ACTIVE) uc = ((act16 >> i) & 1u) ? uc : U;
            const uint32_t prj = ta + b;
            const uint32_t w = s_bits[ACGPU_TRI_MUL24(prj, bw) + (uc >> 5)];
            uint32_t bit = (w >> (uc & 31)) & 1u;
            const uint32_t tb = ACGPU_TRI_MUL24(b, A);
            if (SM) bit |= s_mc2[tb + uc] != 0 ? 1u : 0u;
            m |= bit << i;
            pk[i >> 2] |= uc << (8 * (i & 3));
            ta = tb;
            na = b;
            b = uc;
        }
        nb = b;
        cand = ALL_ACTIVE ? m : (m & act16);
        for (int i = 0; i < 16; i++) s_buf[i] = uint8_t(pk[i >> 2] >> (8 * (i & 3)));
#endif
    }
    // Index (into the table of depth-3 nodes) of the child `uc` of pair `prj`: the pair's base + the rank of the bit.
    ACGPU_TRI_FN uint32_t child_index(uint32_t prj, uint32_t bitsw, uint32_t uc) const {
        uint32_t rank = __builtin_popcount(bitsw & ((1u << (uc & 31)) - 1u));
        const uint32_t w0 = ACGPU_TRI_MUL24(prj, bw);
        for (uint32_t i = 0; i < (uc >> 5); i++) rank += __builtin_popcount(s_bits[w0 + i]);
        uint32_t ce = (uint32_t(s_base[prj]) << gshift) + rank;
#if defined(ACGPU_GUARD) && defined(__HIP_DEVICE_COMPILE__)
        if (ce >= n_child) { if (guard && atomicAdd(guard, 1ull) < 8) printf("tri walk: child entry %u >= %u (pair %u)\n", ce, n_child, prj); ce = 0; }
#endif
        return ce;
    }
    // The shallow lane's next candidate at or behind pos (pos < lim).  0: none in this piece (pos = lim).  1: byte j
    // enters depth 3 (prj, bitsw, uc describe the trigram; pos = j + 1).  2: byte j only ends matches of <= 2 bytes,
    // which have been counted (pos = j + 1).  own_from: index of the first byte whose matches this chunk owns.
    ACGPU_TRI_FN uint32_t shallow_jump(uint32_t lim, uint32_t own_from, uint32_t& prj, uint32_t& bitsw, uint32_t& uc,
                                       uint32_t& j, tri_flag& owned) {
        const uint32_t m = cand >> pos;
        if (m == 0) { pos = lim; return 0; }
        j = pos + uint32_t(__builtin_ctz(m));
        // the two classes in front of byte j: from the piece, or carried over from the piece before
        const uint32_t c1 = j >= 1 ? uint32_t(s_buf[j >= 1 ? j - 1 : 0]) : ub;
        const uint32_t c2 = j >= 2 ? uint32_t(s_buf[j >= 2 ? j - 2 : 0]) : (j == 1 ? ub : ua);
        uc = s_buf[j];
        prj = ACGPU_TRI_MUL24(c2, A) + c1;
        pr = prj;
        bitsw = s_bits[ACGPU_TRI_MUL24(prj, bw) + (uc >> 5)];
        owned = j >= own_from ? 1u : 0u;
        pos = j + 1;
        if ((bitsw >> (uc & 31)) & 1u) return 1;
        if (owned) {   // (a candidate without its bit: a state of depth <= 2 with matches, sm is set)
            const uint32_t p2 = ACGPU_TRI_MUL24(c1, A) + uc;
            note_event(0x80000000u | p2, j, s_mc2[p2]);
        }
        return 2;
    }
    // the pair of compact classes (byte j - 1, byte j) of the piece at hand
    ACGPU_TRI_FN uint32_t pair_at(uint32_t j) const {
        const uint32_t c1 = j >= 1 ? uint32_t(s_buf[j >= 1 ? j - 1 : 0]) : ub;
        return ACGPU_TRI_MUL24(c1, A) + s_buf[j];
    }
};

}  // namespace acgpu
