// The per-byte step of the contiguous-NFA shallow-skip walk (k_cnfa_tri, cnfa_tri.hip), shared with the host: the
// test hook acgpu_test_cnfa_tri_host runs THIS code lane by lane on the CPU (tests/test_cnfa_tables.py).
#pragma once
#include <stdint.h>

#include "../host/cnfa_tri_tables.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ACGPU_TRI_FN __host__ __device__ __forceinline__
#else
#define ACGPU_TRI_FN inline
#endif
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

enum : uint32_t { MD_SHALLOW = 0, MD_NOREC = 1, MD_REC = 2 };

// One lane's walk.  The haystack is consumed in 16-byte pieces:
//   piece_scan  -- branch-free, the same for every lane: compact classes of the 16 bytes (kept in this lane's 16 bytes of
//                  LDS for the walk below) and the candidate mask: bit i = "if the automaton's state has depth <= 2 in
//                  front of byte i, byte i leaves that regime (the trigram is a trie node) or ends a match of <= 2 bytes";
//   piece_walk  -- every lane jumps from candidate to candidate (ctz) while its state is shallow and walks byte by byte
//                  (state record, class compare, failure link: contiguous.rs:186-247) while it is deep; one gather
//                  round per trip for all lanes that need one.
// All lane flags are 32-bit values, not bool: carried around the loops as lane masks they came out wrong on the
// device (identical source; one lane per wavefront: right, 64 lanes: 1e-4 of the counts off).
struct TriWalk {
    const uint32_t* s_bits;
    const uint16_t* s_base;
    const uint8_t* s_uc;            // [256] byte -> compact class (U: the byte labels no trie edge)
    const uint8_t* s_inv;           // [256] compact class -> the automaton's class
    const uint8_t* s_mc2;
    uint8_t* s_buf;                 // this lane's 16 bytes of LDS: the compact classes of the piece at hand
    const TriChild* child;
    const uint32_t* repr3;
    uint32_t A, bw, gshift, U, alen, max_match;
    uint32_t sm;                    // wave-uniform: some state of depth <= 2 is a match state
    uint32_t repr_words, n_child;   // table sizes (bounds-checked flavour only)
    unsigned long long* guard;
    // lane state
    uint32_t md, o, head, fail, d0, d1, cnt;
    uint32_t ua, ub;                // compact classes of the two bytes in front of the piece at hand
    uint32_t na, nb;                // ... of its last two bytes (piece_scan)
    uint32_t hd1, pend, cand, pos;  // pend: 1 + index (in the piece) of the byte that led into state o, whose matches are still to be counted
    uint32_t pr;                    // (diagnostics of the bounds-checked flavour)
    // match events (optional: ev_buf == nullptr counts only): one per match state entered at an owned position --
    // {chunk, records of the chunk in front of it, state, position} -- appended to wave-private segments of kTriSeg
    // events, so that k_cnfa_tri_emit can write the ordered records without walking anything again
    TriEvent* ev_buf;
    uint32_t* ev_seg_fill;
    unsigned long long* ev_ctr;     // [0] segments handed out, [1] overflow flag
    uint32_t ev_max_segs, ci;
    uint32_t wseg, wused;           // wave-uniform: the wavefront's current segment and its fill
    uint32_t ev_has, ev_state, ev_idx, ev_pre;

    ACGPU_TRI_FN uint32_t word(uint32_t i) const {   // word i of the current state's record
        if (i == 0) return head;
        if (i == 1) return fail;
        if (i == 2) return d0;
        if (i == 3 && hd1) return d1;
        uint32_t x = o + i;
        ACGPU_TRI_BOUND(x, repr_words, "record word");
        return repr3[x];
    }
    static ACGPU_TRI_FN uint32_t byte_index(uint32_t w, uint32_t k4) {   // 0..3: the byte of w equal to k, 4: none
        const uint32_t x = w ^ k4;
        const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
        return z ? uint32_t(__builtin_ctz(z)) >> 3 : 4u;
    }
    // the state just entered (its record is in hand) ends matches: contiguous.rs:581-598
    ACGPU_TRI_FN void account(uint32_t idx) {
        if (o > max_match) return;
        const uint32_t kind = head & 0xFFu;
        const uint32_t base = kind == 0xFFu ? 2 + alen : (kind == 0xFEu ? 3u : 2 + ((kind + 3) >> 2) + kind);
        const uint32_t packed = word(base);
        ev_has = 1; ev_state = o; ev_idx = idx; ev_pre = cnt;
        cnt += (packed & (1u << 31)) ? 1u : packed;
    }
    // End of a trip (wave-uniform control flow): the lanes that counted a match append their event.
    ACGPU_TRI_FN void flush_events(int32_t rel0) {
        if (!ev_buf) { ev_has = 0; return; }
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned long long mask = __ballot(ev_has != 0);
        if (mask == 0) return;
        const uint32_t n = uint32_t(__popcll(mask));
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (wused + n > kTriSeg) {   // (also the first event of the wavefront: wused starts at kTriSeg)
            uint32_t ns = 0;
            if (lane == 0) {
                if (wseg < ev_max_segs) ev_seg_fill[wseg] = wused;
                ns = uint32_t(atomicAdd(ev_ctr, 1ull));
            }
            wseg = uint32_t(__builtin_amdgcn_readfirstlane(int(ns)));
            wused = 0;
        }
        if (ev_has) {
            if (wseg < ev_max_segs) ev_buf[size_t(wseg) * kTriSeg + wused + rank] = TriEvent{ci, ev_pre, ev_state, uint32_t(rel0 + int32_t(ev_idx))};
            else ev_ctr[1] = 1ull;   // more events than the buffer holds: the caller falls back to the re-walking fill
        }
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
    // One step of contiguous.rs:186-247 from the record in hand, on the byte at `pos` (class k).
    ACGPU_TRI_FN void attempt(uint32_t k, uint32_t owned) {
        const uint32_t kind = head & 0xFFu;
        uint32_t found = 0, target = 0;
        if (kind == 0xFEu) {
            found = k == ((head >> 8) & 0xFFu) ? 1u : 0u;
            target = d0;
        } else if (kind == 0xFFu) {
            uint32_t x = o + 2 + k;
            ACGPU_TRI_BOUND(x, repr_words, "dense transition");
            target = repr3[x];
            found = target != 1u /*FAIL*/ ? 1u : 0u;
        } else {
            const uint32_t tl = kind, cl = (tl + 3) >> 2, k4 = k * 0x01010101u;
            for (uint32_t i = 0; i < cl && !found; i++) {
                const uint32_t j = byte_index(word(2 + i), k4);
                if (j < 4 && i * 4 + j < tl) { target = word(2 + cl + i * 4 + j); found = 1; }
            }
        }
        if (found) { o = target; md = MD_NOREC; pend = owned ? pos + 1 : 0u; pos++; }
        else if (fail & kTriShallow) md = MD_SHALLOW;
        else { o = fail; md = MD_NOREC; }
    }
    // The gather of a trip and what follows from it: `need_child` lanes fetch the entry of the depth-3 node they enter
    // (base of the pair + rank of the bit among the pair's children), lanes without a record (MD_NOREC) fetch theirs.
    ACGPU_TRI_FN void gather(uint32_t need_child, uint32_t owned, uint32_t prj, uint32_t bitsw, uint32_t uc, uint32_t j) {
        uint32_t x = o;
        if (!need_child) ACGPU_TRI_BOUND(x, repr_words - 3, "state record");
        const uint32_t* addr = repr3 + x;
        if (need_child) {
            uint32_t rank = __builtin_popcount(bitsw & ((1u << (uc & 31)) - 1u));
            const uint32_t w0 = ACGPU_TRI_MUL24(prj, bw);
            for (uint32_t i = 0; i < (uc >> 5); i++) rank += __builtin_popcount(s_bits[w0 + i]);
            uint32_t ce = (uint32_t(s_base[prj]) << gshift) + rank;
            ACGPU_TRI_BOUND(ce, n_child, "child entry");
            addr = reinterpret_cast<const uint32_t*>(child + ce);
        }
        const tri_u32x4 v = *reinterpret_cast<const tri_u32x4*>(addr);
        md = MD_REC;
        if (need_child) {
            o = v.x; head = v.y; fail = v.z; d0 = v.w; hd1 = 0;
            if (owned) account(j);
        } else {
            head = v.x; fail = v.y; d0 = v.z; d1 = v.w; hd1 = 1;
            if (pend) { account(pend - 1); pend = 0; }
        }
    }
    // wds: the 16 bytes; act16 bit i: byte i lies inside the lane's range [walk start, chunk end)
    template <bool ALL_ACTIVE>
    ACGPU_TRI_FN void piece_scan(const uint32_t (&wds)[4], uint32_t act16) {
        uint32_t ta = ACGPU_TRI_MUL24(ua, A), b = ub, m = 0;
        uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t byte = (wds[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            uint32_t uc = s_uc[byte];
            if (!ALL_ACTIVE) uc = ((act16 >> i) & 1u) ? uc : U;
            const uint32_t prj = ta + b;
            const uint32_t w = s_bits[ACGPU_TRI_MUL24(prj, bw) + (uc >> 5)];
            uint32_t bit = (w >> (uc & 31)) & 1u;
            const uint32_t tb = ACGPU_TRI_MUL24(b, A);
            if (sm) bit |= s_mc2[tb + uc] != 0 ? 1u : 0u;
            m |= bit << i;
            pk[i >> 2] |= uc << (8 * (i & 3));
            ta = tb;
            na = b;
            b = uc;
        }
        nb = b;
        cand = ALL_ACTIVE ? m : (m & act16);
#if defined(__HIP_DEVICE_COMPILE__)
        *reinterpret_cast<uint4*>(s_buf) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
#else
        for (int i = 0; i < 16; i++) s_buf[i] = uint8_t(pk[i >> 2] >> (8 * (i & 3)));
#endif
    }
    // lim: bytes of the piece in front of the chunk end (0..16); own_from: index of the first byte whose matches this
    // chunk owns (0..16)
    // rel0: position of the piece's byte 0 relative to the chunk's grid origin (events)
    ACGPU_TRI_FN void piece_walk(uint32_t lim, uint32_t own_from, int32_t rel0) {
        pos = 0;   // (pend == 0 here: a piece ends with every record fetched and every match counted)
        for (;;) {
            if (md == MD_REC && pos < lim) attempt(s_inv[s_buf[pos]], pos >= own_from ? 1u : 0u);
            uint32_t need_child = 0, prj = 0, bitsw = 0, uc = 0, owned = 0, jc = 0;
            if (md == MD_SHALLOW && pos < lim) {
                const uint32_t m = cand >> pos;
                if (m == 0) {
                    pos = lim;
                } else {
                    const uint32_t j = pos + uint32_t(__builtin_ctz(m));
                    // the two classes in front of byte j: from the piece, or carried over from the piece before
                    const uint32_t c1 = j >= 1 ? uint32_t(s_buf[j >= 1 ? j - 1 : 0]) : ub;
                    const uint32_t c2 = j >= 2 ? uint32_t(s_buf[j >= 2 ? j - 2 : 0]) : (j == 1 ? ub : ua);
                    uc = s_buf[j];
                    prj = ACGPU_TRI_MUL24(c2, A) + c1;
                    pr = prj;
                    bitsw = s_bits[ACGPU_TRI_MUL24(prj, bw) + (uc >> 5)];
                    owned = j >= own_from ? 1u : 0u;
                    jc = j;
                    pos = j + 1;
                    if ((bitsw >> (uc & 31)) & 1u) need_child = 1;
                    else if (owned) {   // (a candidate without its bit: a state of depth <= 2 with matches, sm is set)
                        const uint32_t p2 = ACGPU_TRI_MUL24(c1, A) + uc;
                        ev_has = 1; ev_state = 0x80000000u | p2; ev_idx = j; ev_pre = cnt;
                        cnt += s_mc2[p2];
                    }
                }
            }
            const uint32_t need = need_child | (md == MD_NOREC ? 1u : 0u);
            if (ACGPU_TRI_ANY(need != 0)) {
                if (need) gather(need_child, owned, prj, bitsw, uc, jc);
            }
            flush_events(rel0);
            if (!ACGPU_TRI_ANY(md == MD_NOREC || pos < lim)) break;
        }
        ua = na;
        ub = nb;
    }
};

}  // namespace acgpu
