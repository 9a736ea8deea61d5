// Device-side pieces of the LDS transition walk shared by its kernels (lds_walk.hip: the count walk and the chunk fill;
// lds_emit.hip: the streaming record emit and the non-overlapping chain): kernel arguments, the LDS image accessors, the
// class value of a haystack byte (LDS map or computed clamp) and the hand-scheduled steps.  See lds_walk.hip for the design.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../host/lw_tables.hpp"
#include "hot.hpp"

namespace acgpu {
namespace lwdev {

constexpr uint32_t kLwLdsBytes = kLwLdsBudget;
constexpr uint32_t kLwLaneChunk = 512;   // target bytes per lane-chunk

constexpr uint32_t kLwCls = kLwClsBytes;   // the class map occupies LDS bytes [0, 512); table addresses are relative to 512

struct LwArgs {
    const uint32_t* image;     // LDS image: class map | tables (host/lw_tables.hpp)
    uint32_t image_bytes;
    uint32_t row_bytes;        // bytes per row (an odd number of dwords: host/lw_tables.cpp)
    uint32_t fm_addr;          // 4 * first_match: handles whose da is >= this are match / multi / poison
    uint32_t virt_addr;        // 4 * n_states: ... >= this are multi / poison
    uint32_t nxt_off;          // u32 [n_virtual]: next handle of an exception chain
    uint32_t vhid_off;         // u16 [n_virtual]: the real state behind the first slot of a multi state
    uint32_t mlen_off;         // u16 [n_states - first_match]: match-list lengths
    uint32_t poison_base;      // row index of the poison row
    uint32_t start;            // handle of the unanchored start state
    int32_t cc_add, cc_lo, cc_hi;   // computed class value = med3(byte + cc_add, cc_lo, cc_hi)
    uint32_t list_col;         // kLwFull: column of a row that holds the byte offset of the state's match list (k_lw_fill)
    // lane-chunk geometry (sub-division of the scan's count chunks)
    uint32_t lane_chunk;       // bytes per lane-chunk (multiple of 64)
    uint32_t lanes_per_chunk;  // power of two <= 64: lane-chunks per count chunk
    uint32_t warm_pieces;      // ceil(halo / 16)
    uint64_t n_lane_chunks, n_tasks;
};

struct LwLds {
    const uint8_t* base;   // LDS byte 0 of the image
    __device__ __forceinline__ uint32_t rd32(uint32_t table_addr) const {   // the constant lands in the DS offset field
        return *reinterpret_cast<const uint32_t*>(base + kLwCls + table_addr);
    }
    __device__ __forceinline__ uint32_t rd16(uint32_t table_addr) const {
        return *reinterpret_cast<const uint16_t*>(base + kLwCls + table_addr);
    }
    __device__ __forceinline__ uint32_t map16(uint32_t byte_addr) const { return *reinterpret_cast<const uint16_t*>(base + byte_addr); }
};

// ---- class value of byte K of dword w
template <int K> __device__ __forceinline__ uint32_t lw_byte_x2(uint32_t w) {   // 2 * byte K: the address in the u16 class map
    uint32_t r;
    const uint32_t one = 1;
    if constexpr (K == 0) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(w), "v"(one));
    if constexpr (K == 1) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(w), "v"(one));
    if constexpr (K == 2) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(w), "v"(one));
    if constexpr (K == 3) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(w), "v"(one));
    return r;
}
template <int K> __device__ __forceinline__ uint32_t lw_byte_plus(uint32_t w, uint32_t add) {   // byte K + add
    uint32_t r;
    if constexpr (K == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w), "s"(add));
    if constexpr (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w), "s"(add));
    if constexpr (K == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(w), "s"(add));
    if constexpr (K == 3) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(w), "s"(add));
    return r;
}
// Per-lane constants of the computed class: the lower clamp bound lives in a VGPR (a VOP3 reads one scalar operand only)
struct LwCc { uint32_t add; uint32_t v_lo; uint32_t hi; };
template <int K> __device__ __forceinline__ uint32_t lw_byte_x4(uint32_t w) {   // 4 * byte K
    uint32_t r;
    const uint32_t two = 2;
    if constexpr (K == 0) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(w), "v"(two));
    if constexpr (K == 1) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(w), "v"(two));
    if constexpr (K == 2) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(w), "v"(two));
    if constexpr (K == 3) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(w), "v"(two));
    return r;
}
// X4 (kLwFull): the value is 4 * class -- computed as med3(4 * byte + add, lo, hi) with the bounds premultiplied
template <bool CC, int K, bool X4 = false>
__device__ __forceinline__ uint32_t lw_clsval(const LwLds& L, const LwCc& cc, uint32_t w) {
    if constexpr (CC && X4) {
        const uint32_t x = lw_byte_x4<K>(w) + cc.add;
        uint32_t r;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(cc.v_lo), "s"(cc.hi));
        return r;
    } else if constexpr (CC) {
        const uint32_t x = lw_byte_plus<K>(w, cc.add);
        uint32_t r;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(cc.v_lo), "s"(cc.hi));
        return r;
    } else {
        return L.map16(lw_byte_x2<K>(w));
    }
}
// the same for one byte value (edge walk, exact step)
template <bool CC, bool X4 = false>
__device__ __forceinline__ uint32_t lw_clsval_byte(const LwLds& L, const LwArgs& a, uint32_t byte) {
    if constexpr (CC) {
        const int32_t x = int32_t(X4 ? 4 * byte : byte) + a.cc_add;
        return uint32_t(x < a.cc_lo ? a.cc_lo : x > a.cc_hi ? a.cc_hi : x);
    } else {
        return L.map16(byte * 2);
    }
}

// ---- the fast step, hand-scheduled (gfx950).
// Narrow:  a = (class == h.e) ? h.da : h.base * row_bytes + 4 * class_value   -- 4 VALU; h' = LDS[a] -- ds_read_b32.
// The compare writes VCC and the select reads it two instructions later (the wait states gfx950 needs between a VALU
// write of VCC and a VALU read of it); SDWA operand selects pick h.e / h.base / h.da without separate shifts.
__device__ __forceinline__ uint32_t lw_addr(uint32_t h, uint32_t cv, uint32_t row_bytes) {
    uint32_t a, t;
    asm("v_cmp_eq_u32_sdwa vcc, %2, %3 src0_sel:BYTE_2 src1_sel:BYTE_0\n\t"
        "v_mul_u32_u24_sdwa %1, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "v_lshl_add_u32 %1, %3, 2, %1\n\t"
        "v_cndmask_b32_sdwa %0, %1, %2, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=&v"(a), "=&v"(t)
        : "v"(h), "v"(cv), "s"(row_bytes)
        : "vcc");
    return a;
}
// The wide-base layout (base 10 | e 6 | da 16 bits; small alphabets with more than 254 rows): no byte selects, 6 VALU.
__device__ __forceinline__ uint32_t lw_addr_wide(uint32_t h, uint32_t cv, uint32_t row_bytes) {
    const uint32_t e = __builtin_amdgcn_ubfe(h, 16, 6);
    const uint32_t ra = __umul24(h >> 22, row_bytes) + (cv << 2);
    return e == (cv & 0xFFu) ? (h & 0xFFFFu) : ra;
}
// Full: a = h.hi16 + class4
__device__ __forceinline__ uint32_t lw_addr_full(uint32_t h, uint32_t cv4) {
    uint32_t t;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(t) : "v"(cv4), "v"(h));
    return t;
}
__device__ __forceinline__ uint32_t lw_max3_u16(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t r;
    asm("v_max3_u16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}

// The exact step (any state kind): resolves exception chains.  Rare path; LDS only.
template <int FLAV, bool CC>
__device__ __forceinline__ uint32_t lw_careful_step(const LwArgs& a, const LwLds& L, uint32_t h, uint32_t byte) {
    const uint32_t cv = lw_clsval_byte<CC, FLAV == kLwFull>(L, a, byte);
    if constexpr (FLAV == kLwFull) return L.rd32((h >> 16) + cv);
    const uint32_t c = cv & 0xFFu;
    constexpr uint32_t base_shift = FLAV == kLwWide ? 22 : 24, e_mask = FLAV == kLwWide ? 0x3Fu : 0xFFu;
    for (int hop = 0; hop < 4096; hop++) {
        const uint32_t da = h & 0xFFFFu;
        if (((h >> 16) & e_mask) == c) return L.rd32(da);
        const uint32_t b = h >> base_shift;
        if (b != a.poison_base) return L.rd32(b * a.row_bytes + cv * 4);
        h = L.rd32(a.nxt_off + (da - a.virt_addr));   // multi state / chain link: da names a virtual slot
    }
    return h;
}

// Number of matches of the state behind handle h (src/dfa.rs:275-279: the length of its match list).
template <int FLAV>
__device__ __forceinline__ uint32_t lw_match_len(const LwArgs& a, const LwLds& L, uint32_t h) {
    if constexpr (FLAV == kLwFull) return h & kLwFullLenMask;
    uint32_t da = h & 0xFFFFu;
    if (da >= a.virt_addr) da = 4 * L.rd16(a.vhid_off + ((da - a.virt_addr) >> 1));   // first slot of a multi state
    return da >= a.fm_addr ? L.rd16(a.mlen_off + ((da - a.fm_addr) >> 1)) : 0u;
}

// Re-walk of one dword from the saved handle, for the lanes whose fast walk met a multi state.
template <int FLAV, bool CC>
__device__ __forceinline__ uint32_t lw_redo4(const LwArgs& a, const LwLds& L, uint32_t h, uint32_t w, bool owned, uint32_t& cnt) {
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        h = lw_careful_step<FLAV, CC>(a, L, h, (w >> (8 * k)) & 0xFFu);
        if (owned) cnt += lw_match_len<FLAV>(a, L, h);
    }
    return h;
}

// Generic (edge) walk of one lane-chunk -- the first and last wave regions of a shard, and every chunk of a small
// input: exact step, ownership from `lo`.  The bytes come in aligned 16-byte pieces (only pieces holding a live byte are
// touched), two pieces ahead: a dependent global load per byte cost ~0.5 us each, 0.25 ms for one 512-byte chunk.
template <int FLAV, bool CC>
__device__ __forceinline__ uint32_t lw_edge_walk(const LwArgs& a, const LwLds& L, const ScanGeom& g, uint64_t w, uint64_t lo,
                                                 uint64_t hi, uint32_t cnt) {
    const uint8_t* hay16 = g.hay16;
    uint32_t h = a.start;
    const uint64_t p0 = w & ~uint64_t(15);
    auto ld = [&](uint64_t p) {
        if (p < hi) ACGPU_HAY_CHECK(g, p, 16);
        return p < hi ? *reinterpret_cast<const uint4*>(hay16 + p) : make_uint4(0, 0, 0, 0);
    };
    uint4 q0 = ld(p0), q1 = ld(p0 + 16);
    for (uint64_t p = p0; p < hi; p += 16) {
        const uint4 q = q0;
        q0 = q1;
        q1 = ld(p + 32);
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t v = p + k;
            if (v >= w && v < hi) {
                h = lw_careful_step<FLAV, CC>(a, L, h, (wd[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                if (v >= lo) cnt += lw_match_len<FLAV>(a, L, h);
            }
        }
    }
    return cnt;
}

// One dword = 4 steps on the fast path.  One chain per lane: two or three independent chains per lane on fewer, fatter
// wavefronts (768 / 512 threads, the register file bounds chains x 64 line registers) were measured SLOWER -- 3.6 -> 2.4 /
// 2.2 TB/s on the headline set, 3.3 -> 2.1 / 1.9 on one-row-per-state automata (profiles/r05_hot_chains_ab_noinline.jsonl,
// r05_hot_chains3_pmc.json): with 16 wavefronts the VALU is 92 % and the LDS 88 % busy (profiles/r05_hot_pmc.json), latency is
// not what bounds the walk.
template <int FLAV, bool CC, bool OWNED>
__device__ __forceinline__ void lw_step4(const LwArgs& a, const LwLds& L, const LwCc& cc, uint32_t w, uint32_t& h, uint32_t& cnt) {
    constexpr bool X4 = FLAV == kLwFull;
    const uint32_t cv0 = lw_clsval<CC, 0, X4>(L, cc, w), cv1 = lw_clsval<CC, 1, X4>(L, cc, w);
    const uint32_t cv2 = lw_clsval<CC, 2, X4>(L, cc, w), cv3 = lw_clsval<CC, 3, X4>(L, cc, w);
    if constexpr (FLAV == kLwFull) {
        const uint32_t h1 = L.rd32(lw_addr_full(h, cv0));
        const uint32_t h2 = L.rd32(lw_addr_full(h1, cv1));
        const uint32_t h3 = L.rd32(lw_addr_full(h2, cv2));
        h = L.rd32(lw_addr_full(h3, cv3));
        if (OWNED) cnt += (h1 + h2 + h3 + h) & kLwFullSumMask;   // four match-list lengths of at most 4 095 (the sync flags and the high halves only carry upwards)
    } else {
        const uint32_t rb = a.row_bytes;
        auto addr = [&](uint32_t hh, uint32_t c) { return FLAV == kLwWide ? lw_addr_wide(hh, c, rb) : lw_addr(hh, c, rb); };
        const uint32_t h0 = h;
        const uint32_t h1 = L.rd32(addr(h0, cv0));
        const uint32_t h2 = L.rd32(addr(h1, cv1));
        const uint32_t h3 = L.rd32(addr(h2, cv2));
        const uint32_t h4 = L.rd32(addr(h3, cv3));
        h = h4;
        const uint32_t worst = lw_max3_u16(lw_max3_u16(h1, h2, h3), h4, h4) & 0xFFFFu;
        const bool flag = worst >= a.fm_addr;
        if (__builtin_expect(__any(flag), 0)) {
            if (flag) {
                if (worst >= a.virt_addr) {          // a multi state or poison: the handles are not exact -- redo the dword
                    h = lw_redo4<FLAV, CC>(a, L, h0, w, OWNED, cnt);
                } else if (OWNED) {                  // match states only: exact handles, one u16 gather per matching byte
                    const uint32_t moff = a.mlen_off - (a.fm_addr >> 1);
                    const uint32_t d1 = h1 & 0xFFFFu, d2 = h2 & 0xFFFFu, d3 = h3 & 0xFFFFu, d4 = h4 & 0xFFFFu;
                    if (d1 >= a.fm_addr) cnt += L.rd16(moff + (d1 >> 1));
                    if (d2 >= a.fm_addr) cnt += L.rd16(moff + (d2 >> 1));
                    if (d3 >= a.fm_addr) cnt += L.rd16(moff + (d3 >> 1));
                    if (d4 >= a.fm_addr) cnt += L.rd16(moff + (d4 >> 1));
                }
            }
        }
    }
}

}  // namespace lwdev
}  // namespace acgpu
