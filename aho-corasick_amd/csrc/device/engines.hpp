// Device-side views of the automata and the per-byte transition primitives.
//
// Two engines implement the reference's `Automaton` primitives on the GPU:
//   DfaEng  -- src/dfa.rs:218-286   (premultiplied u32 table + 256-entry class map)
//   CnfaEng -- src/nfa/contiguous.rs:186-247, :581-633 (packed `repr` words, failure links)
// Both are used by the generic kernels (chunked overlapping walk, serial
// find_iter).  The LDS-resident fast path for the DFA lives in hot_scan.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acgpu {

struct DevSpecial {
    uint32_t max_special_id, max_match_id, start_unanchored_id, start_anchored_id;
};

struct DevDfa {
    const uint32_t* trans;    // [state_len << stride2], premultiplied ids
    const uint32_t* moff;     // CSR over (sid >> stride2) - 2
    const uint32_t* mpid;
    const uint32_t* plens;    // [n_patterns]
    const uint8_t* classes;   // [256] (global copy; kernels stage it into LDS)
    uint32_t stride2;
    DevSpecial sp;
};

struct DevCnfa {
    const uint32_t* repr;
    const uint32_t* plens;
    const uint8_t* classes;
    uint32_t alphabet_len;
    DevSpecial sp;
};

constexpr uint32_t kDevDead = 0, kDevFail = 1;

struct DfaEng {
    DevDfa d;
    const uint8_t* cls;  // LDS or global class map
    __device__ __forceinline__ uint32_t start(bool anchored) const {
        return anchored ? d.sp.start_anchored_id : d.sp.start_unanchored_id;
    }
    // src/dfa.rs:218-226
    __device__ __forceinline__ uint32_t next(bool /*anchored*/, uint32_t sid, uint8_t byte) const {
        return d.trans[sid + cls[byte]];
    }
    __device__ __forceinline__ bool is_special(uint32_t sid) const { return sid <= d.sp.max_special_id; }
    __device__ __forceinline__ bool is_match(uint32_t sid) const { return sid != kDevDead && sid <= d.sp.max_match_id; }
    // src/dfa.rs:275-286
    __device__ __forceinline__ uint32_t match_len(uint32_t sid) const {
        uint32_t o = (sid >> d.stride2) - 2;
        return d.moff[o + 1] - d.moff[o];
    }
    __device__ __forceinline__ uint32_t match_pattern(uint32_t sid, uint32_t i) const {
        uint32_t o = (sid >> d.stride2) - 2;
        return d.mpid[d.moff[o] + i];
    }
    __device__ __forceinline__ uint32_t pattern_len(uint32_t pid) const { return d.plens[pid]; }
};

struct CnfaEng {
    DevCnfa c;
    const uint8_t* cls;
    static constexpr uint32_t KIND_DENSE = 0xFF, KIND_ONE = 0xFE;
    __device__ __forceinline__ static uint32_t u32_len(uint32_t n) { return (n + 3) >> 2; }
    __device__ __forceinline__ uint32_t start(bool anchored) const {
        return anchored ? c.sp.start_anchored_id : c.sp.start_unanchored_id;
    }
    // src/nfa/contiguous.rs:186-247
    __device__ __forceinline__ uint32_t next(bool anchored, uint32_t sid, uint8_t byte) const {
        const uint32_t* repr = c.repr;
        const uint32_t k = cls[byte];
        for (;;) {
            const uint32_t o = sid;
            const uint32_t head = repr[o];
            const uint32_t kind = head & 0xFF;
            if (kind == KIND_DENSE) {
                uint32_t nx = repr[o + 2 + k];
                if (nx != kDevFail) return nx;
            } else if (kind == KIND_ONE) {
                if (k == ((head >> 8) & 0xFF)) return repr[o + 2];
            } else {
                const uint32_t tl = kind, cl = u32_len(tl);
                const uint32_t to = o + 2 + cl;
                for (uint32_t i = 0; i < cl; i++) {
                    const uint32_t w = repr[o + 2 + i];  // classes packed in memory order (little endian)
                    if ((w & 0xFF) == k) return repr[to + i * 4];
                    if (((w >> 8) & 0xFF) == k) return repr[to + i * 4 + 1];
                    if (((w >> 16) & 0xFF) == k) return repr[to + i * 4 + 2];
                    if ((w >> 24) == k) return repr[to + i * 4 + 3];
                }
            }
            if (anchored) return kDevDead;
            sid = repr[o + 1];
        }
    }
    __device__ __forceinline__ bool is_special(uint32_t sid) const { return sid <= c.sp.max_special_id; }
    __device__ __forceinline__ bool is_match(uint32_t sid) const { return sid != kDevDead && sid <= c.sp.max_match_id; }
    __device__ __forceinline__ uint32_t match_base(uint32_t sid) const {
        const uint32_t kind = c.repr[sid] & 0xFF;
        if (kind == KIND_DENSE) return sid + 2 + c.alphabet_len;
        return sid + 2 + u32_len(kind) + kind;
    }
    // src/nfa/contiguous.rs:581-633
    __device__ __forceinline__ uint32_t match_len(uint32_t sid) const {
        uint32_t packed = c.repr[match_base(sid)];
        return (packed & (1u << 31)) ? 1u : packed;
    }
    __device__ __forceinline__ uint32_t match_pattern(uint32_t sid, uint32_t i) const {
        uint32_t b = match_base(sid);
        uint32_t packed = c.repr[b];
        return (packed & (1u << 31)) ? (packed & ~(1u << 31)) : c.repr[b + 1 + i];
    }
    __device__ __forceinline__ uint32_t pattern_len(uint32_t pid) const { return c.plens[pid]; }
};

// Geometry of one chunked scan (shared by count and fill kernels, and mirrored on the host).
// All positions are "virtual": v = haystack offset + base_mis, where base_mis = address
// misalignment of the haystack pointer w.r.t. 64 bytes, so that v % 64 == 0 <=> 64-byte aligned address (and
// v % 16 == 0 <=> 16-byte aligned: what the 16-byte loads of every kernel rely on).
struct ScanGeom {
    const uint8_t* hay16;     // haystack pointer rounded down to 64 B
    uint64_t base_mis;        // hay - hay16
    uint64_t cold_floor;      // v of span_start: the walk never looks left of it
    uint64_t emit_lo;         // chunks own `at` in [emit_lo, emit_hi)  (at = match end - 1)
    uint64_t emit_hi;
    uint64_t grid0;           // v of the chunk grid origin (multiple of chunk, <= emit_lo)
    uint32_t chunk;           // bytes per lane-chunk (multiple of 64)
    uint32_t halo;            // max_pattern_len - 1
    uint64_t n_chunks;
    uint32_t emit_start_matches;  // start-state (empty pattern) matches at span_start belong to chunk 0
#ifdef ACGPU_GUARD
    // bounds-checked debug build (make -C csrc guard -> libacgpu_guard.so): every haystack access of every kernel is
    // checked against the 16-byte-aligned hull of the searched span; violations are counted, not faulted on
    unsigned long long* guard;    // device counter (acgpu_guard_violations)
    uint64_t guard_lo, guard_hi;  // v in [guard_lo, guard_hi) may be read
#endif
};

#ifdef ACGPU_GUARD
#define ACGPU_HAY_CHECK(g, p, n)                                                                               \
    do {                                                                                                       \
        if ((g).guard && (uint64_t(p) < (g).guard_lo || uint64_t(p) + uint64_t(n) > (g).guard_hi)) atomicAdd((g).guard, 1ull); \
    } while (0)
#else
#define ACGPU_HAY_CHECK(g, p, n) ((void)0)
#endif

struct ChunkRange {
    uint64_t w, lo, hi;  // walk from w (start state), count/emit for at in [lo, hi)
};

__host__ __device__ __forceinline__ ChunkRange chunk_range(const ScanGeom& g, uint64_t ci) {
    ChunkRange r;
    uint64_t glo = g.grid0 + ci * uint64_t(g.chunk);
    uint64_t ghi = glo + g.chunk;
    r.lo = glo > g.emit_lo ? glo : g.emit_lo;
    r.hi = ghi < g.emit_hi ? ghi : g.emit_hi;
    uint64_t w = r.lo >= g.halo ? r.lo - g.halo : 0;
    r.w = w > g.cold_floor ? w : g.cold_floor;
    return r;
}

}  // namespace acgpu
