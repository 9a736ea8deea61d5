// Prefix-filter count engine for the Standard/unanchored overlapping scan (gfx950).
//
// The count pass only has to produce, exactly, the number of matches whose last byte falls into each
// output chunk.  The set of overlapping matches is "every occurrence of every pattern"
// (src/automaton.rs:1491-1534 visits the match list of every entered match state; reference
// DESIGN.md:60-63), so it can be enumerated by START position instead of by carrying the automaton
// state from byte to byte -- which removes the serial dependency of the transition walk:
//
//   level 1  (every haystack position, ~4.75 VALU + 1 LDS gather, no cross-position dependency)
//       a 64 KiB LDS Bloom table, addressed by a hash of the 3 bytes at the position, holds 32-bit words
//       whose bit (31 - (b3 & 31)) says "some pattern may start with these three bytes followed by b3"
//       (all-ones where a pattern of length <= 3 starts).  ~0.2 % of the positions of a random haystack
//       survive for the 1k-pattern set -- almost all of them Bloom false positives.
//   level 2  (survivors, 64 at a time)   exact test of the first three bytes against the LDS-resident
//       bigram table T[b0-lo][b1-lo] = {continuation bytes, "always verify"}; kills the false positives.
//   level 3  (survivors, 64 at a time)   exact walk of the trie-only (anchored) transition table in
//       global memory from the start state; every pattern end is credited to the chunk owning its end.
//
// Survivors move between the levels through per-wavefront LDS queues filled with __ballot/__popcll
// prefix ranks, so levels 2 and 3 always run with full wavefronts.  HBM is read exactly once, fully
// coalesced (lane l loads 16 B at row + 16 l).  The filter has no false negatives by construction and
// every survivor is verified exactly, so the counts are exact for every input.  Unavailable (the host
// falls back to the transition-walk engines) when a pattern is empty, the first two trie levels span
// more than ~170 byte values, or the automaton has > 32767 states.
//
// VALU budget (measured, scripts/ubench/valu_rate.hip): integer shifts / mul24 / alignbit / min / bfe
// / SDWA forms issue at 4 cycles per wavefront-instruction per SIMD on gfx950, add / xor / bitop3 at 2.
// Level 1 is therefore written as alignbit -> mul_u32_u24 -> and (WORD_1) -> ds_read_b32 -> lshl -> alignbit.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hot.hpp"

// PF_EXP: bit mask of timing experiments (scripts/pf_variants.sh); 0 in the product build.
//   1 = no survivor handling   2 = no LDS gathers   4 = conflict-free gathers   8 = no level 1 at all
#ifndef PF_EXP
#define PF_EXP 0
#endif

namespace acgpu {

namespace {

constexpr int kPfBlock = 1024;
constexpr int kPfWaves = kPfBlock / 64;
constexpr int kQueue = 128;           // per-wave survivor queues (drained in batches of 64)
constexpr uint32_t kRowBytes = 1008;  // one wave-row: 63 lanes x 16 B of start positions (lane 63 only supplies
                                      // the 4-byte look-ahead of lane 62 and repeats as lane 0 of the next row)
constexpr uint32_t kTaskRows = 32;    // rows per wave task (about one 64-entry batch of level-1 survivors on random text)
constexpr uint32_t kBitsBytes = 64 * 1024;  // level-1 Bloom table (static LDS at offset 0: no base add per gather)

struct PfArgs {
    const uint32_t* bits;   // level-1 Bloom table (global copy)
    const uint32_t* T;      // level-2 bigram table (global copy)
    const uint16_t* atab;
    const uint32_t* own_cnt;
    uint32_t bits_bytes, w1, lo, root;
    uint64_t scan_lo;     // first start position that may begin an owned match (virtual)
    uint64_t row0;        // scan_lo rounded down to 16
    uint64_t hull_end;    // emit_hi rounded up to 16: no load touches bytes at or beyond it
    uint64_t n_tasks;
};

__device__ __forceinline__ void pf_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// level 3: exact verification of one start position: trie-only walk, credit every pattern end to its chunk
__device__ __forceinline__ void pf_verify(const PfArgs& a, const ScanGeom& g, uint32_t* counts, uint64_t v) {
    uint32_t s = a.root;
    for (uint64_t at = v; at < g.emit_hi; at++) {
        const uint32_t e = a.atab[(s << 8) | g.hay16[at]];
        if (e == 0) break;
        s = e & 0x7FFFu;
        if ((e & 0x8000u) && at >= g.emit_lo)
            atomicAdd(&counts[(at - g.grid0) / g.chunk], a.own_cnt[s]);
    }
}

// level 2: exact test of the first three bytes (key = b0 | b1 << 8 | b2 << 16, taken from the lane's registers)
// against the bigram table in LDS
__device__ __forceinline__ bool pf_exact(const PfArgs& a, const uint32_t* s_T, uint32_t key) {
    const uint32_t W = a.w1 - 1;
    uint32_t x = (key & 0xFFu) - a.lo, y = ((key >> 8) & 0xFFu) - a.lo;
    const uint32_t b2 = (key >> 16) & 0xFFu;
    x = x < W ? x : W;
    y = y < W ? y : W;
    const uint32_t ent = s_T[x * a.w1 + y];
    return ((ent & 0xFFFFu) == b2) | (((ent >> 16) & 0x7FFFu) == b2) | (int32_t(ent) < 0);
}

// Per-wavefront state of the filter pipeline.
struct PfWave {
    const PfArgs& a;
    const ScanGeom& g;
    uint32_t* counts;
    const uint32_t* s_bits;  // level-1 bit table (static LDS)
    const uint32_t* s_T;
    uint2* q1;           // level-1 survivors: {offset from the task base, key bytes b0 b1 b2 b3}
    uint64_t* q2;        // level-2 survivors: absolute (virtual) start positions
    uint64_t task_base = 0;
    uint32_t q1count = 0, q2count = 0;  // wave-uniform fill levels; a batch = the LAST (up to) 64 entries (order is irrelevant)
    int lane = 0;
    uint32_t amask = 0;

    __device__ __forceinline__ void drain_q2(uint32_t n) {
        pf_fence();
        q2count = uni(q2count - n);
        uint64_t v = 0;
        if (uint32_t(lane) < n) v = q2[q2count + lane];
        pf_fence();
        if (uint32_t(lane) < n) pf_verify(a, g, counts, v);
    }
    static __device__ __forceinline__ uint32_t uni(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }

    // level 1 over 16 start positions held in wd[0..4] (16 bytes + 4 look-ahead), in two halves of 8 positions:
    // `probe` issues the 8 LDS gathers of a half, `fold` appends their survivor bits below `hits` (each position
    // shifts the mask left by one).  Split so that the gathers of the next half are in flight while the previous
    // half is folded (LDS pipe and VALU overlap inside one wave, not only across waves).
    // K(o) = a dword whose low byte is b[o] (for o >= 16 only the low 5 bits are consumed)
    static __device__ __forceinline__ uint32_t K(const uint32_t (&wd)[5], int o) {
        if (o > 16) return wd[4] >> (8 * (o - 16));
        return (o & 3) == 0 ? wd[o >> 2] : __builtin_amdgcn_alignbit(wd[(o >> 2) + 1], wd[o >> 2], 8 * (o & 3));
    }
    template <int H>
    __device__ __forceinline__ void probe(const uint32_t (&wd)[5], uint32_t (&word)[8]) const {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t h = pf_hash(K(wd, 8 * H + k)) & amask;  // hash of b[k..k+2]
            if (PF_EXP & 2) word[k] = h;
            else if (PF_EXP & 4) word[k] = s_bits[((h & 0xFF00u) | (uint32_t(lane) << 2)) >> 2];
            else word[k] = s_bits[h >> 2];
        }
    }
    template <int H>
    __device__ __forceinline__ uint32_t fold(uint32_t hits, const uint32_t (&wd)[5], const uint32_t (&word)[8]) const {
#pragma unroll
        for (int k = 0; k < 8; k++)
            hits = __builtin_amdgcn_alignbit(hits, word[k] << (K(wd, 8 * H + k + 3) & 31), 31);  // probe bit b[k+3]
        return hits;
    }
    __device__ __forceinline__ uint32_t level1_pair(const uint32_t (&w0)[5], const uint32_t (&w1)[5]) const {
        uint32_t A[8], B[8], hits = 0;
        probe<0>(w0, A);
        probe<1>(w0, B);
        __builtin_amdgcn_sched_barrier(0);
        hits = fold<0>(hits, w0, A);
        probe<0>(w1, A);
        __builtin_amdgcn_sched_barrier(0);
        hits = fold<1>(hits, w0, B);
        probe<1>(w1, B);
        __builtin_amdgcn_sched_barrier(0);
        hits = fold<0>(hits, w1, A);
        hits = fold<1>(hits, w1, B);
        return hits;
    }

    // the 4-byte window b[k..k+3] of a lane's row registers, k = 0..15 dynamic (cndmask tree + funnel shift)
    static __device__ __forceinline__ uint32_t window(const uint32_t (&wd)[5], uint32_t k) {
        const bool up = (k & 8u) != 0, mid = (k & 4u) != 0;
        const uint32_t c0 = up ? wd[2] : wd[0], c1 = up ? wd[3] : wd[1], c2 = up ? wd[4] : wd[2];
        return __builtin_amdgcn_alignbit(mid ? c2 : c1, mid ? c1 : c0, 8u * (k & 3u));
    }

    // level-1 survivors of one row (bit 15-k of h16 <=> start  task_base + off + k).  The divergent part is kept
    // minimal: each lane that has one peels its lowest survivor, takes its key bytes from the row registers (no
    // re-read of the haystack) and appends {offset, key} to the wave's queue at its ballot rank; levels 2 and 3
    // then run on dense batches of 64.
    __device__ __forceinline__ void survivors(uint32_t h16, const uint32_t (&wd)[5], uint32_t off) {
        while (__any(h16 != 0)) {
            const bool has = h16 != 0;
            const uint32_t k = (15u - uint32_t(__builtin_ctz(h16 | 0x10000u))) & 15u;
            h16 &= h16 - 1;
            const unsigned long long m = __ballot(has);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
            if (has) q1[q1count + rank] = make_uint2(off + k, window(wd, k));
            q1count = uni(q1count + uint32_t(__popcll(m)));
            if (q1count >= 64) drain_q1(64);
        }
    }

    // level 2 on one dense batch: exact test of the queued key bytes against the bigram table
    __device__ __forceinline__ void drain_q1(uint32_t n) {
        pf_fence();
        q1count = uni(q1count - n);
        uint2 e = make_uint2(0, 0);
        bool ok = false;
        if (uint32_t(lane) < n) { e = q1[q1count + lane]; ok = pf_exact(a, s_T, e.y); }
        pf_fence();
        if (__any(ok)) {
            const unsigned long long m = __ballot(ok);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
            if (ok) q2[q2count + rank] = task_base + e.x;
            q2count = uni(q2count + uint32_t(__popcll(m)));
            if (q2count >= 64) drain_q2(64);
        }
    }

    // one task = kTaskRows rows, processed two rows per step.  GUARD = per-lane bounds / ownership checks
    // (only the first and last tasks of a scan need them).
    template <bool GUARD>
    __device__ __forceinline__ void run_task(uint64_t task_base) {
        auto load = [&](uint64_t p, uint4& w) {
            if (GUARD) {
                w = make_uint4(0, 0, 0, 0);
                if (p < a.hull_end) w = *reinterpret_cast<const uint4*>(g.hay16 + p);
            } else {
                w = *reinterpret_cast<const uint4*>(g.hay16 + p);
            }
        };
        this->task_base = task_base;
        uint64_t p = task_base + uint64_t(lane) * 16;
        uint32_t off = uint32_t(lane) * 16;  // p - task_base
        // two register sets (A, B) in ping-pong: while one pair of rows is filtered the other is in flight,
        // and no register copies are needed to free the load destinations
        uint4 a0, a1, b0, b1;
        load(p, a0);
        load(p + kRowBytes, a1);
        auto pair = [&](const uint4& wa, const uint4& wb) {
            // 4-byte look-ahead = first dword of the right neighbour lane (DPP wave shift, no memory traffic)
            const uint32_t w0[5] = {wa.x, wa.y, wa.z, wa.w, uint32_t(__builtin_amdgcn_update_dpp(0, int(wa.x), 0x130, 0xF, 0xF, false))};
            const uint32_t w1[5] = {wb.x, wb.y, wb.z, wb.w, uint32_t(__builtin_amdgcn_update_dpp(0, int(wb.x), 0x130, 0xF, 0xF, false))};
            uint32_t hits = (PF_EXP & 8) ? uint32_t((w0[0] ^ w1[1] ^ w0[2] ^ w1[3] ^ w0[4] ^ w1[4]) == 0x12345678u) : level1_pair(w0, w1);
            if (PF_EXP & 1) hits = hits == 0x9E3779B9u;
            if (lane == 63) hits = 0;  // lane 63's 16 bytes are lane 0 of the next row
            if (GUARD) {  // positions outside [scan_lo, emit_hi) never start an owned match
#pragma unroll
                for (int i = 0; i < 32; i++) {
                    const uint64_t v = p + uint64_t((i >> 4) * kRowBytes + (i & 15));
                    if (!(v >= a.scan_lo && v < g.emit_hi)) hits &= ~(0x80000000u >> i);
                }
            }
            if (__any(hits != 0)) {
                survivors(hits >> 16, w0, off);
                survivors(hits & 0xFFFFu, w1, off + kRowBytes);
            }
            p += 2 * kRowBytes;
            off += 2 * kRowBytes;
        };
        static_assert(kTaskRows % 4 == 0, "two row pairs per iteration");
#pragma unroll 1
        for (uint32_t r = 0; r < kTaskRows; r += 4) {
            if (GUARD && task_base + uint64_t(r) * kRowBytes >= g.emit_hi) break;  // wave-uniform
            load(p + 2 * kRowBytes, b0);
            load(p + 3 * kRowBytes, b1);
            pair(a0, a1);
            if (r + 4 < kTaskRows) {
                load(p + 2 * kRowBytes, a0);
                load(p + 3 * kRowBytes, a1);
            }
            pair(b0, b1);
        }
        if (q1count) drain_q1(q1count);  // queue offsets are relative to this task
    }
};

__global__ __launch_bounds__(kPfBlock) void k_pf_count(PfArgs a, ScanGeom g, uint32_t* __restrict__ counts) {
    // LDS: static [bit table], dynamic [bigram table | per-wave level-3 queues]
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[kBitsBytes / 4];
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_T = reinterpret_cast<uint32_t*>(smem);
    const uint32_t tsz = a.w1 * a.w1;
    uint64_t* s_q = reinterpret_cast<uint64_t*>(smem + ((size_t(tsz) * 4 + 15) & ~size_t(15)));
    for (uint32_t i = threadIdx.x; i < kBitsBytes / 4; i += kPfBlock) s_bits[i] = a.bits[i];
    for (uint32_t i = threadIdx.x; i < tsz; i += kPfBlock) s_T[i] = a.T[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PfWave st{a, g, counts, s_bits, s_T, reinterpret_cast<uint2*>(s_q + wave * (2 * kQueue)), s_q + wave * (2 * kQueue) + kQueue};
    st.lane = lane;
    st.amask = (kBitsBytes - 1) & ~3u;

    const uint64_t task_bytes = uint64_t(kTaskRows) * kRowBytes;
    const uint64_t wave_id = uint64_t(blockIdx.x) * kPfWaves + wave;
    const uint64_t n_waves = uint64_t(gridDim.x) * kPfWaves;
    for (uint64_t task = wave_id; task < a.n_tasks; task += n_waves) {
        const uint64_t task_base = a.row0 + task * task_bytes;
        // interior task: every start position is owned and every load (incl. 4-byte look-ahead) is in bounds
        const bool interior = task_base >= a.scan_lo && task_base + task_bytes + 16 <= a.hull_end &&
                              task_base + task_bytes <= g.emit_hi;  // (+16: lane 63 of the last row)
        if (interior) st.run_task<false>(task_base);
        else st.run_task<true>(task_base);
    }
    // final partial batch of level 3
    while (st.q2count) st.drain_q2(st.q2count < 64 ? st.q2count : 64);
}

}  // namespace

hipError_t launch_pf_count(const HotTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s) {
    PfArgs a{};
    a.bits = h.pf_bits; a.T = h.pf_T; a.atab = h.atab; a.own_cnt = h.own_cnt;
    a.bits_bytes = h.pf_bits_bytes; a.w1 = h.pf_w1; a.lo = h.pf_lo; a.root = h.start;
    const uint64_t lo = g.emit_lo >= g.halo ? g.emit_lo - g.halo : 0;
    a.scan_lo = lo > g.cold_floor ? lo : g.cold_floor;
    a.row0 = a.scan_lo & ~uint64_t(15);
    a.hull_end = (g.emit_hi + 15) & ~uint64_t(15);
    const uint64_t task_bytes = uint64_t(kTaskRows) * kRowBytes;
    a.n_tasks = g.emit_hi > a.row0 ? (g.emit_hi - a.row0 + task_bytes - 1) / task_bytes : 0;
    hipError_t e = hipMemsetAsync(counts, 0, g.n_chunks * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (a.n_tasks == 0) return hipSuccess;
    if (a.bits_bytes != kBitsBytes) return hipErrorInvalidValue;
    const size_t smem = ((size_t(a.w1) * a.w1 * 4 + 15) & ~size_t(15)) + size_t(kPfWaves) * 2 * kQueue * sizeof(uint64_t);
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pf_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - int(kBitsBytes));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const uint64_t blocks_per_cu = std::max<uint64_t>(1, std::min<uint64_t>(2, (160 * 1024) / (smem + kBitsBytes + 1024)));
    uint64_t blocks = uint64_t(cus) * blocks_per_cu;
    const uint64_t need = (a.n_tasks + kPfWaves - 1) / kPfWaves;
    if (blocks > need) blocks = need;
    k_pf_count<<<dim3(uint32_t(blocks)), dim3(kPfBlock), smem, s>>>(a, g, counts);
    return hipGetLastError();
}

}  // namespace acgpu
