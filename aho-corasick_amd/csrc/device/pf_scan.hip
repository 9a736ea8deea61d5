// Prefix-filter count engine for the Standard/unanchored overlapping scan (gfx950).
//
// The count pass only has to produce, exactly, the number of matches whose last byte falls into each
// output chunk.  The set of overlapping matches is "every occurrence of every pattern"
// (src/automaton.rs:1491-1534 visits the match list of every entered match state; DESIGN.md of the
// reference :60-63), so it can be enumerated by START position instead of by walking the automaton
// state byte by byte:
//
//   fast path (every haystack position i, no cross-position dependency)
//       one LDS gather  T[b_i - lo][b_{i+1} - lo]  describes the trie node root->b_i->b_{i+1}
//       (which third bytes continue it, whether a 1-/2-byte pattern ends there); position i survives
//       only if b_{i+2} continues the node or the node demands verification (~0.1 % of the positions
//       of a random haystack for the 1k-pattern set).
//   slow path (survivors only, compacted)
//       survivors are appended to a per-wavefront LDS queue with __ballot/__popcll prefix ranks and
//       verified 64 at a time: an exact walk of the trie-only (anchored) transition table from the
//       start state, adding the number of patterns ending in each visited node to the chunk that owns
//       the end position.
//
// HBM is read exactly once, fully coalesced (lane l loads 16 B at row + 16 l); the only other global
// traffic is the verification walk (L2 resident, rare).  Results are exact for every input: the
// filter has no false negatives by construction and every survivor is verified.  Unavailable (host
// falls back to the transition-walk engines) when a pattern is empty, the first two trie levels span
// more than ~170 byte values, or the automaton has > 32767 states.
#include <hip/hip_runtime.h>

#include "hot.hpp"

namespace acgpu {

namespace {

constexpr int kPfBlock = 512;
constexpr int kPfWaves = kPfBlock / 64;
constexpr int kQueue = 128;           // per-wave survivor queue (drained in batches of 64)
constexpr uint32_t kRowBytes = 1024;  // one wave-row: 64 lanes x 16 B
constexpr uint32_t kTaskRows = 16;    // rows per wave task (16 KiB)

struct PfArgs {
    const uint32_t* T;
    const uint16_t* atab;
    const uint32_t* own_cnt;
    uint32_t w1, lo, root;
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

// exact verification of one start position: trie-only walk, credit every pattern end to its chunk
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

__global__ __launch_bounds__(kPfBlock) void k_pf_count(PfArgs a, ScanGeom g, uint32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_T = reinterpret_cast<uint32_t*>(smem);
    const uint32_t tsz = a.w1 * a.w1;
    uint64_t* s_q = reinterpret_cast<uint64_t*>(smem + ((size_t(tsz) * 4 + 15) & ~size_t(15)));
    for (uint32_t i = threadIdx.x; i < tsz; i += kPfBlock) s_T[i] = a.T[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t* q = s_q + wave * kQueue;
    uint32_t qhead = 0, qcount = 0;  // wave-uniform
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t W = a.w1 - 1;

    const uint64_t wave_id = uint64_t(blockIdx.x) * kPfWaves + wave;
    const uint64_t n_waves = uint64_t(gridDim.x) * kPfWaves;
    for (uint64_t task = wave_id; task < a.n_tasks; task += n_waves) {
        const uint64_t task_base = a.row0 + task * uint64_t(kTaskRows) * kRowBytes;
        // software pipeline: the next row's 16+4 bytes are in flight while the current row is filtered
        uint4 w_next = make_uint4(0, 0, 0, 0);
        uint32_t nx_next = 0;
        {
            const uint64_t p0 = task_base + uint64_t(lane) * 16;
            if (p0 < a.hull_end) w_next = *reinterpret_cast<const uint4*>(g.hay16 + p0);
            if (p0 + 16 < a.hull_end) nx_next = *reinterpret_cast<const uint32_t*>(g.hay16 + p0 + 16);
        }
        for (uint32_t r = 0; r < kTaskRows; r++) {
            const uint64_t row = task_base + uint64_t(r) * kRowBytes;
            if (row >= g.emit_hi) break;  // wave-uniform
            const uint64_t p = row + uint64_t(lane) * 16;
            const uint4 w = w_next;
            const uint32_t nx = nx_next;
            if (r + 1 < kTaskRows) {
                const uint64_t pn = p + kRowBytes;
                w_next = make_uint4(0, 0, 0, 0);
                nx_next = 0;
                if (pn < a.hull_end) w_next = *reinterpret_cast<const uint4*>(g.hay16 + pn);
                if (pn + 16 < a.hull_end) nx_next = *reinterpret_cast<const uint32_t*>(g.hay16 + pn + 16);
            }
            const uint32_t wd[5] = {w.x, w.y, w.z, w.w, nx};
            // ---- straight-line filter over the lane's 16 start positions (16 independent LDS gathers).
            // Branch- and compare-free: per position  m = min3(e1 ^ c2, e2 ^ c2, entry)  as signed ints is
            // <= 0 exactly when the third byte continues the node (a xor is 0) or the entry carries the
            // "always verify" sign bit; (m - 1) >> 31 is then shifted into the lane's hit mask (v_alignbit).
            uint32_t x4[17];
#pragma unroll
            for (int k = 0; k < 17; k++) {
                const uint32_t x = ((wd[k >> 2] >> (8 * (k & 3))) & 0xFFu) - a.lo;
                x4[k] = (x < W ? x : W) << 2;  // unsigned: bytes below lo wrap and clamp to index W; x4 = column byte offset
            }
            uint32_t hits = 0;  // bit (15 - k) <=> start position k survives
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t ent = *reinterpret_cast<const uint32_t*>(smem + __umul24(x4[k], a.w1) + x4[k + 1]);
                const uint32_t c2 = (wd[(k + 2) >> 2] >> (8 * ((k + 2) & 3))) & 0xFFu;
                const int32_t d1 = int32_t((ent & 0xFFFFu) ^ c2);
                const int32_t d2 = int32_t((ent >> 16) ^ c2);   // polluted by the sign bit only when the entry is negative anyway
                int32_t m = d1 < d2 ? d1 : d2;
                m = m < int32_t(ent) ? m : int32_t(ent);
                hits = __builtin_amdgcn_alignbit(hits, uint32_t(m - 1), 31);
            }
            // positions outside [scan_lo, emit_hi) never start an owned match (first / last row only)
            if (!(row >= a.scan_lo && row + kRowBytes <= g.emit_hi)) {  // wave-uniform
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (!(p + k >= a.scan_lo && p + k < g.emit_hi)) hits &= ~(1u << (15 - k));
            }
            // ---- survivors: compact into the wave queue (lane order per round), verify 64 at a time
            while (__any(hits != 0)) {
                const bool has = hits != 0;
                const uint32_t j = has ? 31u - uint32_t(__builtin_clz(hits)) : 0u;  // highest bit = smallest position
                const uint32_t k = 15u - j;
                hits &= ~(1u << j);
                const unsigned long long m = __ballot(has);
                if (has) q[(qhead + qcount + uint32_t(__popcll(m & lt_mask))) & (kQueue - 1)] = p + k;
                qcount += uint32_t(__popcll(m));
                if (qcount >= 64) {
                    pf_fence();
                    const uint64_t v = q[(qhead + lane) & (kQueue - 1)];
                    pf_fence();
                    qhead = (qhead + 64) & (kQueue - 1);
                    qcount -= 64;
                    pf_verify(a, g, counts, v);
                }
            }
        }
    }
    if (qcount) {  // final partial batch
        pf_fence();
        if (uint32_t(lane) < qcount) pf_verify(a, g, counts, q[(qhead + lane) & (kQueue - 1)]);
    }
}

}  // namespace

hipError_t launch_pf_count(const HotTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s) {
    PfArgs a{};
    a.T = h.pf_T; a.atab = h.atab; a.own_cnt = h.own_cnt;
    a.w1 = h.pf_w1; a.lo = h.pf_lo; a.root = h.start;
    const uint64_t lo = g.emit_lo >= g.halo ? g.emit_lo - g.halo : 0;
    a.scan_lo = lo > g.cold_floor ? lo : g.cold_floor;
    a.row0 = a.scan_lo & ~uint64_t(15);
    a.hull_end = (g.emit_hi + 15) & ~uint64_t(15);
    const uint64_t task_bytes = uint64_t(kTaskRows) * kRowBytes;
    a.n_tasks = g.emit_hi > a.row0 ? (g.emit_hi - a.row0 + task_bytes - 1) / task_bytes : 0;
    hipError_t e = hipMemsetAsync(counts, 0, g.n_chunks * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (a.n_tasks == 0) return hipSuccess;
    const size_t smem = ((size_t(a.w1) * a.w1 * 4 + 15) & ~size_t(15)) + size_t(kPfWaves) * kQueue * sizeof(uint64_t);
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pf_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const uint64_t blocks_per_cu = std::max<uint64_t>(1, std::min<uint64_t>(8, (160 * 1024) / (smem + 1024)));
    uint64_t blocks = uint64_t(cus) * blocks_per_cu;
    const uint64_t need = (a.n_tasks + kPfWaves - 1) / kPfWaves;
    if (blocks > need) blocks = need;
    k_pf_count<<<dim3(uint32_t(blocks)), dim3(kPfBlock), smem, s>>>(a, g, counts);
    return hipGetLastError();
}

}  // namespace acgpu
