// Contiguous-NFA failure-link walk (k_cnfa_count): the reference's contiguous::NFA::next_state
// (src/nfa/contiguous.rs:186-247) over the reference's own `repr` words, one haystack lane-chunk per wavefront lane
// restructured around what bounds it on gfx950: dependent L2 gathers at ~230 G lane-gathers/s chip-wide.
//
// The literal loop costs ~5 dependent gathers per byte on random text at 100 000 patterns (state header -> transition ->
// fail link -> the fail target's header -> its transition; SURVEY.md Appendix C: ~1 failure hop per byte).  Here
//   * one step loads the current state's first four words {header, fail, two data words} in one gather and -- only while
//     some dense state lives outside LDS -- the dense-layout transition word repr[sid + 2 + class] beside it,
//     speculatively, before the header says the state is dense (for a KIND_ONE / sparse state that word is ignored;
//     `repr` is padded on the device so that it is always in bounds).  Dense state: the transition word decides;
//     KIND_ONE: header class + data word decide; both without a further load.
//   * the start state and its children -- the fail targets of almost every step -- are held in LDS (fail word + dense row
//     each), and every fail word / transition target that names one of them reads `0x80000000 | slot` in the device's
//     copy of `repr` (CnfaHotDev::repr_t): the second half of a typical step (fail -> distance-1 state -> transition)
//     never leaves the CU and costs a bit test, an LDS read and a compare (the first version hashed the state id into a
//     4 096-entry LDS table at every hop: a third of the instructions of a step that is issue-bound);
//   * second tier: the dense states that do not fit LDS (the grandchildren of the start state: where 89 % of the steps of
//     the 100 000-pattern automaton begin) have their rows in a table of their own and are named `0x40000000 | index`;
//     their fail words and match counts sit in LDS -- a step from one of them is ONE gather, the state word says what the
//     header would (87 -> 131 GB/s at 100 000 patterns);
//   * a sparse state's classes are in ascending order (checked at upload): the packed-class scan stops at the first
//     larger class, and header | fail | the first eight classes arrive in ONE 16-byte gather;
//   * two 1 024-thread workgroups per CU, every lane reading its own lane-chunk in 16-byte pieces (no LDS staging):
//     2 048 dependent chains per CU instead of 256.
// ~1.1 gathers per byte instead of ~5.  Host tables: host/cnfa_tables.cpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <unordered_map>
#include <cstdlib>
#include <vector>

#include "cnfa_walk.hpp"
#include "../host/cnfa_tables.hpp"
#include "launch_util.hpp"
#include "tile_walk.hpp"

namespace acgpu {

namespace {

constexpr int kCwWaves = 16;
constexpr int kCwBlock = kCwWaves * 64;

constexpr uint32_t kCwTag = kCnfaSlotTag;   // state word = kCwTag | slot: the state lives in LDS (CnfaHotDev::repr_t)

struct CnfaFastStep {
    CnfaEng eng;                 // c.repr = the PATCHED copy (references to LDS-resident states are tagged); class map in LDS
    const uint32_t* s_rows;      // LDS [n_slots][alen + 1]: word 0 = fail state, then the dense transitions (tagged likewise)
    const uint32_t* s_mcnt;      // LDS [n_slots]: match-list length of the slot's state (0: not a match state)
    const uint32_t* s_mfail;     // LDS [n_mid]: fail state of a second-tier state
    const uint16_t* s_mmcnt;     // LDS [n_mid]: its match-list length
    const uint32_t* mid_rows;    // global [n_mid][1 << mid_shift]
    uint32_t mid_shift, row_words;
    uint32_t sid, cnt;           // sid: a repr offset, or kCwTag | slot
    bool alive;
    bool dense_outside, sorted_sparse, slot_matches, mid_matches;   // wave-uniform (CnfaHotDev)

    // index 0..3 of the byte of w that equals k, 4 if none (SWAR zero-byte test on w ^ kkkk; the lowest flagged byte is exact)
    static __device__ __forceinline__ uint32_t byte_index(uint32_t w, uint32_t k4) {
        const uint32_t x = w ^ k4;
        const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;
        return z ? uint32_t(__builtin_ctz(z)) >> 3 : 4u;
    }
    __device__ __forceinline__ void step(uint8_t byte, bool owned) {
        const uint32_t* repr = eng.c.repr;
        const uint32_t k = eng.cls[byte];
        const uint32_t k4 = k * 0x01010101u;
        uint32_t o = sid;
        for (;;) {
            if (o & kCwTag) {   // a state in LDS: fail word and row
                const uint32_t row = (o & 0xFFFFu) * row_words;
                const uint32_t nx = s_rows[row + 1 + k];
                if (nx != kDevFail) { o = nx; break; }
                o = s_rows[row];
                continue;
            }
            if (o & kCnfaMidTag) {   // a second-tier dense state: ONE gather (its transition); fail word from LDS
                const uint32_t idx = o & (kCnfaMidTag - 1);
                const uint32_t nx = mid_rows[(idx << mid_shift) + k];
                if (nx != kDevFail) { o = nx; break; }
                o = s_mfail[idx];
                continue;
            }
            // header | fail | first two data words in ONE 16-byte gather (word-aligned); the dense-layout transition
            // beside it only while some dense state lives outside LDS
            // (named registers, no array: a dynamically indexed array goes to scratch memory)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
            const u32x4 h4 = *reinterpret_cast<const u32x4*>(repr + o);
            uint32_t dense_nx = kDevFail;
            if (dense_outside) dense_nx = repr[o + 2 + k];
            const uint32_t head = h4.x, fail = h4.y, d0 = h4.z, d1 = h4.w;
            const uint32_t kind = head & 0xFFu;
            if (kind == CnfaEng::KIND_DENSE) {
                if (!dense_outside) dense_nx = repr[o + 2 + k];   // (cannot happen: every dense state is in LDS)
                if (dense_nx != kDevFail) { o = dense_nx; break; }
            } else if (kind == CnfaEng::KIND_ONE) {
                if (k == ((head >> 8) & 0xFFu)) { o = d0; break; }
            } else {   // sparse: classes packed four per word, then the targets (contiguous.rs:224-243)
                const uint32_t tl = kind, cl = (tl + 3) >> 2;
                bool found = false;
                for (uint32_t i = 0; i < cl && !found; i++) {
                    const uint32_t w = i == 0 ? d0 : (i == 1 ? d1 : repr[o + 2 + i]);
                    const uint32_t j = byte_index(w, k4);
                    if (j < 4 && i * 4 + j < tl) {
                        const uint32_t t = cl + i * 4 + j;   // word index of the target behind the class words
                        o = t == 1 ? d1 : repr[o + 2 + t];   // (t >= 1)
                        found = true;
                    } else if (sorted_sparse && (w >> 24) > k && i * 4 + 3 < tl) {
                        break;   // ascending classes: everything further on is larger still
                    }
                }
                if (found) break;
            }
            o = fail;
        }
        sid = o;
        if (o & kCwTag) {
            if (slot_matches && owned) cnt += s_mcnt[o & 0xFFFFu];
        } else if (o & kCnfaMidTag) {
            if (mid_matches && owned) cnt += s_mmcnt[o & (kCnfaMidTag - 1)];
        } else if (eng.is_special(o)) {
            if (o == kDevDead) alive = false;
            else if (owned && eng.is_match(o)) cnt += eng.match_len(o);
        }
    }
};

// One lane-chunk per lane, haystack read by the lane itself in 16-byte pieces (one piece ahead): no LDS staging, so LDS
// holds only the cached rows (~42 KB at 96 classes) and TWO 1 024-thread workgroups share a CU -- 2 048 dependent chains
// per CU.  The walk is latency-bound at wavefront granularity: a wavefront's step lasts as long as its slowest lane
// (fail -> uncached state -> fail again: two gather rounds), and more resident wavefronts is what hides that.
__global__ __launch_bounds__(kCwBlock, 2) void k_cnfa_count(CnfaEng eng, CnfaHotDev hot, ScanGeom g, uint32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_rows = reinterpret_cast<uint32_t*>(smem);
    uint32_t* s_mcnt = s_rows + size_t(hot.n_slots) * hot.row_words;
    uint32_t* s_mfail = s_mcnt + hot.n_slots;
    uint16_t* s_mmcnt = reinterpret_cast<uint16_t*>(s_mfail + hot.n_mid);
    uint8_t* s_cls = reinterpret_cast<uint8_t*>(s_mmcnt + ((hot.n_mid + 1) & ~1u));
    for (uint32_t i = threadIdx.x; i < hot.n_slots * hot.row_words; i += kCwBlock) s_rows[i] = hot.rows[i];
    for (uint32_t i = threadIdx.x; i < hot.n_slots; i += kCwBlock) s_mcnt[i] = hot.mcnt[i];
    for (uint32_t i = threadIdx.x; i < hot.n_mid; i += kCwBlock) { s_mfail[i] = hot.mid_fail[i]; s_mmcnt[i] = hot.mid_mcnt[i]; }
    if (threadIdx.x < 256) s_cls[threadIdx.x] = eng.cls[threadIdx.x];
    __syncthreads();
    eng.cls = s_cls;
    eng.c.repr = hot.repr_t;

    const uint64_t ci = uint64_t(blockIdx.x) * kCwBlock + threadIdx.x;
    if (ci >= g.n_chunks) return;
    const ChunkRange r = chunk_range(g, ci);
    CnfaFastStep f{eng, s_rows, s_mcnt, s_mfail, s_mmcnt, hot.mid_rows, hot.mid_shift, hot.row_words, kCwTag | 0u, 0u, true,
                   hot.dense_outside != 0, hot.sorted_sparse != 0, hot.slot_matches != 0, hot.mid_matches != 0};   // slot 0 = the unanchored start state
    if (ci == 0 && g.emit_start_matches) f.cnt += s_mcnt[0];
    // the haystack in whole 64-byte sectors held in registers (a step takes microseconds: nothing to prefetch, and a
    // sector that is consumed at once does not sit in L2 between its pieces -- with 2 048 lanes per CU reading 16 bytes at
    // a time the open lines of an XCD exceeded its L2 and evicted the automaton); only 16-byte pieces of the aligned hull
    // of [walk start, chunk end) are touched
    for (uint64_t p = r.w & ~uint64_t(63); p < r.hi && f.alive; p += 64) {
        auto piece = [&](int q) {
            const uint64_t pv = p + uint32_t(16 * q);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (pv + 16 > r.w && pv < r.hi) { ACGPU_HAY_CHECK(g, pv, 16); v = *reinterpret_cast<const uint4*>(g.hay16 + pv); }
            return v;
        };
        const uint4 d0 = piece(0), d1 = piece(1), d2 = piece(2), d3 = piece(3);
#pragma unroll 1
        for (int q = 0; q < 4; q++) {   // (not unrolled: 64 copies of the step do not fit the instruction cache; selects, not an array)
            const uint4 dq = q == 0 ? d0 : (q == 1 ? d1 : (q == 2 ? d2 : d3));
            const uint32_t wds[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const uint64_t at = p + uint32_t(16 * q + b);
                if (at >= r.w && at < r.hi && f.alive) f.step(uint8_t(wds[b >> 2] >> (8 * (b & 3))), at >= r.lo);
            }
        }
    }
    counts[ci] = f.cnt;
}

}  // namespace

// Host tables: host/cnfa_tables.cpp (which states go to LDS, the patched copy of `repr`); uploaded here.
hipError_t build_cnfa_hot(const CNfa& c, CnfaHotTables& out) {
    out.ready = false;
    CnfaHotHost t;
    if (!build_cnfa_hot_host(c, t)) return hipSuccess;
    hipError_t e;
    auto up = [&](uint32_t** dst, const std::vector<uint32_t>& v) -> hipError_t {
        if (hipError_t er = hipMalloc(reinterpret_cast<void**>(dst), v.size() * 4); er != hipSuccess) return er;
        return hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    };
    if ((e = up(&out.dev.rows, t.rows)) != hipSuccess) return e;
    if ((e = up(&out.dev.mcnt, t.mcnt)) != hipSuccess) return e;
    if ((e = up(&out.dev.repr_t, t.repr_t)) != hipSuccess) return e;
    if (t.n_mid) {
        if ((e = up(&out.dev.mid_rows, t.mid_rows)) != hipSuccess) return e;
        if ((e = up(&out.dev.mid_fail, t.mid_fail)) != hipSuccess) return e;
        if ((e = hipMalloc(reinterpret_cast<void**>(&out.dev.mid_mcnt), t.mid_mcnt.size() * 2)) != hipSuccess) return e;
        if ((e = hipMemcpy(out.dev.mid_mcnt, t.mid_mcnt.data(), t.mid_mcnt.size() * 2, hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
    out.dev.n_mid = t.n_mid;
    out.dev.mid_shift = t.mid_shift;
    out.dev.mid_matches = t.mid_matches ? 1u : 0u;
    out.repr_words = c.repr.size();
    out.dev.dense_outside = t.dense_outside ? 1u : 0u;
    out.dev.sorted_sparse = t.sorted_sparse ? 1u : 0u;
    out.dev.slot_matches = t.slot_matches ? 1u : 0u;
    out.dev.n_slots = t.n_slots;
    out.dev.row_words = t.row_words;
    out.ready = true;
    return hipSuccess;
}

CnfaHotTables::~CnfaHotTables() {
    if (dev.rows) (void)hipFree(dev.rows);
    if (dev.mcnt) (void)hipFree(dev.mcnt);
    if (dev.repr_t) (void)hipFree(dev.repr_t);
    if (dev.mid_rows) (void)hipFree(dev.mid_rows);
    if (dev.mid_fail) (void)hipFree(dev.mid_fail);
    if (dev.mid_mcnt) (void)hipFree(dev.mid_mcnt);
}

hipError_t launch_cnfa_count(const CnfaHotTables& h, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts, hipStream_t s) {
    if (!h.ready) return hipErrorInvalidValue;
    const uint64_t blocks = (g.n_chunks + kCwBlock - 1) / kCwBlock;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    CnfaEng eng; eng.c = a.cnfa; eng.cls = a.cnfa.classes;
    size_t smem = size_t(h.dev.n_slots) * h.dev.row_words * 4 + size_t(h.dev.n_slots) * 4 + size_t(h.dev.n_mid) * 4 +
                  size_t((h.dev.n_mid + 1) & ~1u) * 2 + 256;
    // two workgroups per CU while the automaton is small (1 000 patterns: 255 -> 367 GB/s); a large one gains nothing --
    // its steps are issue-bound (~3 divergent loop trips per byte), and the second workgroup's open lines cost L2 hits
    // (100 000 patterns: 86 vs 81 GB/s) -- so it asks for more than half of the LDS and gets the CU to itself
    if (h.repr_words > (size_t(1) << 18)) smem = std::max<size_t>(smem, 84 * 1024);
    if (smem > 156 * 1024) return hipErrorInvalidValue;   // (kCnfaMaxMid keeps it below)
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_cnfa_count), 156 * 1024); e != hipSuccess) return e;
    k_cnfa_count<<<dim3(uint32_t(blocks)), dim3(kCwBlock), smem, s>>>(eng, h.dev, g, counts);
    return hipGetLastError();
}

}  // namespace acgpu
