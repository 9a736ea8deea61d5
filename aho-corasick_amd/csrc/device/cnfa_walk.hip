// Contiguous-NFA failure-link walk (k_cnfa_count): the reference's contiguous::NFA::next_state
// (src/nfa/contiguous.rs:186-247) over the reference's own `repr` words, one haystack lane-chunk per wavefront lane
// (tile_walk.hpp), restructured around what bounds it on gfx950: dependent L2 gathers at ~230 G lane-gathers/s chip-wide.
//
// The literal loop costs ~5 dependent gathers per byte on random text at 100 000 patterns (state header -> transition ->
// fail link -> the fail target's header -> its transition; SURVEY.md Appendix C: ~1 failure hop per byte).  Here
//   * one step issues TWO independent loads for the current state: its first three words {header, fail, first data word}
//     (12 bytes) and the dense-layout transition word repr[sid + 2 + class] -- speculatively, before the header says the
//     state is dense (for a KIND_ONE / sparse state that word is ignored; `repr` is padded on the device so that it is
//     always in bounds).  Dense state: the transition word decides; KIND_ONE: header class + data word decide; both
//     without a further load.  Sparse states (0.7 % of the 100k automaton) scan their packed classes like the reference.
//   * the start state and its children -- the fail targets of almost every step -- are held in LDS (fail word + dense row
//     each), found through a 4 096-entry LDS hash of the state id: the second half of a typical step (fail -> distance-1
//     state -> transition) never leaves the CU.
//   * 16 wavefronts per workgroup (one workgroup per CU): 1 024 dependent chains per CU instead of 256.
// ~2.2 gathers per byte instead of ~5: 50 -> ~100 GB/s; the bound is the gather rate, not occupancy.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "cnfa_walk.hpp"
#include "launch_util.hpp"
#include "tile_walk.hpp"

namespace acgpu {

namespace {

constexpr int kCwWaves = 16;
constexpr int kCwBlock = kCwWaves * 64;
constexpr uint32_t kCwHash = 4096;

struct CnfaFastStep {
    CnfaEng eng;                 // match lists (global), class map (LDS)
    const uint32_t* s_rows;      // LDS [n_slots][alen + 1]: word 0 = fail id, then the dense transitions
    const uint32_t* s_keys;      // LDS [n_slots]: state id of the slot
    const uint8_t* s_htab;       // LDS [kCwHash]: slot of a state id, 0xFF = none
    uint32_t hmul, row_words;
    uint32_t sid, cnt;
    bool alive;

    __device__ __forceinline__ uint32_t slot_of(uint32_t id) const {
        const uint32_t s = s_htab[(id * hmul) >> 20];
        return (s != 0xFFu && s_keys[s] == id) ? s : 0xFFu;
    }
    __device__ __forceinline__ void step(uint8_t byte, bool owned) {
        const uint32_t* repr = eng.c.repr;
        const uint32_t k = eng.cls[byte];
        uint32_t o = sid;
        for (;;) {
            const uint32_t slot = slot_of(o);
            if (slot != 0xFFu) {   // a cached dense state: fail word and row from LDS
                const uint32_t nx = s_rows[slot * row_words + 1 + k];
                if (nx != kDevFail) { o = nx; break; }
                o = s_rows[slot * row_words];
                continue;
            }
            // header | fail | first data word, and the dense-layout transition, in flight together
            const uint32_t head = repr[o], fail = repr[o + 1], data0 = repr[o + 2];
            const uint32_t dense_nx = repr[o + 2 + k];
            const uint32_t kind = head & 0xFFu;
            if (kind == CnfaEng::KIND_DENSE) {
                if (dense_nx != kDevFail) { o = dense_nx; break; }
            } else if (kind == CnfaEng::KIND_ONE) {
                if (k == ((head >> 8) & 0xFFu)) { o = data0; break; }
            } else {   // sparse: classes packed four per word, then the targets (contiguous.rs:224-243)
                const uint32_t tl = kind, cl = (tl + 3) >> 2;
                bool found = false;
                for (uint32_t i = 0; i < cl && !found; i++) {
                    const uint32_t w = i == 0 ? data0 : repr[o + 2 + i];
                    for (uint32_t j = 0; j < 4; j++)
                        if (((w >> (8 * j)) & 0xFFu) == k) { o = repr[o + 2 + cl + i * 4 + j]; found = true; break; }
                }
                if (found) break;
            }
            o = fail;
        }
        sid = o;
        if (eng.is_special(sid)) {
            if (sid == kDevDead) alive = false;
            else if (owned && eng.is_match(sid)) cnt += eng.match_len(sid);
        }
    }
};

__global__ __launch_bounds__(kCwBlock) void k_cnfa_count(CnfaEng eng, CnfaHotDev hot, ScanGeom g, uint32_t* __restrict__ counts,
                                                         uint32_t halo_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* s_tile = smem;                                                        // kCwWaves * 64 * kRow
    uint32_t* s_rows = reinterpret_cast<uint32_t*>(smem + size_t(kCwWaves) * 64 * kRow);
    uint32_t* s_keys = s_rows + size_t(hot.n_slots) * hot.row_words;
    uint8_t* s_htab = reinterpret_cast<uint8_t*>(s_keys + hot.n_slots);
    uint8_t* s_cls = s_htab + kCwHash;
    for (uint32_t i = threadIdx.x; i < hot.n_slots * hot.row_words; i += kCwBlock) s_rows[i] = hot.rows[i];
    for (uint32_t i = threadIdx.x; i < hot.n_slots; i += kCwBlock) s_keys[i] = hot.keys[i];
    for (uint32_t i = threadIdx.x; i < kCwHash; i += kCwBlock) s_htab[i] = hot.htab[i];
    if (threadIdx.x < 256) s_cls[threadIdx.x] = eng.cls[threadIdx.x];
    __syncthreads();
    eng.cls = s_cls;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_chunk0 = (uint64_t(blockIdx.x) * kCwWaves + wave) * 64;
    const uint64_t ci = wave_chunk0 + lane;
    const bool valid = ci < g.n_chunks;
    CnfaFastStep f{eng, s_rows, s_keys, s_htab, hot.hmul, hot.row_words, eng.start(false), 0u, valid};
    if (valid && ci == 0 && g.emit_start_matches && eng.is_match(f.sid)) f.cnt += eng.match_len(f.sid);
    tile_walk(g, halo_tiles, s_tile + size_t(wave) * 64 * kRow, wave_chunk0, lane, f);
    if (valid) counts[ci] = f.cnt;
}

}  // namespace

// Host: which states go to LDS -- the start state and its children while they are dense and LDS lasts -- and a
// collision-free multiplicative hash of their ids.
hipError_t build_cnfa_hot(const CNfa& c, CnfaHotTables& out) {
    out.ready = false;
    const uint32_t alen = uint32_t(c.alphabet_len);
    const uint32_t row_words = alen + 1;
    const std::vector<uint32_t>& r = c.repr;
    const uint32_t start = c.special.start_unanchored_id;
    if (r.empty() || start == 0 || (r[start] & 0xFFu) != 0xFFu) return hipSuccess;   // no unanchored start / not dense
    const size_t lds_budget = 160 * 1024 - size_t(kCwWaves) * 64 * kRow - kCwHash - 256 - 1024;
    const uint32_t max_slots = uint32_t(std::min<size_t>(254, lds_budget / (size_t(row_words) * 4 + 4)));
    if (max_slots < 1) return hipSuccess;
    std::vector<uint32_t> ids{start};
    for (uint32_t k = 0; k < alen && ids.size() < max_slots; k++) {
        const uint32_t t = r[start + 2 + k];
        if (t == 1 /*FAIL*/ || t == 0 || t == start) continue;
        if ((r[t] & 0xFFu) != 0xFFu) continue;                       // only dense records have the row layout
        if (std::find(ids.begin(), ids.end(), t) == ids.end()) ids.push_back(t);
    }
    uint32_t hmul = 0;
    std::vector<uint8_t> htab;
    for (uint32_t m = 0x9E3779B1u, tries = 0; tries < 4096; tries++, m += 0x61C88646u) {
        htab.assign(kCwHash, 0xFF);
        bool ok = true;
        for (size_t s = 0; s < ids.size() && ok; s++) {
            uint8_t& e = htab[(ids[s] * (m | 1u)) >> 20];
            if (e != 0xFF) ok = false; else e = uint8_t(s);
        }
        if (ok) { hmul = m | 1u; break; }
    }
    if (!hmul) return hipSuccess;
    std::vector<uint32_t> rows(ids.size() * row_words);
    for (size_t s = 0; s < ids.size(); s++) {
        rows[s * row_words] = r[ids[s] + 1];
        for (uint32_t k = 0; k < alen; k++) rows[s * row_words + 1 + k] = r[ids[s] + 2 + k];
    }
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.dev.rows), rows.size() * 4)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.dev.rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.dev.keys), ids.size() * 4)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.dev.keys, ids.data(), ids.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.dev.htab), kCwHash)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.dev.htab, htab.data(), kCwHash, hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.dev.n_slots = uint32_t(ids.size());
    out.dev.row_words = row_words;
    out.dev.hmul = hmul;
    out.ready = true;
    return hipSuccess;
}

CnfaHotTables::~CnfaHotTables() {
    if (dev.rows) (void)hipFree(dev.rows);
    if (dev.keys) (void)hipFree(dev.keys);
    if (dev.htab) (void)hipFree(dev.htab);
}

hipError_t launch_cnfa_count(const CnfaHotTables& h, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts, hipStream_t s) {
    if (!h.ready) return hipErrorInvalidValue;
    const uint32_t halo_tiles = (g.halo + kTile - 1) / kTile;
    const uint64_t blocks = (g.n_chunks + kCwBlock - 1) / kCwBlock;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    CnfaEng eng; eng.c = a.cnfa; eng.cls = a.cnfa.classes;
    const size_t smem = size_t(kCwWaves) * 64 * kRow + size_t(h.dev.n_slots) * h.dev.row_words * 4 + size_t(h.dev.n_slots) * 4 + kCwHash + 256;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_cnfa_count), 160 * 1024); e != hipSuccess) return e;
    k_cnfa_count<<<dim3(uint32_t(blocks)), dim3(kCwBlock), smem, s>>>(eng, h.dev, g, counts, halo_tiles);
    return hipGetLastError();
}

}  // namespace acgpu
