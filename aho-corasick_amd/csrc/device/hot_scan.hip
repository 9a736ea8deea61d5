// LDS-resident fast path of the Standard/unanchored full-DFA overlapping scan (gfx950).
//
// Same chunked walk as k_walk_count (tile_walk.hpp), but the per-byte transition
//   next = trans[sid + classes[byte]]                      (src/dfa.rs:218-226)
// is served from a 256-wide u16 table whose hottest rows (start state + the states at distance 1,
// i.e. the rows that serve ~90 % of the bytes of a random haystack) live in LDS;
// deeper rows fall back to the same table in global memory (L2 / Infinity-Cache resident).
// Match states are recognised by one compare (hid >= first_match) exactly like the reference's
// `sid <= max_special_id` trick (src/dfa.rs:229-241), just with the order reversed.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hot.hpp"
#include "tile_walk.hpp"

namespace acgpu {

namespace {

constexpr uint32_t kMaxHotRows = 192;  // 192 x 512 B = 96 KiB of LDS

struct HotStep {
    const uint16_t* lds;       // [n_hot][256]
    const uint16_t* tab;       // [n_states][256]
    const uint32_t* hid2sid;
    DfaEng eng;                // for match-list lengths
    uint32_t n_hot, first_match;
    uint32_t hid;
    uint32_t cnt;
    bool alive;
    __device__ __forceinline__ void step(uint8_t byte, bool owned) {
        const uint32_t idx = (hid << 8) | byte;
        hid = hid < n_hot ? lds[idx] : tab[idx];
        if (hid >= first_match && owned) cnt += eng.match_len(hid2sid[hid]);
    }
};

__global__ __launch_bounds__(kBlock) void k_hot_count(DfaEng eng, const uint16_t* __restrict__ tab,
                                                      const uint32_t* __restrict__ hid2sid, uint32_t n_hot,
                                                      uint32_t first_match, uint32_t start, ScanGeom g,
                                                      uint32_t* __restrict__ counts, uint32_t halo_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* s_tile = smem;                                              // kWaves * 64 * kRow bytes
    uint16_t* s_hot = reinterpret_cast<uint16_t*>(smem + kWaves * 64 * kRow);  // n_hot * 256 u16
    {
        const uint4* src = reinterpret_cast<const uint4*>(tab);
        uint4* dst = reinterpret_cast<uint4*>(s_hot);
        const uint32_t n16 = n_hot * 32;  // 512 B per row = 32 x 16 B
        for (uint32_t i = threadIdx.x; i < n16; i += kBlock) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_chunk0 = (uint64_t(blockIdx.x) * kWaves + wave) * 64;
    const uint64_t ci = wave_chunk0 + lane;
    const bool valid = ci < g.n_chunks;
    HotStep f{s_hot, tab, hid2sid, eng, n_hot, first_match, start, 0u, valid};
    if (valid && ci == 0 && g.emit_start_matches && start >= first_match) f.cnt += eng.match_len(hid2sid[start]);
    tile_walk(g, halo_tiles, s_tile + wave * 64 * kRow, wave_chunk0, lane, f);
    if (valid) counts[ci] = f.cnt;
}

}  // namespace

hipError_t build_hot_tables(const NNfa& n, const Dfa& d, HotTables& out) {
    out.ready = false;
    const size_t N = n.states();
    // hid order: DEAD, non-match states breadth first, match states breadth first.
    // n.bfs = [START_U, START_A, queue...]; START_A and FAIL are unreachable from an unanchored walk.
    std::vector<uint32_t> order;
    order.reserve(N);
    order.push_back(kDead);
    const uint32_t su = n.special.start_unanchored_id, sa = n.special.start_anchored_id;
    for (uint32_t s : n.bfs) if (s != sa && !n.is_match(s)) order.push_back(s);
    const uint32_t first_match = uint32_t(order.size());
    for (uint32_t s : n.bfs) if (s != sa && n.is_match(s)) order.push_back(s);
    const size_t nh = order.size();
    if (nh > 65535) return hipSuccess;  // u16 ids only; larger automata use the generic walk
    std::vector<uint32_t> sid2hid(N, 0);
    for (size_t h = 0; h < nh; h++) sid2hid[order[h]] = uint32_t(h);
    // hot rows: start state + non-match states at distance 1 (stored depth 0), capped by the LDS budget
    uint32_t n_hot = 1;  // DEAD row is row 0 (all zeros); it is never looked up, but keeps indices simple
    for (size_t h = 1; h < first_match; h++) {
        const uint32_t s = order[h];
        const bool shallow = (s == su) || n.depth[s] == 0;
        if (!shallow || n_hot >= kMaxHotRows) break;
        n_hot++;
    }
    std::vector<uint16_t> tab(nh * 256, 0);
    std::vector<uint32_t> hid2sid(nh, 0);
    for (size_t h = 1; h < nh; h++) {
        const uint32_t s = order[h];
        hid2sid[h] = s << d.stride2;
        const uint32_t* row = &d.trans[size_t(s) << d.stride2];
        for (int b = 0; b < 256; b++) tab[h * 256 + b] = uint16_t(sid2hid[row[d.byte_classes[b]] >> d.stride2]);
    }
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.tab), tab.size() * sizeof(uint16_t))) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.hid2sid), hid2sid.size() * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipMemcpy(out.tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.hid2sid, hid2sid.data(), hid2sid.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.n_states = uint32_t(nh);
    out.first_match = first_match;
    out.n_hot = n_hot;
    out.start = sid2hid[su];
    out.ready = true;
    return hipSuccess;
}

hipError_t launch_hot_count(const HotTables& h, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts,
                            hipStream_t s) {
    const uint32_t halo_tiles = (g.halo + kTile - 1) / kTile;
    const uint64_t blocks = (g.n_chunks + kBlock - 1) / kBlock;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    DfaEng eng; eng.d = a.dfa; eng.cls = a.dfa.classes;
    const size_t smem = size_t(kWaves) * 64 * kRow + size_t(h.n_hot) * 512;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_hot_count),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    k_hot_count<<<dim3(uint32_t(blocks)), dim3(kBlock), smem, s>>>(eng, h.tab, h.hid2sid, h.n_hot, h.first_match,
                                                                  h.start, g, counts, halo_tiles);
    return hipGetLastError();
}

}  // namespace acgpu
