// LDS-staged chunk walk shared by the count kernels (generic walk and hot-row fast path).
//
// One wavefront covers 64 consecutive lane-chunks.  Per step every lane consumes one 64-byte tile
// of ITS chunk.  The tiles are fetched cooperatively: load instruction q brings in the tiles of
// chunks 16q..16q+15, four lanes x 16 B per chunk, i.e. HBM is read in contiguous 64-byte segments
// and each byte exactly once (plus the warm-up halo).  The tile goes through a wave-private LDS
// region (row stride 80 B: the 64 ds_read_b128 of a wave are bank-conflict free) and comes back as
// 16 bytes per lane per ds_read.  The next step's global loads are in flight while the current
// tile is processed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "engines.hpp"

namespace acgpu {

constexpr int kTile = 64;          // haystack bytes per lane per step
constexpr int kRow = kTile + 16;   // padded LDS row (80 B: 5 x 16 B, 5 coprime with 16 slots)
constexpr int kWaves = 4;
constexpr int kBlock = kWaves * 64;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int clamp_tile(int64_t x) { return x < 0 ? 0 : (x > kTile ? kTile : int(x)); }

// F provides:  bool alive;  void step(uint8_t byte, bool owned);
template <class F>
__device__ __forceinline__ void tile_walk(const ScanGeom& g, uint32_t halo_tiles, uint8_t* tile,
                                          uint64_t wave_chunk0, int lane, F& f) {
    const uint64_t ci = wave_chunk0 + lane;
    ChunkRange r = {0, 0, 0};
    if (ci < g.n_chunks) r = chunk_range(g, ci);
    const int64_t my_tile0 = int64_t(g.grid0 + ci * uint64_t(g.chunk)) - int64_t(halo_tiles) * kTile;

    // loader roles: in load instruction q this lane fetches 16-byte part (lane&3) of chunk 16q+(lane>>2)
    int64_t role_pv[4];
    int64_t role_w[4], role_hi[4];
    uint32_t role_off[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int c = 16 * q + (lane >> 2), part = lane & 3;
        const uint64_t cq = wave_chunk0 + c;
        ChunkRange rq = {0, 0, 0};
        if (cq < g.n_chunks) rq = chunk_range(g, cq);
        role_w[q] = int64_t(rq.w);
        role_hi[q] = int64_t(rq.hi);
        role_pv[q] = int64_t(g.grid0 + cq * uint64_t(g.chunk)) - int64_t(halo_tiles) * kTile + part * 16;
        role_off[q] = uint32_t(c * kRow + part * 16);
    }
    const int nsteps = int(halo_tiles) + int(g.chunk / kTile);

    auto fetch = [&](int q, int j) -> uint4 {
        const int64_t pv = role_pv[q] + int64_t(j) * kTile;
        uint4 v = make_uint4(0, 0, 0, 0);
        // only 16-byte parts holding at least one live byte are loaded: the kernel never touches
        // memory outside the 16-byte-aligned hull of [walk start, chunk end)
        if (pv + 16 > role_w[q] && pv < role_hi[q]) { ACGPU_HAY_CHECK(g, pv, 16); v = *reinterpret_cast<const uint4*>(g.hay16 + pv); }
        return v;
    };

    uint4 pre[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pre[q] = fetch(q, 0);

    for (int j = 0; j < nsteps; j++) {
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<uint4*>(tile + role_off[q]) = pre[q];
        wave_lds_fence();
        if (j + 1 < nsteps) {
#pragma unroll
            for (int q = 0; q < 4; q++) pre[q] = fetch(q, j + 1);
        }
        const int64_t tv = my_tile0 + int64_t(j) * kTile;
        const int kb = clamp_tile(int64_t(r.w) - tv);    // first live byte of this tile
        const int ke = clamp_tile(int64_t(r.hi) - tv);   // one past the last live byte
        const int kc = clamp_tile(int64_t(r.lo) - tv);   // first byte whose matches this chunk owns
        for (int q = 0; q < 4; q++) {
            const uint4 d = *reinterpret_cast<const uint4*>(tile + lane * kRow + q * 16);
            const uint32_t wds[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const int k = q * 16 + b;
                if (k >= kb && k < ke && f.alive) f.step(uint8_t(wds[b >> 2] >> (8 * (b & 3))), k >= kc);
            }
        }
    }
}

}  // namespace acgpu
