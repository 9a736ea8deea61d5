// Calibration: what 24-byte match records cost to WRITE (gfx950).  Every record-writing kernel of the library -- the chunk
// fill, the event emit, the order pass -- moves records at 1.5-2.6 TB/s; is that the store pattern or the memory system?
// Patterns (2 GiB of output each, 256 persistent workgroups of 1 024 threads):
//   0  plain streaming store: lane i writes 16 bytes at 16 i (whole lines per instruction)
//   1  records, lanes consecutive: lane i writes record i as dwordx4 at 24 i + dwordx2 at 24 i + 16
//   2  records, per-lane runs: lane i writes R = 8 consecutive records of its own (k_lw_fill's pattern: a lane owns a sub-range)
//   3  records staged through LDS and written as whole 16-byte units (lane i writes 16 bytes at 16 i of the wave's 1 536-byte block)
//   4  records, one stream per lane: every lane of every wave appends to a region of its own (262 144 open lines: what 64
//      consecutive events of 64 different lane-chunks did to the event emit before its window sort)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void put(uint8_t* p, uint64_t rec) {
    uint32_t* q = reinterpret_cast<uint32_t*>(p);
    *reinterpret_cast<uint4*>(q) = make_uint4(uint32_t(rec), 0u, uint32_t(rec * 3), 0u);
    *reinterpret_cast<uint2*>(q + 4) = make_uint2(uint32_t(rec * 3 + 5), 0u);
}

template <int PAT>
__global__ __launch_bounds__(1024) void k_write(uint8_t* __restrict__ out, size_t n_bytes) {
    __shared__ __attribute__((aligned(16))) uint8_t s[16 * 1536];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = size_t(blockIdx.x) * 16 + wv, nwaves = size_t(gridDim.x) * 16;
    if (PAT == 0) {
        for (size_t b = wave * 1024; b + 1024 <= n_bytes; b += nwaves * 1024)
            *reinterpret_cast<uint4*>(out + b + 16 * lane) = make_uint4(uint32_t(b), 1u, 2u, 3u);
    } else if (PAT == 1) {
        for (size_t r = wave * 64; (r + 64) * 24 <= n_bytes; r += nwaves * 64) put(out + (r + lane) * 24, r + lane);
    } else if (PAT == 2) {
        constexpr int R = 8;
        for (size_t r = wave * 64 * R; (r + 64 * R) * 24 <= n_bytes; r += nwaves * 64 * R)
            for (int k = 0; k < R; k++) put(out + (r + size_t(lane) * R + k) * 24, r + lane * R + k);
    } else if (PAT == 3) {
        uint8_t* w = s + wv * 1536;
        for (size_t r = wave * 64; (r + 64) * 24 <= n_bytes; r += nwaves * 64) {
            put(w + lane * 24, r + lane);   // (LDS)
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint8_t* dst = out + r * 24;
            // 1 536 bytes = 96 units of 16 bytes: lanes 0..63 then 0..31
            *reinterpret_cast<uint4*>(dst + 16 * lane) = *reinterpret_cast<const uint4*>(w + 16 * lane);
            if (lane < 32) *reinterpret_cast<uint4*>(dst + 1024 + 16 * lane) = *reinterpret_cast<const uint4*>(w + 1024 + 16 * lane);
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        const size_t streams = nwaves * 64, per = n_bytes / streams / 24;   // records per stream
        uint8_t* mine = out + (wave * 64 + lane) * per * 24;
        for (size_t k = 0; k < per; k++) put(mine + k * 24, k);
    }
}

int main() {
    const size_t n = size_t(2) << 30;
    uint8_t* d;
    if (hipMalloc(&d, n) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[5] = {"streaming dwordx4", "records, lanes consecutive (x4 + x2)", "records, 8 per lane in a row", "records through LDS, whole 16-byte units",
                            "records, one stream per lane"};
    for (int pat = 0; pat < 5; pat++) {
        float best = 0;
        for (int it = 0; it < 4; it++) {
            (void)hipEventRecord(e0);
            switch (pat) {
                case 0: k_write<0><<<256, 1024>>>(d, n); break;
                case 1: k_write<1><<<256, 1024>>>(d, n); break;
                case 2: k_write<2><<<256, 1024>>>(d, n); break;
                case 3: k_write<3><<<256, 1024>>>(d, n); break;
                default: k_write<4><<<256, 1024>>>(d, n); break;
            }
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (it && (best == 0 || ms < best)) best = ms;
        }
        printf("pattern %d  %-46s %.3f ms  %.1f GB/s\n", pat, names[pat], best, n / best / 1e6);
    }
    return 0;
}
