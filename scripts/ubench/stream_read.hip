// Calibration: HBM streaming-read rate and FETCH_SIZE/TCC counter semantics for the access patterns of the
// scan kernels (gfx950).  Pattern 0: wave-rows of 1024 B (aligned); pattern 1: wave-rows at a 1008 B stride
// (each lane 16 B, rows overlap by 16 B), both with 2 rows in flight per wave like k_pf_count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int STRIDE>
__global__ __launch_bounds__(1024) void k_stream(const uint8_t* __restrict__ p, size_t n, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = size_t(blockIdx.x) * 16 + (threadIdx.x >> 6);
    const size_t nwaves = size_t(gridDim.x) * 16;
    const size_t task = size_t(16) * STRIDE;
    unsigned acc = 0;
    for (size_t t = wave; (t + 1) * task + 1024 <= n; t += nwaves) {
        const uint8_t* base = p + t * task + size_t(lane) * 16;
        uint4 a = *reinterpret_cast<const uint4*>(base), b = *reinterpret_cast<const uint4*>(base + STRIDE);
#pragma unroll 1
        for (int r = 0; r < 16; r += 2) {
            uint4 x = a, y = b;
            if (r + 2 < 16) {
                a = *reinterpret_cast<const uint4*>(base + size_t(r + 2) * STRIDE);
                b = *reinterpret_cast<const uint4*>(base + size_t(r + 3) * STRIDE);
            }
            acc += x.x ^ x.y ^ x.z ^ x.w ^ y.x ^ y.y ^ y.z ^ y.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t n = size_t(8) << 30;
    uint8_t* d; unsigned* o;
    if (hipMalloc(&d, n) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 1, n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int pat = 0; pat < 2; pat++) {
        for (int it = 0; it < 3; it++) {
            (void)hipEventRecord(e0);
            if (pat == 0) k_stream<1024><<<256, 1024>>>(d, n, o);
            else k_stream<1008><<<256, 1024>>>(d, n, o);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (it == 2) printf("pattern %d (row stride %d): %.3f ms  %.1f GB/s\n", pat, pat ? 1008 : 1024, ms, n / ms / 1e6);
        }
    }
    return 0;
}
