// What is the best streaming-read rate of this GPU?  Sweeps occupancy (workgroups x threads), 16-byte loads in
// flight per lane and temporal / non-temporal loads over an 8 GiB buffer.  (k_pf_count is LDS-limited to one
// 1024-thread workgroup per CU with two row pairs in flight; this shows what that costs.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int INFLIGHT, bool NT>
__global__ void k_stream(const uint4* __restrict__ p, size_t n16, unsigned* out) {
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t nthreads = size_t(gridDim.x) * blockDim.x;
    unsigned acc = 0;
    for (size_t i = tid; i + size_t(INFLIGHT - 1) * nthreads < n16; i += nthreads * INFLIGHT) {
        uint4 v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) {
            const uint4* q = p + i + size_t(k) * nthreads;
            if (NT) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(q));
                v[k] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[k] = *q;
            }
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int INFLIGHT, bool NT>
static void run(const uint4* d, size_t n16, unsigned* o, int blocks, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        (void)hipEventRecord(e0);
        k_stream<INFLIGHT, NT><<<blocks, threads>>>(d, n16, o);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    printf("blocks %5d x %4d threads, %d x 16 B in flight, %s: %.3f ms  %.0f GB/s\n", blocks, threads, INFLIGHT,
           NT ? "nontemporal" : "temporal   ", best, n16 * 16 / best / 1e6);
}

int main() {
    const size_t n = size_t(8) << 30;
    uint4* d; unsigned* o;
    if (hipMalloc(&d, n) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 1, n);
    const size_t n16 = n / 16;
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        run<2, false>(d, n16, o, blocks, 1024);
        run<4, false>(d, n16, o, blocks, 1024);
        run<8, false>(d, n16, o, blocks, 1024);
        run<4, true>(d, n16, o, blocks, 1024);
    }
    run<4, false>(d, n16, o, 8192, 256);
    run<8, false>(d, n16, o, 16384, 256);
    return 0;
}
