// Does gfx950 under this ROCm serve unaligned ds_read_b64 / ds_read_b32 (byte-granular addresses)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint64_t* out64, uint32_t* out32) {
    __shared__ __attribute__((aligned(16))) uint8_t s[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s[i] = uint8_t(i);
    __syncthreads();
    const uint32_t addr = uint32_t(reinterpret_cast<uintptr_t>(s)) + threadIdx.x;   // byte address, lane i -> offset i
    uint64_t v64; uint32_t v32;
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(addr));
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v32) : "v"(addr));
    out64[threadIdx.x] = v64; out32[threadIdx.x] = v32;
}
int main() {
    uint64_t* d64; uint32_t* d32;
    hipMalloc(&d64, 64 * 8); hipMalloc(&d32, 64 * 4);
    k<<<1, 64>>>(d64, d32);
    uint64_t h64[64]; uint32_t h32[64];
    hipMemcpy(h64, d64, sizeof h64, hipMemcpyDeviceToHost); hipMemcpy(h32, d32, sizeof h32, hipMemcpyDeviceToHost);
    int bad64 = 0, bad32 = 0;
    for (int i = 0; i < 64; i++) {
        uint64_t w = 0; for (int b = 7; b >= 0; b--) w = (w << 8) | uint8_t(i + b);
        uint32_t x = 0; for (int b = 3; b >= 0; b--) x = (x << 8) | uint8_t(i + b);
        bad64 += h64[i] != w; bad32 += h32[i] != x;
    }
    printf("unaligned ds_read_b64: %s (%d bad)   unaligned ds_read_b32: %s (%d bad)   e.g. lane 3: %016llx %08x\n",
           bad64 ? "NO" : "yes", bad64, bad32 ? "NO" : "yes", bad32, (unsigned long long)h64[3], h32[3]);
    return 0;
}
