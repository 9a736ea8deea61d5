// Merge of two ordered match-record streams (gfx950): every record finds its output rank by one binary search in the
// OTHER stream (its rank in its own stream is its index), one thread per record, records moved as 16 + 8 byte vectors.
// Used by the overlapping search of a split pattern set (capi_overlap.cpp: a dictionary of long patterns + a few short
// stragglers, each searched by the engine that suits it); the record streams it merges are small next to the haystack.
#include "merge.hpp"

namespace acgpu {

namespace {

// the reference's order inside one end position: longer patterns first (own patterns before the failure chain's), then id
__device__ __forceinline__ bool rec_less(const acgpu_match& x, const acgpu_match& y) {
    if (x.end != y.end) return x.end < y.end;
    const uint64_t lx = x.end - x.start, ly = y.end - y.start;
    if (lx != ly) return lx > ly;
    return x.pattern < y.pattern;
}

__global__ __launch_bounds__(256) void k_merge_records(const acgpu_match* __restrict__ a, const acgpu_match* __restrict__ b, uint64_t na,
                                                       uint64_t nb, acgpu_match* __restrict__ out, uint64_t* __restrict__ totals) {
    const uint64_t n = na + nb;
    if (totals && blockIdx.x == 0 && threadIdx.x == 0) totals[0] = n;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += uint64_t(gridDim.x) * blockDim.x) {
        const bool from_a = i < na;
        const acgpu_match x = from_a ? a[i] : b[i - na];
        const acgpu_match* other = from_a ? b : a;
        uint64_t lo = 0, hi = from_a ? nb : na;   // number of records of the other stream that come before x
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (rec_less(other[mid], x)) lo = mid + 1; else hi = mid;
        }
        const uint64_t pos = (from_a ? i : i - na) + lo;
        uint32_t* p = reinterpret_cast<uint32_t*>(out + pos);
        *reinterpret_cast<uint4*>(p) = make_uint4(x.pattern, 0u, uint32_t(x.start), uint32_t(x.start >> 32));
        *reinterpret_cast<uint2*>(p + 4) = make_uint2(uint32_t(x.end), uint32_t(x.end >> 32));
    }
}

}  // namespace

hipError_t launch_merge_records(const acgpu_match* a, const acgpu_match* b, uint64_t na, uint64_t nb, acgpu_match* out,
                                uint64_t* totals, hipStream_t s) {
    const uint64_t n = na + nb;
    uint64_t blocks = (n + 255) / 256;
    if (blocks == 0) blocks = 1;           // (still writes totals)
    if (blocks > 65536) blocks = 65536;    // grid-stride beyond
    k_merge_records<<<dim3(uint32_t(blocks)), dim3(256), 0, s>>>(a, b, na, nb, out, totals);
    return hipGetLastError();
}

}  // namespace acgpu
