// replace_all on the device: given the non-overlapping matches of find_iter (ordered, resident in
// HBM), build the output of  Automaton::try_replace_all_with_bytes  (src/automaton.rs:530-550):
//
//     last = 0;  for m in find_iter: dst += hay[last..m.start]; last = m.end; dst += replace_with[m.pattern]
//     dst += hay[last..]
//
// as a segmented copy.  Segment i (one per match) = the literal gap before match i followed by its replacement;
// a final segment holds the tail.  k_repl_measure computes the segment lengths, an exclusive u64 scan turns them
// into output offsets, and k_repl_copy lets every thread produce 16 consecutive output bytes (one 16-byte store):
// a per-workgroup binary search finds the first segment of the tile, threads advance linearly from there
// (matches are sparse), and a 16-byte run that lies inside one literal gap is copied with two aligned 16-byte loads
// and a funnel shift.  The &str variants (src/automaton.rs:493-522) skip matches that split a UTF-8 code point:
// such a match is treated as "replaced by its own text", which produces the same bytes.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace acgpu {

namespace {

constexpr int kReplBlock = 256;
constexpr uint32_t kSkipBit = 1u << 31;  // in ReplSeg::rlen: replacement = the matched text itself

__device__ __forceinline__ bool is_char_boundary(const uint8_t* hay, uint64_t len, uint64_t i) {
    return i == 0 || i >= len || (hay[i] & 0xC0u) != 0x80u;   // str::is_char_boundary
}

// per match: gap (literal bytes since the previous match), replacement length, and the segment's output length
__global__ __launch_bounds__(kReplBlock) void k_repl_measure(const acgpu_match* __restrict__ M, uint64_t m,
                                                             const uint8_t* __restrict__ hay, uint64_t hay_len,
                                                             const uint64_t* __restrict__ roff, uint32_t utf8,
                                                             uint64_t* __restrict__ seglen, uint32_t* __restrict__ rlen) {
    const uint64_t i = uint64_t(blockIdx.x) * kReplBlock + threadIdx.x;
    if (i > m) return;
    const uint64_t prev_end = i ? M[i - 1].end : 0;
    if (i == m) { seglen[i] = hay_len - prev_end; return; }  // the tail segment
    const acgpu_match x = M[i];
    uint32_t r = uint32_t(roff[x.pattern + 1] - roff[x.pattern]);
    if (utf8 && !(is_char_boundary(hay, hay_len, x.start) && is_char_boundary(hay, hay_len, x.end)))
        r = uint32_t(x.end - x.start) | kSkipBit;
    rlen[i] = r;
    seglen[i] = (x.start - prev_end) + (r & ~kSkipBit);
}

// ---- exclusive u64 scan, three phases, 1024 items per workgroup
constexpr int kScanB = 256, kScanPer = 4, kScanTile = kScanB * kScanPer;

__device__ __forceinline__ uint64_t block_exclusive(uint64_t v, uint64_t* s_w, uint64_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint64_t base = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < kScanB / 64; k++) {
        if (k < wave) base += s_w[k];
        total += s_w[k];
    }
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(kScanB) void k_u64_block_sums(const uint64_t* __restrict__ v, uint64_t n,
                                                           uint64_t* __restrict__ bsum) {
    __shared__ uint64_t s_w[kScanB / 64];
    const uint64_t i0 = (uint64_t(blockIdx.x) * kScanB + threadIdx.x) * kScanPer;
    uint64_t t = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; j++) if (i0 + j < n) t += v[i0 + j];
    uint64_t total;
    (void)block_exclusive(t, s_w, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the block sums in place; total -> *total_out
__global__ __launch_bounds__(kScanB) void k_u64_scan_tops(uint64_t* __restrict__ bsum, uint64_t nb,
                                                          uint64_t* __restrict__ total_out) {
    __shared__ uint64_t s_w[kScanB / 64];
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < nb; b0 += kScanB) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < nb ? bsum[i] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive(v, s_w, total);
        if (i < nb) bsum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(kScanB) void k_u64_scan_write(uint64_t* __restrict__ v, uint64_t n,
                                                           const uint64_t* __restrict__ bsum) {
    __shared__ uint64_t s_w[kScanB / 64];
    const uint64_t i0 = (uint64_t(blockIdx.x) * kScanB + threadIdx.x) * kScanPer;
    uint64_t x[kScanPer], t = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; j++) { x[j] = i0 + j < n ? v[i0 + j] : 0; t += x[j]; }
    uint64_t total;
    uint64_t run = bsum[blockIdx.x] + block_exclusive(t, s_w, total);
#pragma unroll
    for (int j = 0; j < kScanPer; j++) {
        if (i0 + j < n) v[i0 + j] = run;
        run += x[j];
    }
}

// ---- the copy
struct ReplArgs {
    const acgpu_match* M;
    uint64_t m;
    const uint8_t* hay;      // haystack base (byte pointer, any alignment)
    uint64_t hay_len;
    const uint8_t* rbytes;   // replacement strings, concatenated
    const uint64_t* roff;    // [n_patterns + 1]
    const uint64_t* segoff;  // [m + 1] output offset of each segment (exclusive scan of the lengths)
    const uint32_t* rlen;    // [m]
    const uint64_t* total;   // output length
    uint8_t* out;            // 16-byte aligned
    uint64_t cap;
};

// 16 haystack bytes starting at byte address `src` (any alignment), without reading outside [hay16, end16)
__device__ __forceinline__ uint4 load16_unaligned(const uint8_t* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint4* q = reinterpret_cast<const uint4*>(a & ~uintptr_t(15));
    const uint32_t sh = uint32_t(a & 15);
    const uint4 lo = q[0];
    if (sh == 0) return lo;
    const uint4 hi = q[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const uint32_t d = sh >> 2, b = (sh & 3) * 8;
    uint32_t r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // dwords d+k and d+k+1, funnel-shifted by b bits (d is 0..3, selected without dynamic register indexing)
        uint32_t x0 = w[k], x1 = w[k + 1];
        if (d == 1) { x0 = w[k + 1]; x1 = w[k + 2]; }
        if (d == 2) { x0 = w[k + 2]; x1 = w[k + 3]; }
        if (d == 3) { x0 = w[k + 3]; x1 = k + 4 < 8 ? w[k + 4] : 0u; }
        r[k] = __builtin_amdgcn_alignbit(x1, x0, b);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

__global__ __launch_bounds__(kReplBlock) void k_repl_copy(ReplArgs a) {
    __shared__ uint64_t s_first;
    const uint64_t total = *a.total;
    if (total > a.cap) return;
    const uint64_t tile0 = uint64_t(blockIdx.x) * (kReplBlock * 16);
    if (tile0 >= total) return;
    if (threadIdx.x == 0) {  // last segment whose offset is <= tile0 (segoff[0] == 0)
        uint64_t lo = 0, hi = a.m;  // answer in [lo, hi]
        while (lo < hi) {
            const uint64_t mid = (lo + hi + 1) >> 1;
            if (a.segoff[mid] <= tile0) lo = mid; else hi = mid - 1;
        }
        s_first = lo;
    }
    __syncthreads();
    const uint64_t y0 = tile0 + uint64_t(threadIdx.x) * 16;
    if (y0 >= total) return;
    uint64_t i = s_first;
    while (i < a.m && a.segoff[i + 1] <= y0) i++;
    // segment i: [segoff[i], +gap) literal from hay[prev_end..], then the replacement
    uint64_t seg = a.segoff[i];
    uint64_t prev_end = i ? a.M[i - 1].end : 0;
    uint64_t gap = (i < a.m ? a.M[i].start : a.hay_len) - prev_end;
    const uint64_t n = total - y0 < 16 ? total - y0 : 16;
    const uintptr_t hay16 = reinterpret_cast<uintptr_t>(a.hay) & ~uintptr_t(15);
    const uintptr_t end16 = (reinterpret_cast<uintptr_t>(a.hay) + a.hay_len + 15) & ~uintptr_t(15);
    if (n == 16 && y0 - seg + 16 <= gap) {   // fast path: 16 bytes of one literal gap
        const uint8_t* src = a.hay + prev_end + (y0 - seg);
        const uintptr_t s0 = reinterpret_cast<uintptr_t>(src) & ~uintptr_t(15);
        if (s0 >= hay16 && s0 + 32 <= end16) {
            *reinterpret_cast<uint4*>(a.out + y0) = load16_unaligned(src);
            return;
        }
    }
    uint8_t buf[16];
    for (uint64_t k = 0; k < n; k++) {
        const uint64_t y = y0 + k;
        while (i < a.m && a.segoff[i + 1] <= y) {  // next segment (zero-length segments are skipped)
            i++;
            seg = a.segoff[i];
            prev_end = a.M[i - 1].end;
            gap = (i < a.m ? a.M[i].start : a.hay_len) - prev_end;
        }
        const uint64_t r = y - seg;
        uint8_t c;
        if (r < gap) {
            c = a.hay[prev_end + r];
        } else {
            const acgpu_match x = a.M[i];
            const uint32_t rl = a.rlen[i];
            c = (rl & kSkipBit) ? a.hay[x.start + (r - gap)] : a.rbytes[a.roff[x.pattern] + (r - gap)];
        }
        buf[k] = c;
    }
    if (n == 16) {
        uint4 v;
        __builtin_memcpy(&v, buf, 16);
        *reinterpret_cast<uint4*>(a.out + y0) = v;
    } else {
        for (uint64_t k = 0; k < n; k++) a.out[y0 + k] = buf[k];
    }
}

}  // namespace

size_t replace_scratch_bytes(uint64_t m) {
    const uint64_t nseg = m + 1;
    const uint64_t nb = (nseg + kScanTile - 1) / kScanTile;
    return size_t(nseg * 8 + (nb + 1) * 8 + m * 4 + 64);
}

// Phase 1: segment lengths + exclusive scan.  work: replace_scratch_bytes(m) bytes; *total_out (device u64) receives
// the output length.
hipError_t launch_replace_measure(const acgpu_match* M, uint64_t m, const uint8_t* hay, uint64_t hay_len,
                                  const uint64_t* roff, bool utf8, void* work, uint64_t* total_out, hipStream_t s) {
    const uint64_t nseg = m + 1;
    const uint64_t nb = (nseg + kScanTile - 1) / kScanTile;
    uint64_t* segoff = static_cast<uint64_t*>(work);
    uint64_t* bsum = segoff + nseg;
    uint32_t* rlen = reinterpret_cast<uint32_t*>(bsum + nb + 1);
    k_repl_measure<<<dim3(uint32_t((nseg + kReplBlock - 1) / kReplBlock)), dim3(kReplBlock), 0, s>>>(
        M, m, hay, hay_len, roff, utf8 ? 1u : 0u, segoff, rlen);
    k_u64_block_sums<<<dim3(uint32_t(nb)), dim3(kScanB), 0, s>>>(segoff, nseg, bsum);
    k_u64_scan_tops<<<dim3(1), dim3(kScanB), 0, s>>>(bsum, nb, total_out);
    k_u64_scan_write<<<dim3(uint32_t(nb)), dim3(kScanB), 0, s>>>(segoff, nseg, bsum);
    return hipGetLastError();
}

// Phase 2: the copy; out_len = the value phase 1 left in *total_out (read back by the host to size `out`).
// `out` must be 16-byte aligned and hold out_len bytes.
hipError_t launch_replace_copy(const acgpu_match* M, uint64_t m, const uint8_t* hay, uint64_t hay_len,
                               const uint8_t* rbytes, const uint64_t* roff, const void* work, const uint64_t* total_out,
                               uint8_t* out, uint64_t out_len, hipStream_t s) {
    if (out_len == 0) return hipSuccess;
    const uint64_t nseg = m + 1;
    const uint64_t nb = (nseg + kScanTile - 1) / kScanTile;
    const uint64_t* segoff = static_cast<const uint64_t*>(work);
    const uint32_t* rlen = reinterpret_cast<const uint32_t*>(segoff + nseg + nb + 1);
    const uint64_t blocks = (out_len + kReplBlock * 16 - 1) / (kReplBlock * 16);
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    ReplArgs a{M, m, hay, hay_len, rbytes, roff, segoff, rlen, total_out, out, out_len};
    k_repl_copy<<<dim3(uint32_t(blocks)), dim3(kReplBlock), 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace acgpu
