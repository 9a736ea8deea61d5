// Generic HIP kernels of the acgpu engine (gfx950).
//
//   k_walk_count  one haystack chunk per wavefront lane, byte-at-a-time transition walk
//                 (src/automaton.rs:1491-1534 loop body, any engine), haystack staged through
//                 LDS tiles so that HBM is read with contiguous 64-byte segments; produces one
//                 match count per chunk.
//   k_scan_*      exclusive scan of the per-chunk counts + order-preserving compaction of the
//                 non-empty chunks (__ballot / __popcll / __shfl).
//   k_walk_fill   re-walks only the non-empty chunks and writes the ordered match records.
//   k_find_*_serial  one-lane restatement of FindIter / try_find for the API paths that are
//                 inherently sequential in the reference (src/automaton.rs:857-936, :1259-1420).
//   k_gen_haystack   synthetic haystack generator.
//
// Chunk ownership rule: chunk g owns the matches whose last byte `at` lies in
// [lo_g, hi_g); the lane starts cold (start state) at max(span_start, lo_g - (L-1)) and only
// counts/emits from lo_g on.  Because a Standard Aho-Corasick state is the longest suffix of the
// text that is a pattern prefix (<= L bytes), the warmed-up state equals the true state for every
// owned position, so the concatenation over chunks is exactly the sequential stream.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.hpp"
#include "select.hpp"
#include "tile_walk.hpp"

namespace acgpu {

namespace {

template <class E>
struct CountStep {
    E eng;
    uint32_t sid;
    uint32_t cnt;
    bool alive;
    __device__ __forceinline__ void step(uint8_t byte, bool owned) {
        sid = eng.next(false, sid, byte);
        if (eng.is_special(sid)) {
            if (sid == kDevDead) alive = false;
            else if (owned && eng.is_match(sid)) cnt += eng.match_len(sid);
        }
    }
};

template <class E>
__global__ __launch_bounds__(kBlock) void k_walk_count(E eng, ScanGeom g, uint32_t* __restrict__ counts,
                                                       uint32_t halo_tiles) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[kWaves][64 * kRow];
    __shared__ uint8_t s_cls[256];
    s_cls[threadIdx.x] = eng.cls[threadIdx.x];
    __syncthreads();
    eng.cls = s_cls;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_chunk0 = (uint64_t(blockIdx.x) * kWaves + wave) * 64;
    const uint64_t ci = wave_chunk0 + lane;
    const bool valid = ci < g.n_chunks;
    CountStep<E> f{eng, eng.start(false), 0u, valid};
    if (valid && ci == 0 && g.emit_start_matches && eng.is_match(f.sid)) f.cnt += eng.match_len(f.sid);
    tile_walk(g, halo_tiles, s_tile[wave], wave_chunk0, lane, f);
    if (valid) counts[ci] = f.cnt;
}

// ------------------------------------------------------------------------- scan + compaction
// Each workgroup of 256 threads covers kScanItems * 256 = 1024 consecutive chunks (one uint4 of counts per thread).
constexpr int kScanItems = 4;
constexpr int kScanSpan = kScanItems * 256;

__device__ __forceinline__ uint4 load_counts4(const uint32_t* __restrict__ counts, uint64_t n, uint64_t i0) {
    if (i0 + 4 <= n) return *reinterpret_cast<const uint4*>(counts + i0);  // counts is 16-byte aligned, i0 % 4 == 0
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i0 < n) v.x = counts[i0];
    if (i0 + 1 < n) v.y = counts[i0 + 1];
    if (i0 + 2 < n) v.z = counts[i0 + 2];
    return v;
}

// the same four counts when they are the low words of 64-bit entries (ScanScratch::packed)
__device__ __forceinline__ uint4 load_counts4_packed(const uint64_t* __restrict__ packed, uint64_t n, uint64_t i0) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i0 + 4 <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(packed + i0), b = *reinterpret_cast<const uint4*>(packed + i0 + 2);
        return make_uint4(a.x, a.z, b.x, b.z);
    }
    if (i0 < n) v.x = uint32_t(packed[i0]);
    if (i0 + 1 < n) v.y = uint32_t(packed[i0 + 1]);
    if (i0 + 2 < n) v.z = uint32_t(packed[i0 + 2]);
    return v;
}

// level 1: per-workgroup totals
template <bool PACKED>
__global__ __launch_bounds__(256) void k_scan_block_sums(const void* __restrict__ counts, uint64_t n,
                                                         uint64_t* __restrict__ bsum, uint32_t* __restrict__ bact) {
    __shared__ uint64_t s_sum[4];
    __shared__ uint32_t s_act[4];
    const uint64_t i0 = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * kScanItems;
    const uint4 c = PACKED ? load_counts4_packed(static_cast<const uint64_t*>(counts), n, i0)
                           : load_counts4(static_cast<const uint32_t*>(counts), n, i0);
    uint64_t v = uint64_t(c.x) + c.y + c.z + c.w;
    uint32_t act = (c.x != 0) + (c.y != 0) + (c.z != 0) + (c.w != 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); act += __shfl_down(act, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = v; s_act[threadIdx.x >> 6] = act; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        bact[blockIdx.x] = s_act[0] + s_act[1] + s_act[2] + s_act[3];
    }
}

// level 2: exclusive scan of the block totals by one workgroup (thread-serial slices + wave shuffle scans).
// Slices of up to kTopsRegs entries are held in registers, so their loads are independent (one memory round trip)
// instead of a chain of dependent read-modify-writes.
constexpr int kTopsRegs = 16;
__global__ __launch_bounds__(256) void k_scan_tops(uint64_t* __restrict__ bsum, uint32_t* __restrict__ bact,
                                                   uint64_t nblocks, uint64_t* __restrict__ totals,
                                                   uint64_t* __restrict__ host_totals, const uint32_t* __restrict__ extra32) {
    __shared__ uint64_t s_sum[4];
    __shared__ uint64_t s_act[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t per = (nblocks + 255) / 256;
    const uint64_t b0 = uint64_t(threadIdx.x) * per;
    const uint64_t b1 = b0 + per < nblocks ? b0 + per : nblocks;
    const bool in_regs = per <= uint64_t(kTopsRegs);
    uint64_t xs[kTopsRegs];
    uint32_t xa[kTopsRegs];
    uint64_t ls = 0, la = 0;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kTopsRegs; j++) {
            const bool ok = b0 + j < b1;
            xs[j] = ok ? bsum[b0 + j] : 0;
            xa[j] = ok ? bact[b0 + j] : 0;
        }
#pragma unroll
        for (int j = 0; j < kTopsRegs; j++) { ls += xs[j]; la += xa[j]; }
    } else {
        for (uint64_t b = b0; b < b1; b++) { ls += bsum[b]; la += bact[b]; }
    }
    uint64_t is = ls, ia = la;  // inclusive scans across the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t ts = __shfl_up(is, o, 64), ta = __shfl_up(ia, o, 64);
        if (lane >= o) { is += ts; ia += ta; }
    }
    if (lane == 63) { s_sum[wave] = is; s_act[wave] = ia; }
    __syncthreads();
    uint64_t rs = is - ls, ra = ia - la;
    for (int k = 0; k < wave; k++) { rs += s_sum[k]; ra += s_act[k]; }
    if (threadIdx.x == 255) {
        totals[0] = rs + ls; totals[1] = ra + la;
        if (host_totals) { host_totals[0] = rs + ls; host_totals[1] = ra + la; if (extra32) host_totals[2] = *extra32; }
    }
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kTopsRegs; j++) {
            if (b0 + j < b1) { bsum[b0 + j] = rs; bact[b0 + j] = uint32_t(ra); }
            rs += xs[j]; ra += xa[j];
        }
    } else {
        for (uint64_t b = b0; b < b1; b++) {
            const uint64_t x = bsum[b]; const uint32_t y = bact[b];
            bsum[b] = rs; bact[b] = uint32_t(ra);
            rs += x; ra += y;
        }
    }
}

// level 3: ordered list of the non-empty chunks with their exclusive output offsets (`active`, `aoff`: what the fill
// needs -- a few thousand entries instead of one u64 per chunk); DENSE additionally writes the offset of every chunk
// (the selection kernels index by block).
template <bool DENSE, bool PACKED = false>
__global__ __launch_bounds__(256) void k_scan_write(const void* __restrict__ counts, uint64_t n,
                                                    const uint64_t* __restrict__ bsum,
                                                    const uint32_t* __restrict__ bact,
                                                    uint64_t* __restrict__ offsets, uint64_t* __restrict__ active,
                                                    uint64_t* __restrict__ aoff) {
    __shared__ uint64_t s_sum[4];
    __shared__ uint32_t s_act[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t i0 = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * kScanItems;
    const uint4 c4 = PACKED ? load_counts4_packed(static_cast<const uint64_t*>(counts), n, i0)
                            : load_counts4(static_cast<const uint32_t*>(counts), n, i0);
    const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
    const uint64_t tsum = uint64_t(c[0]) + c[1] + c[2] + c[3];
    const uint32_t tact = (c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0);
    // inclusive wave scans of the per-thread sums / active counts
    uint64_t v = tsum;
    uint32_t av = tact;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(v, o, 64);
        const uint32_t ta = __shfl_up(av, o, 64);
        if (lane >= o) { v += t; av += ta; }
    }
    if (lane == 63) { s_sum[wave] = v; s_act[wave] = av; }
    __syncthreads();
    uint64_t obase = bsum[blockIdx.x] + (v - tsum);
    uint64_t abase = uint64_t(bact[blockIdx.x]) + (av - tact);
    for (int k = 0; k < wave; k++) { obase += s_sum[k]; abase += s_act[k]; }
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
        if (i0 + j < n) {
            if (DENSE) offsets[i0 + j] = obase;
            if (c[j] != 0) { active[abase] = i0 + j; aoff[abase] = obase; abase++; }
            obase += c[j];
        }
    }
}

// ------------------------------------------------------------------------------------- fill
// One wavefront per non-empty chunk.  The chunk (plus its warm-up halo) is staged in LDS with coalesced
// 16-byte loads; lane l then walks sub-range l of the chunk (own warm-up halo, same ownership rule as
// the chunks themselves) ONCE, counting its matches and remembering its first match events in registers.
// The per-lane counts are prefix-summed across the wave (__shfl_up) and the records written at their
// final, ordered slots from the remembered events; a lane with more events than it could remember (rare)
// re-walks its sub-range.
constexpr uint32_t kFillStage = 4096 + 1024;  // staged bytes per chunk incl. halo; longer spans read global memory
constexpr int kFillEvents = 2;

template <class E>
struct FillWalk {
    const E& eng;
    const ScanGeom& g;
    __device__ __forceinline__ uint32_t write(uint32_t s, uint64_t end, acgpu_match* out) const {
        const uint32_t n = eng.match_len(s);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t pid = eng.match_pattern(s, i);
            acgpu_match m; m.pattern = pid; m._pad = 0; m.end = end; m.start = end - eng.pattern_len(pid);
            out[i] = m;
        }
        return n;
    }
    // WRITE = false: count + remember events;  WRITE = true: write every record (re-walk)
    template <bool WRITE, class B>
    __device__ __forceinline__ uint32_t run(uint64_t w, uint64_t lo, uint64_t hi, bool start_matches, B byte_at,
                                            acgpu_match* out, uint64_t (&ev_end)[kFillEvents],
                                            uint32_t (&ev_sid)[kFillEvents], uint32_t& nev) const {
        uint32_t n_out = 0;
        uint32_t sid = eng.start(false);
        auto emit = [&](uint32_t s, uint64_t end) {
            if (WRITE) {
                n_out += write(s, end, out + n_out);
            } else {
#pragma unroll
                for (int k = 0; k < kFillEvents; k++)
                    if (nev == uint32_t(k)) { ev_end[k] = end; ev_sid[k] = s; }
                nev++;
                n_out += eng.match_len(s);
            }
        };
        if (start_matches && eng.is_match(sid)) emit(sid, g.cold_floor - g.base_mis);  // at span_start
        for (uint64_t v = w; v < hi; v++) {
            sid = eng.next(false, sid, byte_at(v));
            if (eng.is_special(sid)) {
                if (sid == kDevDead) break;
                if (v >= lo && eng.is_match(sid)) emit(sid, v + 1 - g.base_mis);
            }
        }
        return n_out;
    }
};

// `totals` = {total matches, number of non-empty chunks} as written by k_scan_tops: the kernel reads them on the
// device, so the host can launch it without first synchronising on the scan (grid-stride over the chunks).
// Nothing is written when the total exceeds `cap` (the host then reports ACGPU_ERR_BUFFER_TOO_SMALL).
template <class E>
__global__ __launch_bounds__(64) void k_walk_fill(E eng, ScanGeom g, const uint64_t* __restrict__ active,
                                                  const uint64_t* __restrict__ totals, uint64_t cap,
                                                  const uint64_t* __restrict__ aoff,
                                                  acgpu_match* __restrict__ out, const unsigned long long* __restrict__ gate) {
    __shared__ uint8_t s_cls[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_hay[kFillStage];
    const int lane = threadIdx.x;
    const uint64_t a0 = blockIdx.x;
    const uint64_t n_active = totals[1];
    if (gate && *gate == 0) return;   // (the event form of the count pass already delivered the records)
    if (totals[0] > cap || a0 >= n_active) return;
    uint64_t ci = active[a0];
    *reinterpret_cast<uint32_t*>(s_cls + lane * 4) = *reinterpret_cast<const uint32_t*>(eng.cls + lane * 4);
    eng.cls = s_cls;
    for (uint64_t a = a0; a < n_active;) {
        const ChunkRange r = chunk_range(g, ci);
        const uint64_t w16 = r.w & ~uint64_t(15);
        const bool staged = r.hi - w16 <= kFillStage;
        if (staged)
            for (uint64_t o = uint64_t(lane) * 16; w16 + o < r.hi; o += 64 * 16) {
                ACGPU_HAY_CHECK(g, w16 + o, 16);
                *reinterpret_cast<uint4*>(s_hay + o) = *reinterpret_cast<const uint4*>(g.hay16 + w16 + o);
            }
        __syncthreads();
        // split [r.lo, r.hi) into 64 sub-ranges of `sub` bytes (the last ones may be empty)
        const uint64_t len = r.hi - r.lo;
        const uint64_t sub = (len + 63) / 64;
        uint64_t lo = r.lo + uint64_t(lane) * sub, hi = lo + sub;
        if (lo > r.hi) lo = r.hi;
        if (hi > r.hi) hi = r.hi;
        uint64_t w = lo >= g.halo ? lo - g.halo : 0;
        if (w < g.cold_floor) w = g.cold_floor;
        const bool sm = ci == 0 && lane == 0 && g.emit_start_matches;
        const bool work = hi > lo || sm;
        FillWalk<E> fw{eng, g};
        uint64_t ev_end[kFillEvents] = {};
        uint32_t ev_sid[kFillEvents] = {}, nev = 0;
        auto lds_byte = [&](uint64_t v) -> uint8_t { return s_hay[v - w16]; };
        auto mem_byte = [&](uint64_t v) -> uint8_t { ACGPU_HAY_CHECK(g, v, 1); return g.hay16[v]; };
        uint32_t c = 0;
        if (work) c = staged ? fw.template run<false>(w, lo, hi, sm, lds_byte, nullptr, ev_end, ev_sid, nev)
                             : fw.template run<false>(w, lo, hi, sm, mem_byte, nullptr, ev_end, ev_sid, nev);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (c) {
            acgpu_match* dst = out + aoff[a] + (incl - c);
            if (nev <= uint32_t(kFillEvents)) {
#pragma unroll
                for (int k = 0; k < kFillEvents; k++)
                    if (uint32_t(k) < nev) dst += fw.write(ev_sid[k], ev_end[k], dst);
            } else if (staged) {
                fw.template run<true>(w, lo, hi, sm, lds_byte, dst, ev_end, ev_sid, nev);
            } else {
                fw.template run<true>(w, lo, hi, sm, mem_byte, dst, ev_end, ev_sid, nev);
            }
        }
        a += gridDim.x;
        if (a < n_active) ci = active[a];
        __syncthreads();  // s_hay is reused by the next chunk
    }
}

// ------------------------------------------------------------------- serial API restatements
// src/automaton.rs:1285-1420 without the prefilter arms
template <class E>
__device__ bool dev_try_find_fwd(const E& eng, const uint8_t* hay, uint64_t start, uint64_t end, bool anchored,
                                 bool earliest, acgpu_match& out) {
    if (start > end) return false;
    uint32_t sid = eng.start(anchored);
    uint64_t at = start;
    bool found = false;
    if (eng.is_match(sid)) {
        const uint32_t pid = eng.match_pattern(sid, 0);
        out.pattern = pid; out._pad = 0; out.end = at; out.start = at - eng.pattern_len(pid);
        found = true;
        if (earliest) return true;
    }
    while (at < end) {
        sid = eng.next(anchored, sid, hay[at]);
        if (eng.is_special(sid)) {
            if (sid == kDevDead) return found;
            if (eng.is_match(sid)) {
                const uint32_t pid = eng.match_pattern(sid, 0);
                const uint64_t e = at + 1, s = e - eng.pattern_len(pid);
                if (!(anchored && s > start)) {
                    out.pattern = pid; out._pad = 0; out.start = s; out.end = e;
                    found = true;
                    if (earliest) return true;
                }
            }
        }
        at++;
    }
    return found;
}

// FindIter, src/automaton.rs:857-936
template <class E>
__global__ void k_find_iter_serial(E eng, SerialArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool anchored = a.anchored != 0;
    // FindIter keeps the caller's Input, earliest flag included (automaton.rs:864-883, :1266)
    const bool earliest = a.match_kind == ACGPU_MATCH_STANDARD || a.earliest != 0;
    uint64_t start = a.span_start, n = 0;
    bool has_last = false;
    uint64_t last_end = 0;
    for (;;) {
        acgpu_match m;
        if (!dev_try_find_fwd(eng, a.hay, start, a.span_end, anchored, earliest, m)) break;
        if (m.start == m.end && has_last && m.end == last_end) {  // handle_overlapping_empty_match :910-920
            start += 1;
            if (!dev_try_find_fwd(eng, a.hay, start, a.span_end, anchored, earliest, m)) break;
        }
        start = m.end;
        has_last = true;
        last_end = m.end;
        if (n < a.cap) a.out[n] = m;
        n++;
    }
    *a.n_out = n;
}

// try_find, src/automaton.rs:1259-1282
template <class E>
__global__ void k_find_serial(E eng, SerialArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool earliest = a.match_kind == ACGPU_MATCH_STANDARD || a.earliest != 0;
    acgpu_match m;
    const bool f = dev_try_find_fwd(eng, a.hay, a.span_start, a.span_end, a.anchored != 0, earliest, m);
    if (f && a.cap > 0) a.out[0] = m;
    *a.n_out = f ? 1 : 0;
}

// ------------------------------------------------------------------------- non-overlapping selection
// One lane walks the (usually sparse) ordered occurrence stream; see select.hpp for the rule.
__global__ void k_select_nonoverlapping(const acgpu_match* __restrict__ S, const uint64_t* __restrict__ n_in,
                                        int match_kind, uint64_t span_start, uint64_t L,
                                        acgpu_match* __restrict__ out, uint64_t cap, uint64_t* __restrict__ n_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    *n_out = select_nonoverlapping(S, *n_in, match_kind, span_start, L,
                                   [&](uint64_t k, const acgpu_match& m) { if (k < cap) out[k] = m; });
}

// ------------------------------------------------------------------------------- generator
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_gen_haystack(uint8_t* __restrict__ dst, uint64_t offset, uint64_t len,
                                                      uint64_t seed, uint32_t lo, uint32_t span) {
    const uint64_t stride = uint64_t(gridDim.x) * 256 * 16;
    for (uint64_t i0 = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * 16; i0 < len; i0 += stride) {
        if (i0 + 16 <= len && ((uintptr_t)(dst + i0) & 15) == 0) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t b = lo + uint32_t(splitmix64(seed ^ (offset + i0 + k)) % span);
                w[k >> 2] |= (b & 0xFF) << (8 * (k & 3));
            }
            *reinterpret_cast<uint4*>(dst + i0) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (uint64_t i = i0; i < len && i < i0 + 16; i++)
                dst[i] = uint8_t(lo + uint32_t(splitmix64(seed ^ (offset + i)) % span));
        }
    }
}

// Read-only streaming kernel: the empirical ceiling of "read every haystack byte once" on this box (acgpu_stream_read;
// 16 bytes per lane, four loads in flight, non-temporal: the best shape of scripts/ubench/stream_ceiling.hip).
__global__ __launch_bounds__(1024) void k_stream_read(const uint4* __restrict__ p, size_t n16, unsigned* __restrict__ sink) {
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t nthreads = size_t(gridDim.x) * blockDim.x;
    unsigned acc = 0;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    size_t i = tid;
    for (; i + 3 * nthreads < n16; i += nthreads * 4) {
        v4u v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + i + size_t(k) * nthreads));
#pragma unroll
        for (int k = 0; k < 4; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    for (; i < n16; i += nthreads) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;   // (never true for real data: keeps the loads)
}

// records of one shard -> global coordinates (multi-device search: every shard searched in local coordinates)
__global__ __launch_bounds__(256) void k_offset_records(acgpu_match* __restrict__ m, uint64_t n, uint64_t off) {
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) { m[i].start += off; m[i].end += off; }
}

DfaEng make_dfa_eng(const DevAutomaton& a) { DfaEng e; e.d = a.dfa; e.cls = a.dfa.classes; return e; }
CnfaEng make_cnfa_eng(const DevAutomaton& a) { CnfaEng e; e.c = a.cnfa; e.cls = a.cnfa.classes; return e; }

}  // namespace

hipError_t launch_walk_count(uint32_t engine, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts,
                             hipStream_t s) {
    const uint32_t halo_tiles = (g.halo + kTile - 1) / kTile;
    const uint64_t blocks = (g.n_chunks + kBlock - 1) / kBlock;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (engine == ENG_DFA) k_walk_count<DfaEng><<<dim3(uint32_t(blocks)), dim3(kBlock), 0, s>>>(make_dfa_eng(a), g, counts, halo_tiles);
    else if (engine == ENG_CNFA) k_walk_count<CnfaEng><<<dim3(uint32_t(blocks)), dim3(kBlock), 0, s>>>(make_cnfa_eng(a), g, counts, halo_tiles);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_walk_fill(uint32_t engine, const DevAutomaton& a, const ScanGeom& g, const uint64_t* active,
                            const uint64_t* totals, uint64_t cap, uint64_t max_blocks, const uint64_t* aoff,
                            acgpu_match* out, hipStream_t s, const unsigned long long* gate) {
    uint64_t blocks = max_blocks < g.n_chunks ? max_blocks : g.n_chunks;  // one wavefront per non-empty chunk, grid-stride
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (engine == ENG_DFA) k_walk_fill<DfaEng><<<dim3(uint32_t(blocks)), dim3(64), 0, s>>>(make_dfa_eng(a), g, active, totals, cap, aoff, out, gate);
    else if (engine == ENG_CNFA) k_walk_fill<CnfaEng><<<dim3(uint32_t(blocks)), dim3(64), 0, s>>>(make_cnfa_eng(a), g, active, totals, cap, aoff, out, gate);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_scan(const ScanScratch& sc, uint64_t n_chunks, hipStream_t s) {
    const uint64_t nb = (n_chunks + kScanSpan - 1) / kScanSpan;
    if (nb == 0 || nb > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (sc.packed) {   // (event_order.hip: always with offsets)
        if (!sc.offsets) return hipErrorInvalidValue;
        k_scan_block_sums<true><<<dim3(uint32_t(nb)), dim3(256), 0, s>>>(sc.packed, n_chunks, sc.bsum, sc.bact);
        k_scan_tops<<<dim3(1), dim3(256), 0, s>>>(sc.bsum, sc.bact, nb, sc.totals, sc.host_totals, sc.extra32);
        k_scan_write<true, true><<<dim3(uint32_t(nb)), dim3(256), 0, s>>>(sc.packed, n_chunks, sc.bsum, sc.bact, sc.offsets, sc.active, sc.aoff);
        return hipGetLastError();
    }
    k_scan_block_sums<false><<<dim3(uint32_t(nb)), dim3(256), 0, s>>>(sc.counts, n_chunks, sc.bsum, sc.bact);
    k_scan_tops<<<dim3(1), dim3(256), 0, s>>>(sc.bsum, sc.bact, nb, sc.totals, sc.host_totals, sc.extra32);
    if (sc.offsets) k_scan_write<true><<<dim3(uint32_t(nb)), dim3(256), 0, s>>>(sc.counts, n_chunks, sc.bsum, sc.bact, sc.offsets, sc.active, sc.aoff);
    else k_scan_write<false><<<dim3(uint32_t(nb)), dim3(256), 0, s>>>(sc.counts, n_chunks, sc.bsum, sc.bact, nullptr, sc.active, sc.aoff);
    return hipGetLastError();
}

hipError_t launch_find_iter_serial(uint32_t engine, const DevAutomaton& a, const SerialArgs& args, hipStream_t s) {
    if (engine == ENG_DFA) k_find_iter_serial<DfaEng><<<dim3(1), dim3(64), 0, s>>>(make_dfa_eng(a), args);
    else if (engine == ENG_CNFA) k_find_iter_serial<CnfaEng><<<dim3(1), dim3(64), 0, s>>>(make_cnfa_eng(a), args);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_find_serial(uint32_t engine, const DevAutomaton& a, const SerialArgs& args, hipStream_t s) {
    if (engine == ENG_DFA) k_find_serial<DfaEng><<<dim3(1), dim3(64), 0, s>>>(make_dfa_eng(a), args);
    else if (engine == ENG_CNFA) k_find_serial<CnfaEng><<<dim3(1), dim3(64), 0, s>>>(make_cnfa_eng(a), args);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_select_nonoverlapping(const acgpu_match* S, const uint64_t* n_in, int match_kind, uint64_t span_start,
                                        uint64_t L, acgpu_match* out, uint64_t cap, uint64_t* n_out, hipStream_t s) {
    k_select_nonoverlapping<<<dim3(1), dim3(64), 0, s>>>(S, n_in, match_kind, span_start, L, out, cap, n_out);
    return hipGetLastError();
}

hipError_t launch_offset_records(acgpu_match* m, uint64_t n, uint64_t off, hipStream_t s) {
    if (n == 0 || off == 0) return hipSuccess;
    const uint64_t blocks = std::min<uint64_t>((n + 255) / 256, 4096);
    k_offset_records<<<dim3(uint32_t(blocks)), dim3(256), 0, s>>>(m, n, off);
    return hipGetLastError();
}

hipError_t launch_stream_read(const uint8_t* src, size_t len, unsigned* sink, hipStream_t s) {
    const size_t n16 = len / 16;
    if (n16 == 0 || (reinterpret_cast<uintptr_t>(src) & 15)) return hipErrorInvalidValue;
    k_stream_read<<<dim3(4096), dim3(1024), 0, s>>>(reinterpret_cast<const uint4*>(src), n16, sink);
    return hipGetLastError();
}

hipError_t launch_gen_haystack(uint8_t* dst, uint64_t offset, size_t len, uint64_t seed, uint32_t lo, uint32_t span,
                               hipStream_t s) {
    if (len == 0) return hipSuccess;
    uint64_t blocks = (len + 256 * 16 - 1) / (256 * 16);
    if (blocks > 65536) blocks = 65536;
    k_gen_haystack<<<dim3(uint32_t(blocks)), dim3(256), 0, s>>>(dst, offset, len, seed, lo, span);
    return hipGetLastError();
}

}  // namespace acgpu
