// Ordered records from a LARGE set of level-3 events of the prefix filters (pf_scan.hip / pfx_scan.hip): match-dense
// inputs produce far more occurrences than the all-pairs rank of the direct mode can order (n^2), and re-walking every
// non-empty chunk (k_hot_fill / k_walk_fill) re-reads a large part of the haystack.  The events arrive unordered (the
// wavefronts append them as they go), but every event carries its end position, and the stream order is "ascending end,
// then the longer occurrence first" = ascending key (k_ev_rank in pf_scan.hip).  So the order is a bucket pass, O(n):
//
//
//   histogram   bucket = 2^shift end positions (2 KiB, or more when the events are few for the span: about four events per
//               bucket -- config 5's 45 k occurrences in 8 GiB paid for zeroing, scanning and missing the cache on four
//               million 2 KiB buckets); events and records per bucket -- ONE 64-bit global atomic per event on the bucket's
//               word {records, events << 32} (spread over the buckets; two 32-bit ones took 50 us per million events: the
//               memory-side atomic rate), whose old value is the event's arrival slot in its bucket.  Either a pass of
//               its own (k_eo_hist), or -- the FUSED chain -- done by the scan kernels themselves when they append the
//               events (PfEoHist in hot.hpp: the atomics hide behind the scan; 31 us per million events as a pass)
//   scan        ONE launch (k_eo_scan): exclusive prefix of the packed words = the bucket's slice of the OUTPUT (low half:
//               records) and of the grouped events (high half); every workgroup sums a contiguous segment, publishes
//               its sum and waits for the sums of the workgroups before it (they were dispatched before it: no
//               deadlock whatever else runs on the device; the wait is bounded and falls back to summing the buckets in
//               front of the segment itself), then writes its prefixes.
//               (Three launches of the generic scan of kernels.hip until round 6: 17 us of launch floors.)
//   scatter     one thread per event: to its bucket's slice of the grouped list (dense: one slot per EVENT)
//   emit        one thread per GROUPED event (neighbouring threads = neighbouring events of the same or the next bucket:
//               the bucket words, the slices and the records they write are neighbours too; round 5 went by arrival
//               order, every access a 64-byte sector of its own: 28 us per million events) ranks it among its bucket's
//               events and writes its records; buckets of more than 48 events (match-saturated text: thousands per
//               bucket), one wavefront each: a second bucket level in LDS -- 2 048 bins of 2^(shift-11) end positions,
//               scan, scatter, and an all-pairs inside each bin (2 KiB buckets: the occurrences ending at one position,
//               at most one per pattern length).
//
// Unfused chain: k_eo_zero, k_eo_hist, k_eo_scan, k_eo_scatter, k_eo_emit_small, k_eo_emit_large(, k_eo_done).  Fused chain
// (the enqueue form while results are dense): the scan kernel, k_eo_scan, k_eo_scatter, k_eo_emit_small, k_eo_emit_large --
// whose last workgroup also reports the totals (to the device words and to page-locked host memory) and re-arms the event
// counters --, and k_eo_zero BEHIND them (the bucket words are zero between calls; the host does not wait for that launch).
// Each reads the event / record counts from device memory and returns at once when the set is small enough for the
// all-pairs rank, too large for its buffers, or the scan was abandoned, so the enqueue-only form can queue them behind a
// scan without a host decision.  A single persistent kernel with grid barriers was tried and dropped: two of them on one
// device -- two streams, two host threads -- wait for each other's CUs forever.  Hand-written throughout (round 2 used
// hipCUB's radix sort + scan here: ~15 library launches, 0.33 ms of a 1.8 ms natural-text step).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hot.hpp"
#include "launch_util.hpp"

namespace acgpu {

namespace {

constexpr uint32_t kEoShift = 11;                 // smallest bucket = 2 KiB of end positions
constexpr uint32_t kEoMaxShift = 24;
constexpr uint32_t kEoBins = 1u << kEoShift;      // bins of the second level (k_eo_emit_large)
constexpr uint32_t kEoSmall = 48;                 // buckets up to this many events: one thread per event
constexpr int kEoBlock = 256, kEoWaves = kEoBlock / 64;
constexpr size_t kEoLds = size_t(kEoWaves) * 3 * kEoBins * 4;   // per wavefront: three arrays of one word per end position
constexpr uint32_t kEoScanBlocks = 1024;          // workgroups of k_eo_scan at most (= published sums)
constexpr uint32_t kEoScanTile = 2048;            // buckets per tile: 256 threads x 8
#ifndef ACGPU_EO_SPIN
#define ACGPU_EO_SPIN (1u << 22)   // (lib/exp builds with 0 exercise the fallback of k_eo_scan)
#endif
constexpr uint32_t kEoSpin = ACGPU_EO_SPIN;       // polls of a predecessor's word before a workgroup of the scan sums the buckets itself (~0.1 s)

// flag words at the head of the work area
enum { kFlagLarge = 0, kFlagTicket = 1, kFlagScanErr = 2 };

struct EoArgs {
    const PfEvent* ev;
    // the counts of this scan: totals[0] records, totals[1] events (written by k_ev_rank) -- or, fused chain, the scan's own
    // counters ctr[0] events, ctr[1] records, ctr[2] != 0: abandoned
    const unsigned long long* n_events;
    const unsigned long long* n_records;
    const unsigned long long* abandoned;   // nullptr: not asked
    uint64_t min_events;         // sets up to this size were ordered by the all-pairs rank already
    uint64_t max_events;         // capacity of `ev`, `slot`, `tmp`, `tmp2`
    uint64_t max_records;        // capacity of the output (< 2^32: the prefixes are packed)
    uint64_t origin;             // end position - 1 - origin = offset into the bucket grid (origin = shard begin)
    uint64_t n_buckets;
    uint32_t shift;              // bucket = 2^shift end positions
    uint32_t* flags;             // [4] kFlag*
    unsigned long long* agg;     // [kEoScanBlocks] k_eo_scan: bit 63 = the workgroup's sum (the other bits) is published
    unsigned long long* bb;      // [n_buckets] records of the bucket | events of the bucket << 32
    uint64_t* offsets;           // [n_buckets] exclusive prefix of bb: first record | first grouped event << 32
    uint32_t* slot;              // [max_events] arrival slot of the event in its bucket
    PfEvent* tmp;                // [max_events] events grouped by bucket
    PfEvent* tmp2;               // [max_events] ... and by end position inside large buckets
    // what the last workgroup of the chain reports (any of them may be nullptr)
    uint64_t* done_totals;       // unfused enqueue-only form: totals[1] <- 0 once the records are delivered
    unsigned long long* rearm;   // fused chain: the scan's counters, zeroed for the next scan ...
    uint64_t* fin_totals;        // ... after {records, delivered ? 0 : UINT64_MAX} went here (device) ...
    uint64_t* fin_host;          // ... and {records, the same, events} here (page-locked host memory), then fin_seq at [3]
    uint64_t fin_seq;
};

__device__ __forceinline__ uint64_t eo_pos(const EoArgs& a, const PfEvent& e) { return (e.key >> 16) - 1 - a.origin; }

// bin of the second level: 2^(shift-11) end positions of the event's bucket
__device__ __forceinline__ uint32_t eo_bin(const EoArgs& a, const PfEvent& e) {
    return uint32_t((eo_pos(a, e) & ((uint64_t(1) << a.shift) - 1)) >> (a.shift - kEoShift));
}

__device__ __forceinline__ bool eo_active(const EoArgs& a, uint64_t& n) {
    n = *a.n_events;
    if (a.abandoned && *a.abandoned) return false;
    return n > a.min_events && n <= a.max_events && *a.n_records <= a.max_records;
}

__device__ __forceinline__ void eo_write(const DfaEng& eng, const uint32_t* __restrict__ hid2sid, const uint32_t* __restrict__ own_pid,
                                         const PfEvent& e, acgpu_match* __restrict__ dst) {
    const uint64_t end = e.key >> 16, len = 0xFFFFull - (e.key & 0xFFFFull);
    if (e.cnt == 1) {   // the node's only own pattern: one gather instead of three dependent ones (state id, list offset, list)
        acgpu_match m; m.pattern = own_pid[e.node]; m._pad = 0; m.start = end - len; m.end = end;
        dst[0] = m;
        return;
    }
    const uint32_t sid = hid2sid[e.node];
    for (uint32_t k = 0; k < e.cnt; k++) {
        acgpu_match m; m.pattern = eng.match_pattern(sid, k); m._pad = 0; m.start = end - len; m.end = end;
        dst[k] = m;
    }
}

// (separate launches, no grid-wide barrier inside a kernel: a persistent kernel whose workgroups wait for each other
// deadlocks as soon as two of them -- two streams, two host threads -- share the device)
__global__ __launch_bounds__(256) void k_eo_zero(uint64_t* __restrict__ p, uint64_t words) {
    // (not gated on eo_active: the bucket scan always runs and must not read counters nobody initialised)
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < words; i += uint64_t(gridDim.x) * 256) p[i] = 0;
}

// events and records per bucket, the event's arrival slot in its bucket (unfused chain)
// (Tried: the workgroup that finishes last -- a ticket -- also writes the exclusive prefix of the buckets, to save the three
// scan launches when the buckets are few.  72-77 us instead of 5 + 14: what one workgroup reads of the other workgroups'
// atomics has to come past its L2 with device-scope loads, a few microseconds per dependent trip, and an agent-scope fence
// is an L2 write-back.  Small launches that follow each other on a stream cost 4.5-5.5 us each and no gap.)
__global__ __launch_bounds__(256) void k_eo_hist(EoArgs a) {
    uint64_t n;
    if (!eo_active(a, n)) return;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const PfEvent e = a.ev[i];
        const uint64_t b = eo_pos(a, e) >> a.shift;
        // (records of a bucket < 2^32: the whole set has fewer)
        a.slot[i] = uint32_t(atomicAdd(&a.bb[b], (1ull << 32) | e.cnt) >> 32);
    }
}

// sum of x over the workgroup (256 threads), in every thread
__device__ __forceinline__ uint64_t eo_block_sum(uint64_t x, uint64_t* s_w) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(x)), o, 64)), hi = uint32_t(__shfl_xor(int(uint32_t(x >> 32)), o, 64));
        x += uint64_t(lo) | (uint64_t(hi) << 32);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = x;
    __syncthreads();
    return s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// Exclusive prefix of the packed bucket words, one launch.  Workgroup k owns buckets [k seg, (k + 1) seg): (1) their sum,
// published as {value, flag} with device-scope stores; (2) the sums of the workgroups before it -- one parallel round of
// device-scope loads, 256 predecessors at a time; (3) its prefixes, tile by tile (the segment is in L2 from (1)).  Also
// notes whether some bucket holds more than kEoSmall events (k_eo_emit_large runs only then).
__global__ __launch_bounds__(256) void k_eo_scan(EoArgs a, uint64_t seg) {
    __shared__ uint64_t s_w[4];
    __shared__ uint64_t s_scan[4];
    uint64_t n;
    if (!eo_active(a, n)) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lo = uint64_t(blockIdx.x) * seg, hi = lo + seg < a.n_buckets ? lo + seg : a.n_buckets;
    uint64_t sum = 0;
    uint32_t big = 0;
    for (uint64_t i = lo + tid; i < hi; i += 256) {
        const uint64_t v = a.bb[i];
        sum += v;
        big |= uint32_t(v >> 32) > kEoSmall ? 1u : 0u;
    }
    if (big) a.flags[kFlagLarge] = 1u;
    sum = eo_block_sum(sum, s_w);
    // (value and flag in ONE word, relaxed device-scope accesses: nothing else is communicated, so no fence -- a release /
    // acquire pair at device scope is an L2 write-back and an invalidate per workgroup: 41 us for 256 workgroups)
    constexpr unsigned long long kPublished = 1ull << 63;
    if (tid == 0) __hip_atomic_store(&a.agg[blockIdx.x], sum | kPublished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t before = 0;
    bool gave_up = false;
    for (uint32_t j = tid; j < blockIdx.x && !gave_up; j += 256) {
        uint32_t polls = 0;
        unsigned long long v;
        while (((v = __hip_atomic_load(&a.agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & kPublished) == 0) {
            if (++polls > kEoSpin) { gave_up = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        before += v & ~kPublished;
    }
    // (a workgroup before this one that does not publish within ~0.1 s -- it cannot happen while workgroups are dispatched in
    // order; the wait is bounded all the same -- : this workgroup adds up the buckets in front of its segment itself)
    if (__syncthreads_or(gave_up ? 1 : 0)) {
        before = 0;
        for (uint64_t i = tid; i < lo; i += 256) before += a.bb[i];
    }
    uint64_t carry = eo_block_sum(before, s_w);
    for (uint64_t t = lo; t < hi; t += kEoScanTile) {
        const uint64_t i0 = t + uint64_t(tid) * 8;
        uint64_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = i0 + k < hi ? a.bb[i0 + k] : 0;
        uint64_t mine = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { const uint64_t x = v[k]; v[k] = mine; mine += x; }
        uint64_t inc = mine;   // inclusive over the wavefront
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t l = uint32_t(__shfl_up(int(uint32_t(inc)), o, 64)), h = uint32_t(__shfl_up(int(uint32_t(inc >> 32)), o, 64));
            if (int(lane) >= o) inc += uint64_t(l) | (uint64_t(h) << 32);
        }
        __syncthreads();
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        uint64_t wbase = 0, tile = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) { if (w < wave) wbase += s_scan[w]; tile += s_scan[w]; }
        const uint64_t base = carry + wbase + (inc - mine);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (i0 + k < hi) a.offsets[i0 + k] = base + v[k];
        carry += tile;
    }
}

__device__ __forceinline__ bool eo_scanned(const EoArgs& a, uint64_t& n) { return eo_active(a, n) && a.flags[kFlagScanErr] == 0u; }

// every event to its bucket's slice of the grouped list
__global__ __launch_bounds__(256) void k_eo_scatter(EoArgs a) {
    uint64_t n;
    if (!eo_scanned(a, n)) return;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const PfEvent e = a.ev[i];
        a.tmp[(a.offsets[eo_pos(a, e) >> a.shift] >> 32) + a.slot[i]] = e;
    }
}

// Buckets of up to kEoSmall events (natural text against a dictionary: two or three per bucket): one thread per grouped
// event ranks it against its bucket's events (its neighbours) and writes its records.
// (Tried in round 5: one thread per BUCKET.  48.5 us instead of 29.6 us per launch: a quarter of the threads, each with its
// bucket's events in series behind the own_pid gather.)
__global__ __launch_bounds__(256) void k_eo_emit_small(EoArgs a, DfaEng eng, const uint32_t* __restrict__ hid2sid,
                                                       const uint32_t* __restrict__ own_pid, acgpu_match* __restrict__ out) {
    uint64_t n;
    if (!eo_scanned(a, n)) return;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const PfEvent e = a.tmp[i];
        const uint64_t b = eo_pos(a, e) >> a.shift;
        const uint32_t m = uint32_t(a.bb[b] >> 32);
        if (m > kEoSmall) continue;
        const uint64_t o = a.offsets[b];
        const PfEvent* mates = a.tmp + (o >> 32);
        uint32_t r = 0;
        for (uint32_t j = 0; j < m; j++) {
            const PfEvent q = mates[j];
            if (q.key < e.key) r += q.cnt;
        }
        eo_write(eng, hid2sid, own_pid, e, out + (o & 0xFFFFFFFFull) + r);
    }
}

// what the chain reports when it is through (one thread of its last workgroup)
__device__ __forceinline__ void eo_finish(const EoArgs& a) {
    uint64_t n;
    const bool ran = eo_scanned(a, n);
    const uint64_t records = *a.n_records;
    if (a.done_totals && ran) a.done_totals[1] = 0;
    if (a.fin_totals || a.fin_host) {
        const bool abandoned = a.abandoned && *a.abandoned;
        const bool ok = ran || (n == 0 && !abandoned && a.min_events == 0);
        const uint64_t t1 = ok ? 0ull : ~0ull;
        if (a.fin_totals) { a.fin_totals[0] = records; a.fin_totals[1] = t1; }
        if (a.fin_host) {
            a.fin_host[0] = records; a.fin_host[1] = t1; a.fin_host[2] = abandoned ? ~0ull : n;
            // (a host thread polling word 3 sees the three words above once it reads fin_seq there)
            __hip_atomic_store(&a.fin_host[3], a.fin_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (a.rearm) { a.rearm[0] = 0ull; a.rearm[1] = 0ull; a.rearm[2] = 0ull; }
}

// Larger buckets (match-saturated text: thousands of events per bucket), one wavefront per bucket: a second bucket
// level in LDS -- one bin per end position; scan; scatter; all-pairs inside each bin (the occurrences ending at one
// position: at most one per pattern length).  Works only if k_eo_scan saw such a bucket; its last workgroup to finish
// (a ticket; workgroup 0 when there was nothing to do) closes the chain: eo_finish.
__global__ __launch_bounds__(kEoBlock) void k_eo_emit_large(EoArgs a, DfaEng eng, const uint32_t* __restrict__ hid2sid,
                                                            const uint32_t* __restrict__ own_pid, acgpu_match* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bins[];
    uint64_t n;
    const bool finish = a.done_totals || a.fin_totals || a.fin_host || a.rearm;
    if (!eo_scanned(a, n) || a.flags[kFlagLarge] == 0) {
        // (the other workgroups read the same words and return as well: nobody is left to wait for)
        if (finish && blockIdx.x == 0 && threadIdx.x == 0) eo_finish(a);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* ecnt = s_bins + size_t(wave) * 3 * kEoBins;   // events per end position -> exclusive prefix
    uint32_t* rcnt = ecnt + kEoBins;                         // records per end position -> exclusive prefix
    uint32_t* fill = rcnt + kEoBins;
    const uint64_t wid = uint64_t(blockIdx.x) * kEoWaves + wave, nwaves = uint64_t(gridDim.x) * kEoWaves;
    for (uint64_t b = wid; b < a.n_buckets; b += nwaves) {
        const uint32_t m = uint32_t(a.bb[b] >> 32);
        if (m <= kEoSmall) continue;
        const uint64_t o = a.offsets[b];
        const uint64_t base = o >> 32, rbase = o & 0xFFFFFFFFull;
        for (uint32_t i = lane; i < kEoBins; i += 64) { ecnt[i] = 0; rcnt[i] = 0; fill[i] = 0; }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < m; i += 64) {
            const PfEvent e = a.tmp[base + i];
            const uint32_t eo = eo_bin(a, e);
            atomicAdd(&ecnt[eo], 1u);
            atomicAdd(&rcnt[eo], e.cnt);
        }
        __builtin_amdgcn_wave_barrier();
        {   // exclusive prefix over the 2 048 bins: 32 consecutive bins per lane + a wave scan of the lane sums
            uint32_t es = 0, rs = 0;
            for (uint32_t k = 0; k < kEoBins / 64; k++) { es += ecnt[lane * (kEoBins / 64) + k]; rs += rcnt[lane * (kEoBins / 64) + k]; }
            uint32_t ei = es, ri = rs;
#pragma unroll
            for (int o2 = 1; o2 < 64; o2 <<= 1) {
                const uint32_t te = uint32_t(__shfl_up(int(ei), o2, 64)), tr = uint32_t(__shfl_up(int(ri), o2, 64));
                if (lane >= o2) { ei += te; ri += tr; }
            }
            uint32_t ep = ei - es, rp = ri - rs;
            for (uint32_t k = 0; k < kEoBins / 64; k++) {
                const uint32_t idx = lane * (kEoBins / 64) + k;
                const uint32_t ce = ecnt[idx], cr = rcnt[idx];
                ecnt[idx] = ep; rcnt[idx] = rp;
                ep += ce; rp += cr;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < m; i += 64) {
            const PfEvent e = a.tmp[base + i];
            const uint32_t eo = eo_bin(a, e);
            a.tmp2[base + ecnt[eo] + atomicAdd(&fill[eo], 1u)] = e;
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < m; i += 64) {
            // (volatile: written a moment ago by other lanes of this wavefront -- past a possibly stale L1 line)
            const volatile PfEvent* t2 = a.tmp2 + base;
            PfEvent e; e.key = t2[i].key; e.node = t2[i].node; e.cnt = t2[i].cnt;
            const uint32_t eo = eo_bin(a, e);
            const uint32_t g0 = ecnt[eo], g1 = g0 + fill[eo];
            uint32_t r = rcnt[eo];
            for (uint32_t j = g0; j < g1; j++)
                if (t2[j].key < e.key) r += t2[j].cnt;
            eo_write(eng, hid2sid, own_pid, e, out + rbase + r);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!finish) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&a.flags[kFlagTicket], 1u) == gridDim.x - 1) { a.flags[kFlagTicket] = 0u; eo_finish(a); }
    }
}

struct Layout {
    size_t flags, agg, bb, offsets, slot, tmp, tmp2, total;
};
Layout layout(uint64_t max_events, uint64_t nb) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    Layout L{};
    size_t o = 0;
    L.flags = o; o += up(16);
    L.agg = o; o += up(kEoScanBlocks * 8);
    L.bb = o; o += up(nb * 8);
    L.offsets = o; o += up(nb * 8);
    L.slot = o; o += up(max_events * 4);
    L.tmp = o; o += up(max_events * sizeof(PfEvent));
    L.tmp2 = o; o += up(max_events * sizeof(PfEvent));
    L.total = o;
    return L;
}
// About four events per bucket at `max_events`, never below 2 KiB (the synchronous pipelines pass the exact count; the
// enqueue-only form its event capacity, span / 64: 2 KiB buckets).
uint32_t eo_shift(uint64_t max_events, uint64_t max_records, uint64_t span_bytes) {
    uint32_t sh = kEoShift;
    if (max_records >= (uint64_t(1) << 31)) return sh;
    while (sh < kEoMaxShift && (span_bytes >> (sh + 1)) * 4 >= std::max<uint64_t>(max_events, 1)) sh++;
    return sh;
}
}  // namespace
uint32_t event_order_shift(uint64_t max_events, uint64_t max_records, uint64_t span_bytes) { return eo_shift(max_events, max_records, span_bytes); }
namespace {
uint64_t buckets_of(uint64_t span_bytes, uint32_t shift) { return std::max<uint64_t>(1, (span_bytes + (uint64_t(1) << shift) - 1) >> shift) + 1; }

}  // namespace

// The words a pass expects to be zero on entry (the flags, the scan's published sums and the bucket counters: the first
// bytes of `work`); a caller that has a kernel in flight anyway (k_ev_write) zeroes them there and passes zeroed = true.
size_t event_order_zero_bytes(uint64_t max_events, uint64_t max_records, uint64_t span_bytes) {
    const Layout L = layout(std::max<uint64_t>(max_events, 1), buckets_of(span_bytes, eo_shift(max_events, max_records, span_bytes)));
    static_assert(sizeof(unsigned long long) == 8, "");
    return L.offsets;   // flags | agg | bb
}

size_t event_order_work_bytes(uint64_t max_events, uint64_t max_records, uint64_t span_bytes) {
    return layout(std::max<uint64_t>(max_events, 1), buckets_of(span_bytes, eo_shift(max_events, max_records, span_bytes))).total;
}

PfEoHist event_order_hist(uint64_t max_events, uint64_t max_records, uint64_t span_begin, uint64_t span_bytes, void* work) {
    const uint32_t shift = eo_shift(max_events, max_records, span_bytes);
    const Layout L = layout(std::max<uint64_t>(max_events, 1), buckets_of(span_bytes, shift));
    uint8_t* w = static_cast<uint8_t*>(work);
    PfEoHist h;
    h.bb = reinterpret_cast<unsigned long long*>(w + L.bb); h.slot = reinterpret_cast<uint32_t*>(w + L.slot);
    h.origin = span_begin; h.shift = shift;
    return h;
}

hipError_t launch_event_order_zero(void* work, size_t bytes, hipStream_t s) {
    const uint64_t words = bytes / 8;
    const uint32_t blocks = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((words + 255) / 256, uint64_t(device_cus()) * 16)));
    k_eo_zero<<<dim3(blocks), dim3(256), 0, s>>>(static_cast<uint64_t*>(work), words);
    return hipGetLastError();
}

hipError_t launch_event_order_emit(const HotTables& h, const DevAutomaton& a, const void* events, const uint64_t* totals,
                                   uint64_t min_events, uint64_t max_events, uint64_t max_records, uint64_t span_begin,
                                   uint64_t span_bytes, void* work, acgpu_match* out, hipStream_t s, uint64_t* done_totals,
                                   bool zeroed, const EoFused* fused) {
    const uint32_t shift = eo_shift(max_events, max_records, span_bytes);
    const uint64_t nb = buckets_of(span_bytes, shift);
    const Layout L = layout(std::max<uint64_t>(max_events, 1), nb);
    uint8_t* w = static_cast<uint8_t*>(work);
    EoArgs ea{};
    ea.ev = static_cast<const PfEvent*>(events);
    if (fused) {
        ea.n_events = fused->ctr; ea.n_records = fused->ctr + 1; ea.abandoned = fused->ctr + 2;
        ea.rearm = fused->ctr; ea.fin_totals = fused->totals; ea.fin_host = fused->host_totals; ea.fin_seq = fused->seq;
    } else {
        ea.n_events = reinterpret_cast<const unsigned long long*>(totals) + 1; ea.n_records = reinterpret_cast<const unsigned long long*>(totals);
    }
    ea.min_events = min_events; ea.max_events = max_events;
    ea.max_records = std::min<uint64_t>(max_records, 0xFFFFFFFFull); ea.origin = span_begin; ea.n_buckets = nb; ea.shift = shift;
    ea.flags = reinterpret_cast<uint32_t*>(w + L.flags);
    ea.agg = reinterpret_cast<unsigned long long*>(w + L.agg);
    ea.bb = reinterpret_cast<unsigned long long*>(w + L.bb);
    ea.offsets = reinterpret_cast<uint64_t*>(w + L.offsets);
    ea.slot = reinterpret_cast<uint32_t*>(w + L.slot);
    ea.tmp = reinterpret_cast<PfEvent*>(w + L.tmp); ea.tmp2 = reinterpret_cast<PfEvent*>(w + L.tmp2);
    ea.done_totals = done_totals;
    {   // the large-bucket kernel wants kEoLds bytes of dynamic LDS: a device that cannot give them (not gfx950) has no order pass
        if (device_max_lds() < int(kEoLds)) return hipErrorInvalidConfiguration;
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_eo_emit_large), int(kEoLds)); e != hipSuccess) return e;
    DfaEng eng; eng.d = a.dfa; eng.cls = a.dfa.classes;
    const uint32_t eblocks = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((max_events + 255) / 256, uint64_t(device_cus()) * 16)));
    if (!fused) {
        if (!zeroed) if (hipError_t e = launch_event_order_zero(work, L.offsets, s); e != hipSuccess) return e;
        k_eo_hist<<<dim3(eblocks), dim3(256), 0, s>>>(ea);
    }
    // segments: whole tiles, at most kEoScanBlocks of them
    uint64_t seg = (nb + kEoScanBlocks - 1) / kEoScanBlocks;
    seg = std::max<uint64_t>(kEoScanTile, (seg + kEoScanTile - 1) / kEoScanTile * kEoScanTile);
    const uint32_t sblocks = uint32_t((nb + seg - 1) / seg);
    k_eo_scan<<<dim3(sblocks), dim3(256), 0, s>>>(ea, seg);
    k_eo_scatter<<<dim3(eblocks), dim3(256), 0, s>>>(ea);
    k_eo_emit_small<<<dim3(eblocks), dim3(256), 0, s>>>(ea, eng, h.hid2sid, h.own_pid, out);
    k_eo_emit_large<<<dim3(uint32_t(std::min<int>(device_cus(), 1024))), dim3(kEoBlock), kEoLds, s>>>(ea, eng, h.hid2sid, h.own_pid, out);
    return hipGetLastError();
}

}  // namespace acgpu
