// Ordered records from a LARGE set of level-3 events of the prefix filters (pf_scan.hip / pfx_scan.hip): match-dense
// inputs produce far more occurrences than the all-pairs rank of the direct mode can order (n^2), and re-walking every
// non-empty chunk (k_hot_fill / k_walk_fill) re-reads a large part of the haystack.  The events arrive unordered (the
// wavefronts append them as they go), but every event carries its end position, and the stream order is "ascending end,
// then the longer occurrence first" = ascending key (k_ev_rank in pf_scan.hip).  So the order is a bucket pass, O(n):
//
//
//   histogram   one thread per event: bucket = 2^shift end positions (2 KiB, or more when the events are few for the span:
//               about four events per bucket -- config 5's 45 k occurrences in 8 GiB paid for zeroing, scanning and
//               missing the cache on four million 2 KiB buckets); events and records per bucket -- ONE 64-bit
//               global atomic per event on the bucket's word {records, events << 32} (spread over the buckets; two
//               32-bit ones took 50 us per million events: the memory-side atomic rate), whose old value is the event's
//               arrival slot in its bucket
//   scan        exclusive prefix of the records per bucket = the bucket's slice of the output AND of the scratch array
//               (every event stands for >= 1 record: slices of records are large enough for the events)
//   scatter     one thread per event: to its bucket's slice
//   emit        buckets of <= 48 events: one thread per event ranks it among the bucket's events and writes its records;
//               larger buckets (match-saturated text: thousands per bucket), one wavefront each: a second bucket level in
//               LDS -- 2 048 bins of 2^(shift-11) end positions, scan, scatter, and an all-pairs inside each bin (2 KiB
//               buckets: the occurrences ending at one position, at most one per pattern length).
//
// Nine small launches (k_eo_zero, k_eo_hist, the three scan kernels of kernels.hip, k_eo_scatter, k_eo_emit_small,
// k_eo_emit_large, k_eo_done): each reads the event / record counts from device memory and returns at once when the
// set is small enough for the all-pairs rank, too large for its buffers, or the scan was abandoned, so the enqueue-only
// form can queue them behind a scan without a host decision (it does so while the automaton's recent results were
// dense, capi_enqueue.cpp).  A single persistent kernel with grid barriers was tried and dropped: two of them on one device --
// two streams, two host threads -- wait for each other's CUs forever.  Hand-written throughout (round 2 used hipCUB's
// radix sort + scan here: ~15 library launches, 0.33 ms of a 1.8 ms natural-text step).
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

struct EoArgs {
    const PfEvent* ev;
    const uint64_t* totals;      // [0] records, [1] events of this scan (written by k_ev_rank)
    uint64_t min_events;         // sets up to this size were ordered by the all-pairs rank already
    uint64_t max_events;         // capacity of `ev`
    uint64_t max_records;        // capacity of tmp / tmp2 / the output
    uint64_t origin;             // end position - 1 - origin = offset into the bucket grid (origin = shard begin)
    uint64_t n_buckets;
    uint32_t shift;              // bucket = 2^shift end positions
    unsigned long long* bb;      // [n_buckets] records of the bucket | events of the bucket << 32
    uint64_t* offsets;           // [n_buckets] exclusive prefix of brec: the bucket's slice of the output and of tmp
    uint32_t* slot;              // [max_events] arrival slot of the event in its bucket
    PfEvent* tmp;                // [max_records] events grouped by bucket
    PfEvent* tmp2;               // [max_records] ... and by end position inside large buckets
    uint32_t* large;             // [1] set by k_eo_hist when some bucket holds more than kEoSmall events
    uint64_t* done_totals;       // enqueue-only form: totals[1] <- 0 once the records are delivered (nullptr: not wanted)
};

__device__ __forceinline__ uint64_t eo_pos(const EoArgs& a, const PfEvent& e) { return (e.key >> 16) - 1 - a.origin; }

// bin of the second level: 2^(shift-11) end positions of the event's bucket
__device__ __forceinline__ uint32_t eo_bin(const EoArgs& a, const PfEvent& e) {
    return uint32_t((eo_pos(a, e) & ((uint64_t(1) << a.shift) - 1)) >> (a.shift - kEoShift));
}

__device__ __forceinline__ bool eo_active(const EoArgs& a, uint64_t& n) {
    n = a.totals[1];
    return n > a.min_events && n <= a.max_events && a.totals[0] <= a.max_records;
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
__global__ __launch_bounds__(256) void k_eo_zero(EoArgs a) {
    // (not gated on eo_active: the bucket scan between k_eo_hist and k_eo_scatter always runs and must not read counters
    // nobody initialised -- its results are unused when the pass is inactive, but sanitizers flag the reads)
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < a.n_buckets; i += uint64_t(gridDim.x) * 256) a.bb[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.large = 0;
}

// events and records per bucket, the event's arrival slot in its bucket
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
        // (records of a bucket < 2^32: 2 048 end positions x at most 2^17 patterns each; larger buckets only with fewer than
        // 2^31 records in all, eo_shift)
        const uint32_t sl = uint32_t(atomicAdd(&a.bb[b], (1ull << 32) | e.cnt) >> 32);
        a.slot[i] = sl;
        if (sl == kEoSmall) *a.large = 1u;   // some bucket is beyond the one-thread-per-event kernel
    }
}

// every event to its bucket's slice
__global__ __launch_bounds__(256) void k_eo_scatter(EoArgs a) {
    uint64_t n;
    if (!eo_active(a, n)) return;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const PfEvent e = a.ev[i];
        a.tmp[a.offsets[eo_pos(a, e) >> a.shift] + a.slot[i]] = e;
    }
}

// Buckets of up to kEoSmall events (natural text against a dictionary: two or three per bucket): one thread per event
// ranks it against the bucket's events (its neighbours in tmp: cache hits) and writes its records.
// (Tried: one thread per BUCKET -- neighbouring threads on neighbouring slices of tmp and of the output.  48.5 us instead of
// 29.6 us per launch averaged over the bench's 78 order passes, profiles/r05_emit_per_bucket_negative.csv: a quarter of the
// threads, each with its bucket's events in series behind the own_pid gather; the scattered form hides that latency.)
__global__ __launch_bounds__(256) void k_eo_emit_small(EoArgs a, DfaEng eng, const uint32_t* __restrict__ hid2sid,
                                                       const uint32_t* __restrict__ own_pid, acgpu_match* __restrict__ out) {
    uint64_t n;
    if (!eo_active(a, n)) return;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const PfEvent e = a.ev[i];
        const uint64_t b = eo_pos(a, e) >> a.shift;
        const uint32_t m = uint32_t(a.bb[b] >> 32);
        if (m > kEoSmall) continue;
        const uint64_t base = a.offsets[b];
        uint32_t r = 0;
        for (uint32_t j = 0; j < m; j++) {
            const PfEvent o = a.tmp[base + j];
            if (o.key < e.key) r += o.cnt;
        }
        eo_write(eng, hid2sid, own_pid, e, out + base + r);
    }
}

// Larger buckets (match-saturated text: thousands of events per bucket), one wavefront per bucket: a second bucket
// level in LDS -- one bin per end position; scan; scatter; all-pairs inside each bin (the occurrences ending at one
// position: at most one per pattern length).  Runs only if k_eo_hist saw such a bucket.
__global__ __launch_bounds__(kEoBlock) void k_eo_emit_large(EoArgs a, DfaEng eng, const uint32_t* __restrict__ hid2sid,
                                                            const uint32_t* __restrict__ own_pid, acgpu_match* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bins[];
    uint64_t n;
    if (!eo_active(a, n) || *a.large == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* ecnt = s_bins + size_t(wave) * 3 * kEoBins;   // events per end position -> exclusive prefix
    uint32_t* rcnt = ecnt + kEoBins;                         // records per end position -> exclusive prefix
    uint32_t* fill = rcnt + kEoBins;
    const uint64_t wid = uint64_t(blockIdx.x) * kEoWaves + wave, nwaves = uint64_t(gridDim.x) * kEoWaves;
    for (uint64_t b = wid; b < a.n_buckets; b += nwaves) {
        const uint32_t m = uint32_t(a.bb[b] >> 32);
        if (m <= kEoSmall) continue;
        const uint64_t base = a.offsets[b];
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
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t te = uint32_t(__shfl_up(int(ei), o, 64)), tr = uint32_t(__shfl_up(int(ri), o, 64));
                if (lane >= o) { ei += te; ri += tr; }
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
            eo_write(eng, hid2sid, own_pid, e, out + base + r);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// enqueue-only form: the set was delivered after all -- totals[1] = 0 tells the caller so (include/acgpu.h).  A launch of
// its own behind the emit kernels: they all test totals[1] on entry.
__global__ void k_eo_done(EoArgs a) {
    uint64_t n;
    if (threadIdx.x == 0 && a.done_totals && eo_active(a, n)) a.done_totals[1] = 0;
}

struct Layout {
    size_t large, bb, offsets, active, aoff, bsum, bact, totals, slot, tmp, tmp2, total;
};
Layout layout(uint64_t max_events, uint64_t max_records, uint64_t nb) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    Layout L{};
    size_t o = 0;
    L.large = o; o += up(16);
    L.bb = o; o += up(nb * 8);
    L.offsets = o; o += up(nb * 8);
    L.active = o; o += up(nb * 8);
    L.aoff = o; o += up(nb * 8);
    L.bsum = o; o += up(((nb + 255) / 256) * 8);
    L.bact = o; o += up(((nb + 255) / 256) * 4);
    L.totals = o; o += up(2 * 8);
    L.slot = o; o += up(max_events * 4);
    L.tmp = o; o += up(max_records * sizeof(PfEvent));
    L.tmp2 = o; o += up(max_records * sizeof(PfEvent));
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

// The words a pass expects to be zero on entry (the flags and the bucket counters: the first bytes of `work`); a caller that
// has a kernel in flight anyway (k_ev_write) zeroes them there and passes zeroed = true.
size_t event_order_zero_bytes(uint64_t max_events, uint64_t max_records, uint64_t span_bytes) {
    const Layout L = layout(std::max<uint64_t>(max_events, 1), std::max<uint64_t>(max_records, 1),
                            buckets_of(span_bytes, eo_shift(max_events, max_records, span_bytes)));
    static_assert(sizeof(unsigned long long) == 8, "");
    return L.offsets;   // large | bb
}

size_t event_order_work_bytes(uint64_t max_events, uint64_t max_records, uint64_t span_bytes) {
    return layout(std::max<uint64_t>(max_events, 1), std::max<uint64_t>(max_records, 1),
                  buckets_of(span_bytes, eo_shift(max_events, max_records, span_bytes))).total;
}

hipError_t launch_event_order_emit(const HotTables& h, const DevAutomaton& a, const void* events, const uint64_t* totals,
                                   uint64_t min_events, uint64_t max_events, uint64_t max_records, uint64_t span_begin,
                                   uint64_t span_bytes, void* work, acgpu_match* out, hipStream_t s, uint64_t* done_totals,
                                   bool zeroed) {
    const uint32_t shift = eo_shift(max_events, max_records, span_bytes);
    const uint64_t nb = buckets_of(span_bytes, shift);
    const Layout L = layout(std::max<uint64_t>(max_events, 1), std::max<uint64_t>(max_records, 1), nb);
    uint8_t* w = static_cast<uint8_t*>(work);
    EoArgs ea{};
    ea.ev = static_cast<const PfEvent*>(events); ea.totals = totals; ea.min_events = min_events; ea.max_events = max_events;
    ea.max_records = max_records; ea.origin = span_begin; ea.n_buckets = nb; ea.shift = shift;
    ea.large = reinterpret_cast<uint32_t*>(w + L.large);
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
    const uint32_t bblocks = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>((nb + 255) / 256, uint64_t(device_cus()) * 16)));
    if (!zeroed) k_eo_zero<<<dim3(bblocks), dim3(256), 0, s>>>(ea);
    k_eo_hist<<<dim3(eblocks), dim3(256), 0, s>>>(ea);
    ScanScratch sc;   // exclusive prefix of the records per bucket (kernels.hip)
    sc.packed = reinterpret_cast<const uint64_t*>(ea.bb); sc.offsets = ea.offsets;
    sc.active = reinterpret_cast<uint64_t*>(w + L.active); sc.aoff = reinterpret_cast<uint64_t*>(w + L.aoff);
    sc.bsum = reinterpret_cast<uint64_t*>(w + L.bsum); sc.bact = reinterpret_cast<uint32_t*>(w + L.bact);
    sc.totals = reinterpret_cast<uint64_t*>(w + L.totals);
    if (hipError_t e = launch_scan(sc, nb, s); e != hipSuccess) return e;
    k_eo_scatter<<<dim3(eblocks), dim3(256), 0, s>>>(ea);
    k_eo_emit_small<<<dim3(eblocks), dim3(256), 0, s>>>(ea, eng, h.hid2sid, h.own_pid, out);
    k_eo_emit_large<<<dim3(uint32_t(std::min<int>(device_cus(), 1024))), dim3(kEoBlock), kEoLds, s>>>(ea, eng, h.hid2sid, h.own_pid, out);
    if (done_totals) k_eo_done<<<dim3(1), dim3(64), 0, s>>>(ea);
    return hipGetLastError();
}

}  // namespace acgpu
