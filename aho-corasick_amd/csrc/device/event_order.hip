// Ordered records from a LARGE set of level-3 events of the prefix filters (pf_scan.hip / pfx_scan.hip): match-dense
// inputs produce far more occurrences than the all-pairs rank of the direct mode can order (n^2), and re-walking every
// non-empty chunk (k_hot_fill / k_walk_fill) re-reads a large part of the haystack.  The events arrive unordered (the
// wavefronts append them as they go), but every event carries its end position, and the stream order is "ascending end,
// then the longer occurrence first" = ascending key (k_ev_rank in pf_scan.hip).  So the order is a bucket pass, O(n):
//
//   k_eo_hist     one thread per event: bucket = 2 KiB of end positions; events and records per bucket (global atomics
//                 spread over the buckets), the event's arrival slot in its bucket
//   launch_scan   exclusive prefix of the records per bucket = the bucket's slice of the output AND of the scratch
//                 array (every event stands for >= 1 record: slices of records are large enough for the events),
//                 list of the non-empty buckets                                    (the scan kernels of kernels.hip)
//   k_eo_scatter  one thread per event: to its bucket's slice
//   k_eo_emit     one wavefront per non-empty bucket: up to 64 events are ranked by a register all-pairs (shuffles);
//                 more (match-saturated text: thousands per bucket) by a second bucket level in LDS -- one bin per end
//                 position, scan, scatter, and an all-pairs inside each bin (the occurrences ending at one position:
//                 at most one per pattern length) -- then every event writes its records.
//
// Hand-written throughout (round 2 used hipCUB's radix sort + scan here: ~15 library launches, 0.33 ms of a 1.8 ms
// natural-text step).  Every kernel reads the event / record counts from device memory and does nothing when the set
// is small enough for the all-pairs rank, too large for its buffers, or the scan was abandoned -- so the enqueue-only
// form queues them unconditionally behind the scan (no host decision).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hot.hpp"
#include "launch_util.hpp"

namespace acgpu {

namespace {

constexpr uint32_t kEoShift = 11;                 // bucket = 2 KiB of end positions
constexpr uint32_t kEoBins = 1u << kEoShift;
constexpr int kEoWaves = 2;                       // wavefronts (= buckets in flight) per workgroup of k_eo_emit: 48 KiB of LDS bins

struct EoArgs {
    const PfEvent* ev;
    const uint64_t* totals;      // [0] records, [1] events of this scan (written by k_ev_rank)
    uint64_t min_events;         // sets up to this size were ordered by the all-pairs rank already
    uint64_t max_events;         // capacity of `ev`
    uint64_t max_records;        // capacity of tmp / tmp2 / the output
    uint64_t origin;             // end position - 1 - origin = offset into the bucket grid (origin = shard begin)
    uint64_t n_buckets;
    uint32_t* bcnt;              // [n_buckets] events per bucket   (zeroed)
    uint32_t* brec;              // [n_buckets] records per bucket  (zeroed)
    uint32_t* slot;              // [max_events] arrival slot of the event in its bucket
    PfEvent* tmp;                // [max_records] events grouped by bucket
    PfEvent* tmp2;               // [max_records] ... and by end position inside large buckets
};

__device__ __forceinline__ bool eo_active(const EoArgs& a, uint64_t& n) {
    n = a.totals[1];
    return n > a.min_events && n <= a.max_events && a.totals[0] <= a.max_records;
}
__device__ __forceinline__ uint64_t eo_pos(const EoArgs& a, const PfEvent& e) { return (e.key >> 16) - 1 - a.origin; }

__global__ __launch_bounds__(256) void k_eo_hist(EoArgs a) {
    uint64_t n;
    if (!eo_active(a, n)) return;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const PfEvent e = a.ev[i];
        const uint64_t b = eo_pos(a, e) >> kEoShift;
        a.slot[i] = atomicAdd(&a.bcnt[b], 1u);
        atomicAdd(&a.brec[b], e.cnt);
    }
}

__global__ __launch_bounds__(256) void k_eo_scatter(EoArgs a, const uint64_t* __restrict__ offsets) {
    uint64_t n;
    if (!eo_active(a, n)) return;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const PfEvent e = a.ev[i];
        const uint64_t b = eo_pos(a, e) >> kEoShift;
        a.tmp[offsets[b] + a.slot[i]] = e;
    }
}

__device__ __forceinline__ void eo_write(const DfaEng& eng, const uint32_t* __restrict__ hid2sid, const PfEvent& e,
                                         acgpu_match* __restrict__ dst) {
    const uint32_t sid = hid2sid[e.node];
    const uint64_t end = e.key >> 16, len = 0xFFFFull - (e.key & 0xFFFFull);
    for (uint32_t k = 0; k < e.cnt; k++) {
        acgpu_match m; m.pattern = eng.match_pattern(sid, k); m._pad = 0; m.start = end - len; m.end = end;
        dst[k] = m;
    }
}

// One wavefront per non-empty bucket (grid-stride over the list of the scan).  `bucket_totals` = {records, non-empty
// buckets} of the bucket scan.
__global__ __launch_bounds__(64 * kEoWaves) void k_eo_emit(EoArgs a, DfaEng eng, const uint32_t* __restrict__ hid2sid,
                                                 const uint64_t* __restrict__ offsets, const uint64_t* __restrict__ active,
                                                 const uint64_t* __restrict__ bucket_totals, acgpu_match* __restrict__ out) {
    __shared__ uint32_t s_ecnt[kEoWaves][kEoBins];   // events per end position -> exclusive prefix
    __shared__ uint32_t s_rcnt[kEoWaves][kEoBins];   // records per end position -> exclusive prefix
    __shared__ uint32_t s_fill[kEoWaves][kEoBins];
    uint64_t n;
    if (!eo_active(a, n)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t n_active = bucket_totals[1];
    for (uint64_t ai = uint64_t(blockIdx.x) * kEoWaves + wave; ai < n_active; ai += uint64_t(gridDim.x) * kEoWaves) {
        const uint64_t b = active[ai];
        const uint32_t m = a.bcnt[b];
        const uint64_t base = offsets[b];
        if (m <= 64) {   // rank = records of the events with a smaller key, all-pairs through shuffles
            PfEvent e{~0ull, 0, 0};
            if (uint32_t(lane) < m) e = a.tmp[base + lane];
            uint32_t r = 0;
            for (uint32_t j = 0; j < m; j++) {
                const uint32_t klo = uint32_t(__shfl(int(uint32_t(e.key)), int(j), 64)), khi = uint32_t(__shfl(int(uint32_t(e.key >> 32)), int(j), 64));
                const uint32_t c = uint32_t(__shfl(int(e.cnt), int(j), 64));
                if (((uint64_t(khi) << 32) | klo) < e.key) r += c;
            }
            if (uint32_t(lane) < m) eo_write(eng, hid2sid, e, out + base + r);
            continue;
        }
        // a second bucket level: one bin per end position of the bucket
        uint32_t* ecnt = s_ecnt[wave];
        uint32_t* rcnt = s_rcnt[wave];
        uint32_t* fill = s_fill[wave];
        for (uint32_t i = lane; i < kEoBins; i += 64) { ecnt[i] = 0; rcnt[i] = 0; fill[i] = 0; }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < m; i += 64) {
            const PfEvent e = a.tmp[base + i];
            const uint32_t eo = uint32_t(eo_pos(a, e)) & (kEoBins - 1);
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
            const uint32_t eo = uint32_t(eo_pos(a, e)) & (kEoBins - 1);
            a.tmp2[base + ecnt[eo] + atomicAdd(&fill[eo], 1u)] = e;
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < m; i += 64) {
            // (volatile: written a moment ago by other lanes of this wavefront -- past a possibly stale L1 line)
            const volatile PfEvent* t2 = a.tmp2 + base;
            PfEvent e; e.key = t2[i].key; e.node = t2[i].node; e.cnt = t2[i].cnt;
            const uint32_t eo = uint32_t(eo_pos(a, e)) & (kEoBins - 1);
            const uint32_t g0 = ecnt[eo], g1 = g0 + fill[eo];
            uint32_t r = rcnt[eo];
            for (uint32_t j = g0; j < g1; j++)
                if (t2[j].key < e.key) r += t2[j].cnt;
            eo_write(eng, hid2sid, e, out + base + r);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// enqueue-only form: the set was delivered after all -- totals[1] = 0 tells the caller so (include/acgpu.h)
__global__ void k_eo_done(EoArgs a, uint64_t* __restrict__ totals) {
    uint64_t n;
    if (threadIdx.x == 0 && eo_active(a, n)) totals[1] = 0;
}

struct Layout {
    size_t bcnt, brec, slot, tmp, tmp2, offsets, active, aoff, bsum, bact, totals, total;
};
Layout layout(uint64_t max_events, uint64_t max_records, uint64_t nb) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    Layout L{};
    size_t o = 0;
    L.bcnt = o; o += up(nb * 4);
    L.brec = o; o += up(nb * 4);
    L.slot = o; o += up(max_events * 4);
    L.tmp = o; o += up(max_records * sizeof(PfEvent));
    L.tmp2 = o; o += up(max_records * sizeof(PfEvent));
    L.offsets = o; o += up(nb * 8);
    L.active = o; o += up(nb * 8);
    L.aoff = o; o += up(nb * 8);
    L.bsum = o; o += up(((nb + 255) / 256) * 8);
    L.bact = o; o += up(((nb + 255) / 256) * 4);
    L.totals = o; o += up(2 * 8);
    L.total = o;
    return L;
}
uint64_t buckets_of(uint64_t span_bytes) { return std::max<uint64_t>(1, (span_bytes + kEoBins - 1) >> kEoShift) + 1; }

}  // namespace

size_t event_order_work_bytes(uint64_t max_events, uint64_t max_records, uint64_t span_bytes) {
    return layout(std::max<uint64_t>(max_events, 1), std::max<uint64_t>(max_records, 1), buckets_of(span_bytes)).total;
}

hipError_t launch_event_order_emit(const HotTables& h, const DevAutomaton& a, const void* events, const uint64_t* totals,
                                   uint64_t min_events, uint64_t max_events, uint64_t max_records, uint64_t span_begin,
                                   uint64_t span_bytes, void* work, acgpu_match* out, hipStream_t s, uint64_t* done_totals) {
    const uint64_t nb = buckets_of(span_bytes);
    const Layout L = layout(std::max<uint64_t>(max_events, 1), std::max<uint64_t>(max_records, 1), nb);
    uint8_t* w = static_cast<uint8_t*>(work);
    EoArgs ea{};
    ea.ev = static_cast<const PfEvent*>(events); ea.totals = totals; ea.min_events = min_events; ea.max_events = max_events;
    ea.max_records = max_records; ea.origin = span_begin; ea.n_buckets = nb;
    ea.bcnt = reinterpret_cast<uint32_t*>(w + L.bcnt); ea.brec = reinterpret_cast<uint32_t*>(w + L.brec);
    ea.slot = reinterpret_cast<uint32_t*>(w + L.slot);
    ea.tmp = reinterpret_cast<PfEvent*>(w + L.tmp); ea.tmp2 = reinterpret_cast<PfEvent*>(w + L.tmp2);
    hipError_t e = hipMemsetAsync(w + L.bcnt, 0, L.slot - L.bcnt, s);   // bcnt + brec
    if (e != hipSuccess) return e;
    const uint32_t blocks = uint32_t(std::min<uint64_t>((max_events + 255) / 256, uint64_t(device_cus()) * 16));
    k_eo_hist<<<dim3(std::max(blocks, 1u)), dim3(256), 0, s>>>(ea);
    ScanScratch sc;
    sc.counts = ea.brec; sc.offsets = reinterpret_cast<uint64_t*>(w + L.offsets);
    sc.active = reinterpret_cast<uint64_t*>(w + L.active); sc.aoff = reinterpret_cast<uint64_t*>(w + L.aoff);
    sc.bsum = reinterpret_cast<uint64_t*>(w + L.bsum); sc.bact = reinterpret_cast<uint32_t*>(w + L.bact);
    sc.totals = reinterpret_cast<uint64_t*>(w + L.totals);
    if ((e = launch_scan(sc, nb, s)) != hipSuccess) return e;
    k_eo_scatter<<<dim3(std::max(blocks, 1u)), dim3(256), 0, s>>>(ea, sc.offsets);
    DfaEng eng; eng.d = a.dfa; eng.cls = a.dfa.classes;
    const uint32_t eblocks = uint32_t(std::min<uint64_t>((nb + kEoWaves - 1) / kEoWaves, uint64_t(device_cus()) * 24));
    k_eo_emit<<<dim3(std::max(eblocks, 1u)), dim3(64 * kEoWaves), 0, s>>>(ea, eng, h.hid2sid, sc.offsets, sc.active, sc.totals, out);
    if (done_totals) k_eo_done<<<dim3(1), dim3(64), 0, s>>>(ea, done_totals);
    return hipGetLastError();
}

}  // namespace acgpu
