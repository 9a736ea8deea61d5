// Ordered records from a LARGE set of level-3 events of the prefix filter (pf_scan.hip): match-dense inputs produce
// far more occurrences than the all-pairs rank of the direct mode can order (n^2), and re-walking every non-empty chunk
// (k_hot_fill / k_walk_fill) re-reads a large part of the haystack.  Here the events are ordered by a device radix sort
// of their keys (hipCUB / rocPRIM DeviceRadixSort over the used key bits), the record counts are prefix-summed in
// sorted order (DeviceScan) and every event scatters its records -- O(n) work, no second look at the haystack.
// The key order is the reference's stream order (see k_ev_rank in pf_scan.hip).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "hot.hpp"

namespace acgpu {

namespace {

__global__ __launch_bounds__(256) void k_es_split(const PfEvent* __restrict__ ev, uint64_t n, uint64_t* __restrict__ keys,
                                                  uint32_t* __restrict__ idx) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) { keys[i] = ev[i].key; idx[i] = uint32_t(i); }
}

__global__ __launch_bounds__(256) void k_es_counts(const PfEvent* __restrict__ ev, const uint32_t* __restrict__ idx,
                                                   uint64_t n, uint64_t* __restrict__ offs) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) offs[i] = ev[idx[i]].cnt;
}

__global__ __launch_bounds__(256) void k_es_emit(DfaEng eng, const uint32_t* __restrict__ hid2sid,
                                                 const PfEvent* __restrict__ ev, const uint32_t* __restrict__ idx,
                                                 const uint64_t* __restrict__ offs, uint64_t n,
                                                 acgpu_match* __restrict__ out) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const PfEvent e = ev[idx[i]];
    const uint32_t sid = hid2sid[e.node];
    const uint64_t end = e.key >> 16, len = 0xFFFFull - (e.key & 0xFFFFull);
    acgpu_match* dst = out + offs[i];
    for (uint32_t k = 0; k < e.cnt; k++) {
        acgpu_match m; m.pattern = eng.match_pattern(sid, k); m._pad = 0; m.start = end - len; m.end = end;
        dst[k] = m;
    }
}

struct Layout {
    size_t keys, keys2, idx, idx2, offs, offs2, temp, temp_bytes, total;
};
Layout layout(uint64_t n) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    Layout L{};
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, static_cast<const uint64_t*>(nullptr),
                                             static_cast<uint64_t*>(nullptr), static_cast<const uint32_t*>(nullptr),
                                             static_cast<uint32_t*>(nullptr), int(n));
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, static_cast<const uint64_t*>(nullptr),
                                           static_cast<uint64_t*>(nullptr), int(n));
    L.temp_bytes = std::max(sort_bytes, scan_bytes);
    size_t o = 0;
    L.keys = o; o += up(n * 8);
    L.keys2 = o; o += up(n * 8);
    L.idx = o; o += up(n * 4);
    L.idx2 = o; o += up(n * 4);
    L.offs = o; o += up(n * 8);
    L.offs2 = o; o += up(n * 8);
    L.temp = o; o += up(L.temp_bytes);
    L.total = o;
    return L;
}

}  // namespace

size_t event_sort_work_bytes(uint64_t n) { return layout(n ? n : 1).total; }

hipError_t launch_event_sort_emit(const HotTables& h, const DevAutomaton& a, const void* events, uint64_t n, uint64_t max_end,
                                  void* work, acgpu_match* out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (n > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const Layout L = layout(n);
    uint8_t* w = static_cast<uint8_t*>(work);
    uint64_t* keys = reinterpret_cast<uint64_t*>(w + L.keys);
    uint64_t* keys2 = reinterpret_cast<uint64_t*>(w + L.keys2);
    uint32_t* idx = reinterpret_cast<uint32_t*>(w + L.idx);
    uint32_t* idx2 = reinterpret_cast<uint32_t*>(w + L.idx2);
    uint64_t* offs = reinterpret_cast<uint64_t*>(w + L.offs);
    uint64_t* offs2 = reinterpret_cast<uint64_t*>(w + L.offs2);
    const PfEvent* ev = static_cast<const PfEvent*>(events);
    const uint32_t blocks = uint32_t((n + 255) / 256);
    k_es_split<<<dim3(blocks), dim3(256), 0, s>>>(ev, n, keys, idx);
    int key_bits = 16;   // key = end << 16 | (0xFFFF - len): only the bits an end offset <= max_end can set are sorted
    while (key_bits < 64 && (max_end >> (key_bits - 16)) != 0) key_bits++;
    size_t tb = L.temp_bytes;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(w + L.temp, tb, keys, keys2, idx, idx2, int(n), 0, key_bits, s);
    if (e != hipSuccess) return e;
    k_es_counts<<<dim3(blocks), dim3(256), 0, s>>>(ev, idx2, n, offs);
    tb = L.temp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(w + L.temp, tb, offs, offs2, int(n), s);
    if (e != hipSuccess) return e;
    DfaEng eng; eng.d = a.dfa; eng.cls = a.dfa.classes;
    k_es_emit<<<dim3(blocks), dim3(256), 0, s>>>(eng, h.hid2sid, ev, idx2, offs2, n, out);
    return hipGetLastError();
}

}  // namespace acgpu
