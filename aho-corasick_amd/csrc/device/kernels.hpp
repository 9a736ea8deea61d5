// Host-callable launch wrappers around the HIP kernels (kernels.hip, hot_scan.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acgpu.h"
#include "engines.hpp"

namespace acgpu {

enum EngineId : uint32_t { ENG_DFA = 1, ENG_CNFA = 2, ENG_HOT = 3, ENG_PF = 4 };

struct DevAutomaton {
    bool has_dfa = false, has_cnfa = false, has_hot = false;
    DevDfa dfa{};
    DevCnfa cnfa{};
};

// Scratch for one chunked overlapping scan.
struct ScanScratch {
    uint32_t* counts = nullptr;    // [n_chunks]
    const uint64_t* packed = nullptr;   // instead of `counts`: the counts are the LOW words of 64-bit entries (event_order.hip: records | events << 32 per bucket)
    uint64_t* offsets = nullptr;   // [n_chunks]   exclusive prefix of counts; nullptr = not needed (only `aoff` is written)
    uint64_t* active = nullptr;    // [n_chunks]   ids of chunks with count > 0, ascending
    uint64_t* aoff = nullptr;      // [n_chunks]   exclusive prefix of counts, per ACTIVE chunk (parallel to `active`)
    uint64_t* bsum = nullptr;      // [n_blocks]   per-256-chunk block sums -> exclusive prefix
    uint32_t* bact = nullptr;      // [n_blocks]   per-block active counts -> exclusive prefix
    uint64_t* totals = nullptr;    // [2] total matches, total active chunks (device)
    uint64_t cap_chunks = 0;
    // page-locked, device-visible landing zone (or nullptr): the scan's second kernel also stores the totals there -- and the
    // 32-bit word at `extra32`, if given, as host_totals[2] -- which saves the copy launches between the scan and the host
    uint64_t* host_totals = nullptr;
    const uint32_t* extra32 = nullptr;
};

// generic engines (reference-faithful per-byte walk), kernels.hip
hipError_t launch_walk_count(uint32_t engine, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts, hipStream_t s);
hipError_t launch_walk_fill(uint32_t engine, const DevAutomaton& a, const ScanGeom& g, const uint64_t* active,
                            const uint64_t* totals, uint64_t cap, uint64_t max_blocks, const uint64_t* aoff,
                            acgpu_match* out, hipStream_t s, const unsigned long long* gate = nullptr);
hipError_t launch_scan(const ScanScratch& sc, uint64_t n_chunks, hipStream_t s);

struct SerialArgs {
    const uint8_t* hay;
    uint64_t span_start, span_end;
    int32_t anchored, earliest, match_kind;
    acgpu_match* out;      // device
    uint64_t cap;
    uint64_t* n_out;       // device: total number of matches
};
hipError_t launch_find_iter_serial(uint32_t engine, const DevAutomaton& a, const SerialArgs& args, hipStream_t s);
hipError_t launch_find_serial(uint32_t engine, const DevAutomaton& a, const SerialArgs& args, hipStream_t s);

// parallel selection (select.hip)
size_t select_scratch_bytes(uint64_t m);
// streams of up to this many occurrences are selected without the scan launches; `n_in` may then be sc.totals itself
// (nothing reads the stream length after the count of selected records is written there)
constexpr uint64_t kSelectFewLimit = uint64_t(1024) * 1024;
// The stream was produced by an enqueue-only search sized by a guess (capi_find.cpp): `totals` = {records, events} of that
// search on the device; the selection kernels see an EMPTY stream unless it was delivered -- events <= max_events (an
// abandoned scan reports UINT64_MAX) and records <= max_records.  host (page-locked, device-visible): receives {records,
// events, selected} from the last kernel, which saves the copy launch.
struct SelectGate {
    const uint64_t* totals = nullptr;
    uint64_t max_events = 0, max_records = 0;
    uint64_t* host = nullptr;
};
hipError_t launch_select_parallel(const acgpu_match* S, uint64_t m, const uint64_t* n_in, int match_kind,
                                  uint64_t span_start, uint64_t L, void* work, const ScanScratch& sc, acgpu_match* out,
                                  uint64_t cap, hipStream_t s, SelectGate gate = SelectGate());
hipError_t launch_select_nonoverlapping(const acgpu_match* S, const uint64_t* n_in, int match_kind, uint64_t span_start,
                                        uint64_t L, acgpu_match* out, uint64_t cap, uint64_t* n_out, hipStream_t s);

// replace_all (replace.hip)
size_t replace_scratch_bytes(uint64_t m);
hipError_t launch_replace_measure(const acgpu_match* M, uint64_t m, const uint8_t* hay, uint64_t hay_len,
                                  const uint64_t* roff, bool utf8, void* work, uint64_t* total_out, hipStream_t s);
hipError_t launch_replace_copy(const acgpu_match* M, uint64_t m, const uint8_t* hay, uint64_t hay_len,
                               const uint8_t* rbytes, const uint64_t* roff, const void* work, const uint64_t* total_out,
                               uint8_t* out, uint64_t out_len, hipStream_t s);

hipError_t launch_offset_records(acgpu_match* m, uint64_t n, uint64_t off, hipStream_t s);
hipError_t launch_stream_read(const uint8_t* src, size_t len, unsigned* sink, hipStream_t s);
hipError_t launch_gen_haystack(uint8_t* dst, uint64_t offset, size_t len, uint64_t seed, uint32_t lo, uint32_t span,
                               hipStream_t s);

}  // namespace acgpu
