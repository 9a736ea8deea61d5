// Internals of the C ABI implementation shared by its translation units (capi.cpp: build / upload / getters; capi_overlap.cpp: the overlapping search; capi_enqueue.cpp: its enqueue-only form;
// capi_find.cpp: find_iter, find, is_match; capi_stream.cpp: replace_all and the stream search): per-device state,
// scratch, and the entry points they call in each other.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>
#include "acgpu.h"
#include "capi_internal.hpp"
#include "device/cnfa_walk.hpp"
#include "device/cnfa_tri.hpp"
#include "device/cnfa_tri_step.hpp"
#include "device/dfa_tri.hpp"
#include "device/dfa_fill.hpp"
#include "device/hot.hpp"
#include "device/kernels.hpp"
#include "device/merge.hpp"
#include "device/select.hpp"
#include "device/start_select.hpp"
#include "host/automaton.hpp"
#include "host/cnfa_tables.hpp"
#include "host/cnfa_tri_tables.hpp"
#include "host/devbuf.hpp"
#include "host/engine_plan.hpp"
#include "host/lw_tables.hpp"
#include "host/pf_tables.hpp"

namespace acgpu_capi {

using namespace acgpu;


extern thread_local std::string g_last_error;
// Density rule of an internal-mode (dev_result) overlapping call, passed down explicitly by the caller that owns the
// alternative (find_iter / find / replace_all / the stream search).  When the occurrence stream is too dense to be worth
// materialising the call returns ACGPU_ERR_NOMEM with `hit` set, and the caller takes its alternative: the per-start
// table (start_select.hip), or the reference loop on one lane (~30 ns per haystack byte whatever the number of
// occurrences).  div != 0 (a leftmost find_iter that CAN select from the per-start table): already "more than one
// occurrence per `div` haystack bytes" counts as dense -- that path costs the same whatever the density, the occurrence
// stream 24 bytes per occurrence plus its ordering and selection.  guard = false (the stream search, which has no
// alternative): never dense.
struct DenseRule {
    uint32_t div = 0;
    bool guard = true;
    // hard bound (overlapping_split: the records its caller has room for): a stream longer than this is counted, not
    // materialised -- the call returns ACGPU_ERR_BUFFER_TOO_SMALL with the count
    uint64_t max_records = UINT64_MAX;
    bool hit = false;   // out
    bool too_dense(uint64_t records, uint64_t span_bytes) const {
        if (!guard) return false;
        if (div) return records > std::max<uint64_t>(uint64_t(1) << 16, span_bytes / div);
        return records > std::max<uint64_t>(uint64_t(1) << 24, 32 * span_bytes);
    }
};

acgpu_status hip_fail(hipError_t e, const char* what);
#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);     \
    } while (0)

// One scan's worth of scratch; pooled per device so concurrent searches do not share state.
struct Scratch {
    DevBuf counts, offsets, active, aoff, bsum, bact, totals, result, hay, sel, selwork, seltot;
    DevBuf rhay, rmatch, rtab, roff, rwork, rout;  // replace_all
    DevBuf events, evrank, evctr, eswork;          // prefix-filter direct / sorted-events modes (level-3 events -> ordered records)
    DevBuf hitwork;                                // large-set filter: global hit list of its second-pass level 3
    DevBuf triev, triseg, trictr;                  // contiguous-NFA walk: match events of the count pass (cnfa_tri.hip)
    DevBuf lwev, lwtn, lwovf;                      // LDS walk: match events of the count walk by task, their counts, the overflow word (lds_emit.hip)
    uint32_t lw_gen = 0;                           // ... generation of the overflow word (a new value per call; the word is zeroed when made)
    DevBuf probe;                                  // prefix-filter probe: 8 counters + the decision word at byte 64 (zeroed once)
    bool probe_ready = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_armed = false;        // evrank[] == 0 and evctr[] == 0 (the invariant k_ev_write restores; false after a failed call)
    // fused order chain (event_order.hip): the first eo_zero_bytes of eswork at eo_zero_p are zero (or will be when the stream
    // gets there: the chain re-zeroes them BEHIND its last kernel); nullptr after any other use of eswork or a failed call
    void* eo_zero_p = nullptr;
    size_t eo_zero_bytes = 0;
    bool events_served = false;   // the last internal-mode search on this scratch took its records from the prefix filter's events
    bool rank_over = false;       // ... which had more events than the all-pairs rank takes
    uint32_t rank_hint = 0;       // events of the previous event-mode call on this scratch: sizes the all-pairs grid only
    uint64_t* pinned = nullptr;   // [8] page-locked landing zone for the totals (a pageable target makes the copy a staged, blocking one)
    hipError_t ensure_pinned() {
        if (pinned) return hipSuccess;
        const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&pinned), 8 * sizeof(uint64_t));
        if (e == hipSuccess) std::memset(pinned, 0, 8 * sizeof(uint64_t));
        return e;
    }
    uint64_t fin_seq = 0;         // fused order chain: the word its last kernel stores at pinned[3] (a new value per call)
    ~Scratch() {
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (pinned) (void)hipHostFree(pinned);
    }
};

// An adaptive hint of a DeviceState: a small counter the searches of one automaton leave for each other ("recent scans were
// abandoned", "results were dense lately").  With acgpu_config.deterministic_routing the hints read 0 and ignore writes:
// the engine choice of a call then follows from the automaton and the span alone.
struct Hint {
    std::atomic<int> v{0};
    const bool* on = nullptr;
    bool live() const { return on && *on; }
    int load(std::memory_order = std::memory_order_relaxed) const { return live() ? v.load(std::memory_order_relaxed) : 0; }
    void store(int x, std::memory_order = std::memory_order_relaxed) { if (live()) v.store(x, std::memory_order_relaxed); }
    int fetch_add(int d, std::memory_order = std::memory_order_relaxed) { return live() ? v.fetch_add(d, std::memory_order_relaxed) : 0; }
    int fetch_sub(int d, std::memory_order = std::memory_order_relaxed) { return live() ? v.fetch_sub(d, std::memory_order_relaxed) : 0; }
};

// An enqueue-only search queued by a SYNCHRONOUS caller that has a guess of the result size (find_iter over an occurrence
// stream like the last one's, capi_find.cpp): the bucket order pass is queued whatever the density hints say, sized by the
// guess instead of by the capacities.
struct EnqueueGuess {
    uint64_t max_events = 0;    // the order pass serves up to this many events ...
    bool over_all_pairs = false;   // ... and the last stream had more than the all-pairs rank takes (grid size of that kernel only)
    // out: the bound the order pass was queued with.  totals[1] is NOT reset by the pass in this form (one launch less): the
    // records are there iff totals[1] <= served_events and totals[0] <= cap
    uint64_t served_events = 0;
    // out: the fused order chain was queued (event_order.hip): totals = {records, 0 = delivered | UINT64_MAX}, and the caller
    // launches launch_event_order_zero(rearm_p, rearm_bytes) BEHIND its own kernels (then marks the scratch: Scratch::eo_zero_p)
    void* rearm_p = nullptr;
    size_t rearm_bytes = 0;
};

// A synchronous caller of enqueue_impl (overlapping_impl borrowing the stream's context): where the totals may be reported
// in page-locked host memory, and what to wait for.
struct EnqueueSync {
    uint64_t* host_totals = nullptr;   // in: [4] page-locked, device-visible
    uint64_t seq = 0;                  // in: stored at host_totals[3] behind the three words below
    hipEvent_t done = nullptr;         // out: != nullptr: host_totals = {records, 0 = delivered | UINT64_MAX, events} once this event has passed
                                       //      (or host_totals[3] == seq, whichever the caller sees first)
};

struct DeviceState {
    int device = -1;
    bool adaptive = true;   // !acgpu_config.deterministic_routing
    Variants var;           // the automaton's engine variants at upload (host/variants.hpp)
    DeviceState() { for (Hint* h : {&route_hint, &probe_away_run, &probe_skip, &dense_hint, &ss_hint, &walk_hint, &stream_hint, &stream_cool, &lw_dense_hint}) h->on = &adaptive; }
    DevAutomaton da;
    DevBuf dfa_trans, dfa_moff, dfa_mpid, dfa_cls, cnfa_repr, cnfa_cls, plens;
    HotTables hot;   // LDS-resident fast path (hot_scan.hip), optional
    CnfaHotTables cnfa_hot;   // contiguous-NFA walk with the start state's neighbourhood in LDS (cnfa_walk.hip)
    CnfaTriTables cnfa_tri;   // contiguous-NFA walk that skips the depth <= 2 regime by a trigram bitmap in LDS (cnfa_tri.hip)
    DfaTriTables dfa_tri;     // the same skip in front of the DFA transition walk (dfa_tri.hip)
    bool derived_dfa = false;  // da.dfa was derived from an NFA-kind automaton at upload (device only)
    // > 0 while recent scans of this automaton were abandoned by the two-type filter (PfArgs::route_*): the next scans
    // ask the probe (launch_pf_probe, ~10 us) which engine to run instead of paying for an abandoned pass each; every
    // probe that finds the filter adequate counts it down, so a caller with harmless input stops paying for probes
    Hint route_hint;
    // consecutive probes that sent the scan to the large-set filter, and the searches that then skip the probe altogether
    // (natural text against a dictionary, call after call: the probe and its host round trip were ~50 us of a 0.85 ms call)
    Hint probe_away_run, probe_skip;
    // > 0 while recent searches of this automaton returned more occurrences than the all-pairs rank orders: the next
    // enqueue-only calls queue the bucket order pass (event_order.hip: nine small launches) behind their scan, so dense
    // results are delivered without a host decision; callers with sparse results never pay for those launches
    Hint dense_hint;
    // > 0 while recent leftmost find_iter calls of this automaton met occurrence-dense input: the next ones go straight to
    // the per-start table (start_select.hip) instead of counting the occurrence stream first
    Hint ss_hint;
    // > 0 while recent scans by the large-set filter recorded more occurrences than its event list holds (a dictionary whose
    // words are everywhere in the text): the next searches go straight to the transition walk, whose count pass does not
    // pay per occurrence
    Hint walk_hint;
    // length of the last occurrence stream a parallel find_iter of this automaton took from the prefix filter's events (0: none,
    // or too long to guess at): the next one queues scan, order pass and selection sized by twice that and synchronises ONCE;
    // stream_cool > 0 after a guess that did not hold (that search was repeated the regular way): no guessing for a while
    Hint stream_hint, stream_cool;
    // > 0 while recent searches by the LDS walk's event form overflowed their slabs (a record every few bytes): the next
    // device-output calls queue the chunk fill, gated on the overflow word, behind the events -- no host round trip in between
    Hint lw_dense_hint;
    std::mutex pool_mu;
    std::vector<std::unique_ptr<Scratch>> pool;
    // enqueue-only calls: one scratch per stream, never pooled (work of earlier calls may still be in flight on it;
    // stream order makes the reuse by the next call on the same stream safe)
    struct AsyncCtx {
        std::mutex busy;   // held by a SYNCHRONOUS call that borrows this context (overlapping_impl); enqueue-only callers follow the one-thread-per-stream rule of acgpu.h
        Scratch sc;
        hipEvent_t ev[130] = {};   // slots 0..63 are the caller's (acgpu_enqueue_kernel_ms); 64 the library's own
        hipEvent_t fin = nullptr;  // fused order chain: recorded behind the kernel that reports the totals (EnqueueSync)
        ~AsyncCtx() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); if (fin) (void)hipEventDestroy(fin); }
    };
    std::mutex async_mu;
    std::map<hipStream_t, std::unique_ptr<AsyncCtx>> async;
    AsyncCtx* async_ctx(hipStream_t s) {
        std::lock_guard<std::mutex> lk(async_mu);
        auto& p = async[s];
        if (!p) p = std::make_unique<AsyncCtx>();
        return p.get();
    }

    std::unique_ptr<Scratch> take() {
        std::lock_guard<std::mutex> lk(pool_mu);
        if (!pool.empty()) { auto s = std::move(pool.back()); pool.pop_back(); return s; }
        return std::make_unique<Scratch>();
    }
    void give(std::unique_ptr<Scratch> s) {
        std::lock_guard<std::mutex> lk(pool_mu);
        if (pool.size() < 4) pool.push_back(std::move(s));
    }
};


struct ScratchLease {
    DeviceState* ds;
    std::unique_ptr<Scratch> s;
    ScratchLease(DeviceState* d) : ds(d), s(d->take()) {}
    ~ScratchLease() { ds->give(std::move(s)); }
    Scratch* operator->() { return s.get(); }
};

// ---- host haystacks: copy / scan overlap ---------------------------------------------------------------------------
// A helper thread copies the haystack to the device piece by piece on its own stream (a hipMemcpyAsync from pageable
// memory returns only when the runtime has staged the source, so it has to be a thread, not just a second stream) and
// records one event per piece; the caller waits for piece k (condition variable, then hipStreamWaitEvent on its compute
// stream) and scans it while pieces k+1.. are still crossing PCIe.  Mirrors the roll buffer of the reference's stream
// searcher (src/util/buffer.rs:113-123: keep min_buffer_len bytes, refill behind the search), with the search side on
// all CUs: only the last piece's scan is not hidden by a copy.
struct HostPipe {
    int device = 0;
    uint8_t* dst = nullptr;
    const uint8_t* src = nullptr;
    size_t len = 0, piece = 0, n_pieces = 0;
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> ev;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    size_t submitted = 0;          // pieces whose copy is enqueued and whose event is recorded
    hipError_t err = hipSuccess;

    hipError_t start(int dev, uint8_t* d, const uint8_t* s, size_t n, size_t piece_bytes) {
        device = dev; dst = d; src = s; len = n; piece = piece_bytes;
        n_pieces = (n + piece - 1) / piece;
        hipError_t e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        ev.assign(n_pieces, nullptr);
        for (auto& x : ev) if ((e = hipEventCreateWithFlags(&x, hipEventDisableTiming)) != hipSuccess) return e;
        th = std::thread([this] {
            hipError_t e2 = hipSetDevice(device);
            for (size_t k = 0; k < n_pieces; k++) {
                const size_t off = k * piece, nb = std::min(piece, len - off);
                if (e2 == hipSuccess) e2 = hipMemcpyAsync(dst + off, src + off, nb, hipMemcpyHostToDevice, copy_stream);
                if (e2 == hipSuccess) e2 = hipEventRecord(ev[k], copy_stream);
                std::lock_guard<std::mutex> lk(mu);
                if (e2 != hipSuccess && err == hipSuccess) err = e2;
                submitted = k + 1;
                cv.notify_all();
            }
        });
        return hipSuccess;
    }
    // blocks until piece k's copy has been enqueued, then orders `compute` behind it
    hipError_t wait(size_t k, hipStream_t compute) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return submitted > k; });
        if (err != hipSuccess) return err;
        lk.unlock();
        return hipStreamWaitEvent(compute, ev[k], 0);
    }
    ~HostPipe() {
        if (th.joinable()) th.join();
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
        for (auto& x : ev) if (x) (void)hipEventDestroy(x);
    }
};

}  // namespace acgpu_capi

// Stream search state (src/automaton.rs:1036-1244): what StreamChunkIter carries between reads.
struct acgpu_stream {
    acgpu_automaton* aut = nullptr;
    acgpu::DevBuf buf;              // [halo | chunk] on the device
    acgpu::DevBuf stage;            // device copy of a large host feed, filled piece by piece under the search (HostPipe)
    std::vector<uint8_t> halo;      // last max_pattern_len-1 bytes of the stream so far
    std::vector<acgpu_match> last;  // matches completed by the most recent feed (absolute offsets)
    uint64_t total = 0;             // bytes consumed so far
    uint64_t pos = 0;               // end of the last reported match
};

namespace acgpu_capi {

// ---- capi.cpp
acgpu_status get_device_state(acgpu_automaton* aut, DeviceState** out);
acgpu_status enforce_anchored_consistency(int have, bool want_anchored);
acgpu_status check_input(const acgpu_input* in);
uint32_t generic_engine(const acgpu_automaton* aut, const DeviceState* ds);
acgpu_status check_start(const acgpu_automaton* aut, bool anchored);
acgpu_status device_haystack(const acgpu_input* in, size_t lo, size_t hi, Scratch* sc, hipStream_t stream, const uint8_t** out);
acgpu_status ensure_events(Scratch* sc);
size_t host_piece_bytes();
// `ext` / `dev_result`: internal mode used by the parallel find_iter -- run on the caller's scratch and leave the
// ordered records in scratch->result (returned through *dev_result) instead of copying them anywhere.
acgpu_status overlapping_impl(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end,
                              acgpu_match* out, size_t cap, size_t* n_out, acgpu_profile* prof,
                              Scratch* ext = nullptr, acgpu_match** dev_result = nullptr, DenseRule* dense = nullptr);
uint32_t default_chunk(const acgpu_automaton* aut, size_t span_len);
// scan geometry of one shard: 64-byte aligned base, ownership window, chunk grid
ScanGeom make_geom(const acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end, const uint8_t* dhay, size_t halo);
// ---- capi_overlap.cpp (shared with the enqueue-only form, capi_enqueue.cpp)
constexpr uint32_t ENG_PF_LARGE = 100;                   // pf_alternative only: the prefix filter's other kernel (reported as ENG_PF)
constexpr uint32_t kEvAllPairs = 16384;                  // events the all-pairs rank orders (k_ev_rank)
constexpr uint64_t kSortMaxEvents = uint64_t(12) << 20;  // events the bucket order pass takes (event_order.hip)
constexpr uint64_t kProbeMinSpan = uint64_t(16) << 20;   // shards below this pay less for an abandoned pass than a probe is worth
acgpu_status pf_route_prepare(Scratch* sc, const HotTables& h, uint64_t span_bytes, PfRoute* r);
acgpu_status ensure_order_work(Scratch* sc, size_t bytes, hipStream_t);
acgpu_status ensure_probe(Scratch* sc, hipStream_t stream);
bool tri_walk_selected(uint32_t eng, const DeviceState* ds);
acgpu_status cnfa_tri_events(const DeviceState* ds, Scratch* sc, const ScanGeom& g, uint64_t span_bytes, hipStream_t stream, TriEvents* ev);
hipError_t launch_generic_count(uint32_t eng, DeviceState* ds, const ScanGeom& g, uint32_t* counts, hipStream_t stream, const TriEvents* tev = nullptr);
uint32_t pf_alternative(const acgpu_automaton* aut, const DeviceState* ds, PfRoute* route);
// scratch of the LDS walk's event form for geometry g (lds_emit.hip); *gen: the call's value of the overflow word
acgpu_status ensure_lw_events(Scratch* sc, const ScanGeom& g, hipStream_t stream, uint32_t* gen);
acgpu_status overlapping_entry(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end,
                               acgpu_match* out, size_t cap, size_t* n_out, acgpu_profile* prof);
// the engine plan's inputs for this automaton on this device
EngineFacts engine_facts(const acgpu_automaton* aut, const DeviceState* ds);
// enqueue-only overlapping search (acgpu_enqueue_overlapping*); `guess`: see EnqueueGuess
acgpu_status enqueue_impl(acgpu_automaton* aut, const acgpu_input* in, size_t shard_begin, size_t shard_end, acgpu_match* out,
                          size_t cap, uint64_t* totals, int32_t slot, uint32_t flags, bool* probed, EnqueueGuess* guess = nullptr,
                          EnqueueSync* sync = nullptr);
// ---- capi_find.cpp
acgpu_status serial_impl(acgpu_automaton* aut, const acgpu_input* in, bool single, acgpu_match* out, size_t cap,
                         size_t* n_out, acgpu_profile* prof);
bool parallel_find_eligible(const acgpu_automaton* aut, const acgpu_input* in);
// (direct / direct_cap: a device buffer of the caller that receives the selection itself when it can hold the whole
// occurrence stream; *went_direct says whether it did -- otherwise the selection is in sc->sel)
acgpu_status nonoverlapping_core(acgpu_automaton* occ, DeviceState* ds, Scratch* sc, const acgpu_input* in,
                                 size_t shard_begin, size_t shard_end, size_t pos0, int rule_kind, uint64_t* n_sel,
                                 acgpu_profile* prof, DenseRule* dense, acgpu_match* direct = nullptr, size_t direct_cap = 0,
                                 bool* went_direct = nullptr);
acgpu_status nonoverlapping_parallel(acgpu_automaton* aut, const acgpu_input* in, int rule_kind, acgpu_match* out,
                                     size_t cap, size_t* n_out, acgpu_profile* prof, DenseRule* dense);
acgpu_status nonoverlapping_windowed(acgpu_automaton* aut, const acgpu_input* in, int rule, acgpu_match* out, size_t cap,
                                     size_t* n_out, DenseRule* dense);
acgpu_status check_nonoverlapping(acgpu_automaton* aut, const acgpu_input* in);

}  // namespace acgpu_capi
