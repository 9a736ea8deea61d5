// acgpu_find_overlapping_multi: one overlapping search partitioned over several devices of a node, from ONE host process
// through the C ABI (contiguous byte shards, max_pattern_len-1 bytes of warm-up at every seam, a
// shard owns the matches that end inside it; the only exchange is the gather of the records in shard order).
//
//   scan    every shard is ENQUEUED on its own device's stream (acgpu_find_overlapping_enqueue: filter scan -> event
//           rank -> ordered records, or count -> scan -> fill for automata of the other engines; no host round trip), so
//           all devices scan concurrently; a shard the enqueue form could not finish (more than 16 384 occurrences
//           through the event form, an abandoned scan, records that outgrew the shard's buffer) is repeated with the
//           synchronous acgpu_find_overlapping_shard.
//   gather  record counts are read back (16 bytes per shard), prefix-summed on the host, and the records move to their
//           final slots in `out` on the destination device: over RCCL (ncclSend / ncclRecv inside one group, xGMI
//           peer-to-peer underneath) when the shards live on distinct devices, with hipMemcpyPeerAsync otherwise (several
//           virtual shards on one device: what a 1-GPU box can run).  librccl.so is opened at run time (dlopen): the
//           library has no link-time dependency on it, and a node without RCCL still gets the copy path.
//
// The reference has no counterpart (it is single-threaded); the seam rule is the bound its stream searcher keeps,
// src/automaton.rs:1108.
#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "acgpu.h"
#include "capi_internal.hpp"
#include "device/kernels.hpp"
#include "host/devbuf.hpp"

using namespace acgpu;

namespace {

thread_local std::string g_multi_error;
std::atomic<int> g_last_transport{0};   // 0 none yet, 1 copies, 2 RCCL

// ---- RCCL through dlopen: just the six entry points the gather needs (signatures of rccl/rccl.h, ROCm 7.2)
struct Rccl {
    using comm_t = void*;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    Rccl() {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        auto sym = [&](const char* n) { return dlsym(h, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
Rccl& rccl() { static Rccl r; return r; }
constexpr int kNcclUint8 = 1;   // ncclUint8, rccl.h

// one communicator per distinct device list (ranks = positions in the list), kept for the life of the process
struct CommSet { std::vector<int> devs; std::vector<Rccl::comm_t> comms; };
std::mutex g_comm_mu;
std::map<std::vector<int>, std::unique_ptr<CommSet>> g_comms;
CommSet* get_comms(const std::vector<int>& devs) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    auto& p = g_comms[devs];
    if (p) return p.get();
    auto cs = std::make_unique<CommSet>();
    cs->devs = devs;
    cs->comms.assign(devs.size(), nullptr);
    if (rccl().CommInitAll(cs->comms.data(), int(devs.size()), devs.data()) != 0) return nullptr;
    p = std::move(cs);
    return p.get();
}

// Streams: a pool per device.  A call checks out ONE stream per device it uses and returns it at the end, so that two
// host threads searching the same automaton never share a stream (the enqueue-only form keeps its scratch per
// (automaton, stream): sharing one would race on it), while the number of streams -- and of those scratch contexts --
// stays at the peak concurrency instead of growing with every call.
std::mutex g_stream_mu;
std::map<int, std::vector<hipStream_t>> g_free_streams;
hipError_t checkout_stream(int dev, hipStream_t* out) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        auto& v = g_free_streams[dev];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);   // (the caller has made `dev` current)
}
void return_stream(int dev, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_free_streams[dev].push_back(s);
}
struct StreamLease {   // the streams of one call, by device
    std::map<int, hipStream_t> held;
    ~StreamLease() { for (auto& kv : held) return_stream(kv.first, kv.second); }
    hipError_t get(int dev, hipStream_t* out) {
        auto it = held.find(dev);
        if (it != held.end()) { *out = it->second; return hipSuccess; }
        hipError_t e = checkout_stream(dev, out);
        if (e == hipSuccess) held[dev] = *out;
        return e;
    }
};

void set_error(const std::string& msg) {   // (also what acgpu_last_error reports: the bindings read that one)
    g_multi_error = msg;
    acgpu_set_last_error(msg.c_str());
}
acgpu_status fail(hipError_t e, const char* what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    (void)hipGetLastError();
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return ACGPU_ERR_NO_DEVICE;
    return e == hipErrorOutOfMemory ? ACGPU_ERR_NOMEM : ACGPU_ERR_HIP;
}
#define MHIP(expr)                                            \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return fail(e_, #expr);         \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct ShardWork {
    DevBuf recs, totals;
    hipStream_t stream = nullptr;
    size_t cap = 0;
    uint64_t n = 0;
    bool enqueued = false;
};

}  // namespace

extern "C" {

const char* acgpu_multi_last_error(void) { return g_multi_error.c_str(); }
int32_t acgpu_multi_last_transport(void) { return g_last_transport.load(); }

acgpu_status acgpu_find_overlapping_multi(acgpu_automaton* aut, const acgpu_shard* shards, size_t n_shards,
                                          int32_t dst_device, acgpu_match* out, size_t cap, size_t* n_out,
                                          uint64_t* shard_counts) {
    if (!aut || !n_out || (n_shards && !shards)) return ACGPU_ERR_INVALID_ARGUMENT;
    *n_out = 0;
    if (n_shards == 0) return ACGPU_OK;
    if (n_shards > 1 && acgpu_min_pattern_len(aut) == 0) {
        // an empty pattern matches at every position, the seam included: a shard whose halo is empty cannot tell whether
        // the match at its first position belongs to it or to its left neighbour (include/acgpu.h)
        set_error("acgpu_find_overlapping_multi: automata with an empty pattern cannot be sharded");
        return ACGPU_ERR_INVALID_ARGUMENT;
    }
    DeviceGuard guard;
    StreamLease lease;   // (declared before `work`: the streams go back to the pool after the shard buffers are released)
    std::vector<std::unique_ptr<ShardWork>> work(n_shards);

    // ---- 1. enqueue every shard on its device
    for (size_t i = 0; i < n_shards; i++) {
        const acgpu_shard& sh = shards[i];
        MHIP(hipSetDevice(sh.device));
        acgpu_status st = acgpu_upload(aut, sh.device);
        if (st) return st;
        auto w = std::make_unique<ShardWork>();
        MHIP(lease.get(sh.device, &w->stream));
        w->cap = std::max<size_t>(4096, std::min<size_t>(size_t(1) << 16, (sh.shard_end - sh.shard_begin) / 1024));
        MHIP(w->recs.ensure(w->cap * sizeof(acgpu_match)));
        MHIP(w->totals.ensure(2 * sizeof(uint64_t)));
        acgpu_input in{};
        in.haystack = sh.haystack; in.haystack_len = sh.haystack_len; in.span_start = sh.span_start; in.span_end = sh.span_end;
        in.haystack_on_device = 1; in.out_on_device = 1; in.stream = w->stream;
        st = acgpu_find_overlapping_enqueue(aut, &in, sh.shard_begin, sh.shard_end, w->recs.as<acgpu_match>(), w->cap,
                                            w->totals.as<uint64_t>(), -1);
        if (st == ACGPU_OK) w->enqueued = true;
        else if (st != ACGPU_ERR_INVALID_ARGUMENT) return st;   // INVALID_ARGUMENT: not an automaton of the enqueue form -> synchronous below
        work[i] = std::move(w);
    }
    // ---- 2. counts; shards the enqueue form could not finish are repeated synchronously
    uint64_t total = 0;
    std::vector<uint64_t> offs(n_shards, 0);
    for (size_t i = 0; i < n_shards; i++) {
        const acgpu_shard& sh = shards[i];
        ShardWork& w = *work[i];
        MHIP(hipSetDevice(sh.device));
        bool redo = !w.enqueued;
        if (w.enqueued) {
            uint64_t t[2] = {0, 0};
            MHIP(hipMemcpyAsync(t, w.totals.p, sizeof t, hipMemcpyDeviceToHost, w.stream));
            MHIP(hipStreamSynchronize(w.stream));
            w.n = t[0];
            redo = t[1] > ACGPU_ENQUEUE_MAX_EVENTS || t[0] > w.cap;
        }
        if (redo) {
            acgpu_input in{};
            in.haystack = sh.haystack; in.haystack_len = sh.haystack_len; in.span_start = sh.span_start; in.span_end = sh.span_end;
            in.haystack_on_device = 1; in.out_on_device = 1; in.stream = w.stream;
            for (;;) {
                size_t n = 0;
                acgpu_status st = acgpu_find_overlapping_shard(aut, &in, sh.shard_begin, sh.shard_end, w.recs.as<acgpu_match>(), w.cap, &n, nullptr);
                if (st == ACGPU_ERR_BUFFER_TOO_SMALL && n > w.cap) {
                    w.cap = n;
                    MHIP(w.recs.ensure(w.cap * sizeof(acgpu_match)));
                    continue;
                }
                if (st) return st;
                w.n = n;
                break;
            }
        }
        offs[i] = total;
        total += w.n;
        if (shard_counts) shard_counts[i] = w.n;
    }
    *n_out = size_t(total);
    if (total > cap) return ACGPU_ERR_BUFFER_TOO_SMALL;
    if (total == 0) return ACGPU_OK;
    if (!out) return ACGPU_ERR_INVALID_ARGUMENT;

    // ---- 3. local -> global coordinates, on the device that holds the records
    for (size_t i = 0; i < n_shards; i++) {
        ShardWork& w = *work[i];
        if (!w.n || !shards[i].global_offset) continue;
        MHIP(hipSetDevice(shards[i].device));
        MHIP(launch_offset_records(w.recs.as<acgpu_match>(), w.n, shards[i].global_offset, w.stream));
    }
    // ---- 4. gather in shard order.  RCCL when every shard has its own device (and RCCL can be opened); copies otherwise.
    std::vector<int> devs;
    bool distinct = true;
    for (size_t i = 0; i < n_shards; i++) {
        for (int d : devs) if (d == shards[i].device) distinct = false;
        devs.push_back(shards[i].device);
    }
    int dst_rank = -1;
    for (size_t i = 0; i < devs.size(); i++) if (devs[i] == dst_device) dst_rank = int(i);
    const bool force_rccl = std::getenv("ACGPU_MULTI_FORCE_RCCL") != nullptr;   // test knob, read per call: RCCL even for one device
    const bool no_rccl = std::getenv("ACGPU_MULTI_NO_RCCL") != nullptr;
    bool use_rccl = !no_rccl && rccl().ok && distinct && dst_rank >= 0 && (n_shards > 1 || force_rccl);
    CommSet* cs = use_rccl ? get_comms(devs) : nullptr;
    if (use_rccl && !cs) use_rccl = false;   // communicator creation failed: copies
    if (use_rccl) {
        hipStream_t dst_stream = work[size_t(dst_rank)]->stream;
        // Nothing returns between GroupStart and GroupEnd: an open group would poison every later RCCL call of this
        // thread.  The first failure is remembered, the group is always closed, then the error is reported.
        std::string err;
        if (rccl().GroupStart() != 0) err = "ncclGroupStart failed";
        else {
            for (size_t i = 0; i < n_shards && err.empty(); i++) {
                ShardWork& w = *work[i];
                if (!w.n) continue;
                const size_t bytes = size_t(w.n) * sizeof(acgpu_match);
                hipError_t he = hipSetDevice(shards[i].device);
                if (he != hipSuccess) { err = std::string("hipSetDevice: ") + hipGetErrorString(he); break; }
                if (rccl().Send(w.recs.p, bytes, kNcclUint8, dst_rank, cs->comms[i], w.stream) != 0) { err = "ncclSend failed"; break; }
                he = hipSetDevice(dst_device);
                if (he != hipSuccess) { err = std::string("hipSetDevice: ") + hipGetErrorString(he); break; }
                if (rccl().Recv(out + offs[i], bytes, kNcclUint8, int(i), cs->comms[size_t(dst_rank)], dst_stream) != 0) err = "ncclRecv failed";
            }
            if (rccl().GroupEnd() != 0 && err.empty()) err = "ncclGroupEnd failed";
        }
        if (!err.empty()) { set_error("RCCL gather: " + err); (void)hipGetLastError(); return ACGPU_ERR_HIP; }
        g_last_transport = 2;
    } else {
        for (size_t i = 0; i < n_shards; i++) {
            ShardWork& w = *work[i];
            if (!w.n) continue;
            MHIP(hipSetDevice(shards[i].device));
            const size_t bytes = size_t(w.n) * sizeof(acgpu_match);
            if (shards[i].device == dst_device) MHIP(hipMemcpyAsync(out + offs[i], w.recs.p, bytes, hipMemcpyDeviceToDevice, w.stream));
            else MHIP(hipMemcpyPeerAsync(out + offs[i], dst_device, w.recs.p, shards[i].device, bytes, w.stream));
        }
        g_last_transport = 1;
    }
    for (size_t i = 0; i < n_shards; i++) {
        MHIP(hipSetDevice(shards[i].device));
        MHIP(hipStreamSynchronize(work[i]->stream));
    }
    return ACGPU_OK;
}

acgpu_status acgpu_device_count(int32_t* n) {
    if (!n) return ACGPU_ERR_INVALID_ARGUMENT;
    int c = 0;
    MHIP(hipGetDeviceCount(&c));
    *n = c;
    return ACGPU_OK;
}
acgpu_status acgpu_device_malloc(int32_t device, size_t bytes, void** out) {
    if (!out) return ACGPU_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    DeviceGuard guard;
    MHIP(hipSetDevice(device));
    MHIP(hipMalloc(out, bytes ? bytes : 1));
    return ACGPU_OK;
}
acgpu_status acgpu_device_free(int32_t device, void* p) {
    if (!p) return ACGPU_OK;
    DeviceGuard guard;
    MHIP(hipSetDevice(device));
    MHIP(hipFree(p));
    return ACGPU_OK;
}
acgpu_status acgpu_device_copy(int32_t device, void* dst, const void* src, size_t bytes, int32_t kind) {
    if (bytes && (!dst || !src)) return ACGPU_ERR_INVALID_ARGUMENT;
    if (kind < 0 || kind > 2) return ACGPU_ERR_INVALID_ARGUMENT;
    DeviceGuard guard;
    MHIP(hipSetDevice(device));
    MHIP(hipMemcpy(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice));
    return ACGPU_OK;
}

}  // extern "C"
