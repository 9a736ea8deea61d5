// Match events of the LDS transition walk (one-row-per-state form, kLwFull): the ordered records of a search WITHOUT a
// second walk over the haystack.
//
// The reference's overlapping loop emits inline (src/automaton.rs:1491-1534: `get_match` at every match state entered).
// A wavefront cannot do that -- its 64 lanes walk 64 different texts, and a record written by one lane stalls the other 63 --
// and the chunk fill of lds_walk.hip (k_lw_fill: one wavefront per non-empty 2 KiB chunk, every lane re-walking its share
// behind a warm-up, twice) runs at a fraction of the count walk's geometry: 165 us for 0.5 M records, 310 us for 9 M, behind
// a 65 us count (profiles/r06_call_timelines_before.txt).  Here the count walk itself (k_lw_count_ev, the walk of k_lw_count)
// notes every dword in which a lane entered a match state as ONE 16-byte event
//     { dword index in the shard, records of the lane-chunk so far, state before the dword | byte masks, the four bytes }
// -- wave-compacted into an LDS queue (ballot + mbcnt) and flushed to the SLAB of the wave's task: every task (64 lane-chunks,
// 32 KiB of haystack) owns room for one event per 16 haystack bytes, so appending takes no atomic (a first version with one
// global list paid 12 ns per flush on its counter -- same-address atomics serialise in L2 -- and a 49 us stampede when all
// 4 096 wavefronts flushed their remainders at the end) -- and k_lw_ev_emit turns the slabs into records with every lane
// busy: one wavefront per task, one event per lane, four steps from the saved state through the LDS image, the match lists
// {pattern, length} of the states entered (src/dfa.rs:275-286) read from the image, records written at   offset of the
// lane-chunk (scan of the per-lane-chunk counts) + records so far   -- their place in the reference's order, whatever order
// the events were appended in; a task's records are one contiguous range written by one wavefront, so its 24-byte stores
// meet in one L2 (events taken from a global list by whichever workgroup came next wrote every 128-byte line in pieces
// from several XCDs: 0.9 TB/s).  The scan counts lane-chunks (512 B), so that a lane knows its own rank; a task with more
// events than its slab holds (a record every few bytes: the call is bound by its record writes then) marks the call, and
// k_lw_fill fills the records instead.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "../host/lw_tables.hpp"
#include "hot.hpp"
#include "launch_util.hpp"
#include "lw_dev.hpp"

namespace acgpu {

namespace {

using namespace lwdev;

constexpr int kEvBlock = 1024;
constexpr int kEvWaves = kEvBlock / 64;
// events per wave queue: flushed from LwEvArgs::q_flush on, checked once per 16-byte piece -- four dwords of 64 lanes may arrive
// in between, so the queue holds q_flush + 256.  The queues take the LDS the image leaves (ev_queue_flush): a flush stalls
// its wave for the round trip of its stores (the prefetched haystack lines share their counter), 14 times per task at a
// threshold of 64 and a record every 28 bytes -- 125 us against the 65 us of the plain count walk
constexpr uint32_t kEvFlushMin = 64, kEvSlack = 4 * 64;
constexpr uint32_t kEvQueueBytesMin = kEvWaves * (kEvFlushMin + kEvSlack) * 16;

struct LwEvArgs {
    uint4* ev;                    // slabs: slab_events events per task
    uint32_t* task_n;             // [n_tasks] events of each task
    uint32_t* task_rec;           // [n_tasks] records of each task | [n_tasks] lane-chunks of it that have records (count walk: out; emit: unused)
    const uint64_t* task_off;     // [n_tasks] exclusive prefix of task_rec (emit: in)
    // emit, enqueue-only form: report[0] <- records, report[1] <- 0, or UINT64_MAX when a slab overflowed and no chunk fill is
    // queued behind this kernel (report_fail): "repeat with the synchronous call", like an abandoned filter scan (acgpu.h)
    uint64_t* report;
    uint32_t report_fail;
    uint32_t* overflow;           // *overflow = gen when a task had more events than its slab holds
    uint32_t gen;                 // (a new value per call: the word is never reset)
    uint32_t slab_events;
    uint32_t q_flush;             // a wave flushes its queue from this many events on
    uint32_t q_off;               // LDS byte offset of the workgroup's queues (behind the image)
};

// The queue of one wavefront.  `n` (waiting) and `pos` (already in the slab) are wave-uniform.
struct LwEvQ {
    uint8_t* q;
    uint32_t n, pos;
    bool dead;        // the call has overflowed already: no more events (wave-uniform)
    uint4* slab;      // of the wave's current task
    int lane;
    LwEvArgs a;
    // appends the events of the lanes with f set (m = their ballot, not zero)
    __device__ __forceinline__ void push(bool f, unsigned long long m, const uint4& e) {
        const uint32_t r = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        if (f) *reinterpret_cast<uint4*>(q + 16u * (n + r)) = e;
        n += uint32_t(__popcll(m));
    }
    // everything waiting goes to the slab
    __device__ __forceinline__ void flush() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (n == 0) return;
        for (uint32_t i = uint32_t(lane); i < n; i += 64)
            if (pos + i < a.slab_events) slab[pos + i] = *reinterpret_cast<const uint4*>(q + 16u * i);
        pos += n;
        n = 0;
        // retire the stores before the walk goes on: with store-type operations pending the compiler orders the next use of
        // a prefetched line with s_waitcnt vmcnt(0), draining the haystack prefetch at every dword (see pf_scan.hip)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_wave_barrier();
    }
    // (a call one of whose slabs overflowed is filled by k_lw_fill: the tasks that start after that only count)
    __device__ __forceinline__ void begin_task(uint64_t task) {
        slab = a.ev + task * a.slab_events; pos = 0;
        dead = __builtin_amdgcn_readfirstlane(int(*const_cast<volatile uint32_t*>(a.overflow))) == int(a.gen);
    }
    __device__ __forceinline__ void end_task(uint64_t task) {
        flush();
        if (lane == 0) {
            a.task_n[task] = pos;
            if (pos > a.slab_events) *a.overflow = a.gen;
        }
    }
};

// events a task's slab holds: one per 8 haystack bytes of its 64 lane-chunks (the scratch is twice the haystack; a record
// every 14 bytes -- the reference's teddy3-48pat-common -- overflowed slabs of half that and fell to the chunk fill at half the rate)
__host__ __device__ constexpr uint32_t kEvSlabPerChunk(uint32_t chunk) { return 8u * chunk; }

// event word 2: row address of the state before the dword << 16 | owned-byte mask << 4 | walked-byte mask
__device__ __forceinline__ uint32_t ev_state(uint32_t h0, uint32_t walked, uint32_t owned) { return (h0 & 0xFFFF0000u) | (owned << 4) | walked; }

// One dword of the interior walk: the four steps of lw_step4<kLwFull>; returns the records it gained (0 while warming up).
template <bool CC, bool OWNED>
__device__ __forceinline__ uint32_t ev_step4(const LwLds& L, const LwCc& cc, uint32_t w, uint32_t& h) {
    const uint32_t cv0 = lw_clsval<CC, 0, true>(L, cc, w), cv1 = lw_clsval<CC, 1, true>(L, cc, w);
    const uint32_t cv2 = lw_clsval<CC, 2, true>(L, cc, w), cv3 = lw_clsval<CC, 3, true>(L, cc, w);
    const uint32_t h1 = L.rd32(lw_addr_full(h, cv0));
    const uint32_t h2 = L.rd32(lw_addr_full(h1, cv1));
    const uint32_t h3 = L.rd32(lw_addr_full(h2, cv2));
    h = L.rd32(lw_addr_full(h3, cv3));
    return OWNED ? (h1 + h2 + h3 + h) & kLwFullSumMask : 0u;
}

// One 16-byte piece of the interior walk: its four dwords walked back to back, THEN the events of those that gained a
// record -- one wave-uniform branch per piece.  (A branch per dword put a control dependency behind every fourth gather:
// the class lookups of the next dword could not be issued before the last gather of this one had come back and been
// tested -- 120 us for a count walk that takes 65 without events, on text with a match every 28 bytes where every dword
// of every wave has a flagged lane.  No branch hint: marked unlikely, the pushes went to a cold block far from the loop.)
template <bool CC>
__device__ __forceinline__ void ev_piece(const LwLds& L, const LwCc& cc, const uint4& q, uint32_t& h, uint32_t& cnt, uint32_t gd, LwEvQ& Q) {
    const uint32_t h0 = h;
    const uint32_t c0 = ev_step4<CC, true>(L, cc, q.x, h);
    const uint32_t h1 = h;
    const uint32_t c1 = ev_step4<CC, true>(L, cc, q.y, h);
    const uint32_t h2 = h;
    const uint32_t c2 = ev_step4<CC, true>(L, cc, q.z, h);
    const uint32_t h3 = h;
    const uint32_t c3 = ev_step4<CC, true>(L, cc, q.w, h);
    const uint32_t n0 = cnt, n1 = n0 + c0, n2 = n1 + c1, n3 = n2 + c2;
    cnt = n3 + c3;
    if (__ballot((c0 | c1 | c2 | c3) != 0) != 0 && !Q.dead) {
        unsigned long long m;
        if ((m = __ballot(c0 != 0)) != 0) Q.push(c0 != 0, m, make_uint4(gd, n0, ev_state(h0, 0xFu, 0xFu), q.x));
        if ((m = __ballot(c1 != 0)) != 0) Q.push(c1 != 0, m, make_uint4(gd + 1, n1, ev_state(h1, 0xFu, 0xFu), q.y));
        if ((m = __ballot(c2 != 0)) != 0) Q.push(c2 != 0, m, make_uint4(gd + 2, n2, ev_state(h2, 0xFu, 0xFu), q.z));
        if ((m = __ballot(c3 != 0)) != 0) Q.push(c3 != 0, m, make_uint4(gd + 3, n3, ev_state(h3, 0xFu, 0xFu), q.w));
    }
}

// Edge walk (first / last regions of a shard, small inputs): the exact byte loop of lw_edge_walk with one event per dword
// that gained a record -- the masks say which of its bytes were walked and which of them this lane-chunk owns.  The loop
// is wave-uniform (its trip count is the longest lane's; a lane without bytes left walks nothing): the queue is the wave's.
template <bool CC>
__device__ __forceinline__ uint32_t ev_edge_walk(const LwArgs& a, const LwLds& L, const ScanGeom& g, uint64_t w, uint64_t lo,
                                                 uint64_t hi, LwEvQ& Q) {
    uint32_t cnt = 0, h = a.start;
    const uint64_t p0 = w & ~uint64_t(15);
    uint32_t trips = hi > p0 ? uint32_t((hi - p0 + 15) >> 4) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) trips = max(trips, uint32_t(__shfl_xor(int(trips), o, 64)));
    auto ld = [&](uint64_t p) {
        if (p < hi) ACGPU_HAY_CHECK(g, p, 16);
        return p < hi ? *reinterpret_cast<const uint4*>(g.hay16 + p) : make_uint4(0, 0, 0, 0);
    };
    uint4 q0 = ld(p0), q1 = ld(p0 + 16);
    uint64_t p = p0;
    for (uint32_t t = 0; t < trips; t++, p += 16) {
        const uint4 q = q0;
        q0 = q1;
        q1 = ld(p + 32);
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t h0 = h, c0 = cnt;
            uint32_t walked = 0, owned = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t v = p + 4 * d + k;
                if (v >= w && v < hi) {
                    walked |= 1u << k;
                    h = L.rd32((h >> 16) + lw_clsval_byte<CC, true>(L, a, (wd[d] >> (8 * k)) & 0xFFu));
                    if (v >= lo) { owned |= 1u << k; cnt += h & kLwFullLenMask; }
                }
            }
            const bool f = cnt != c0;
            const unsigned long long m = __ballot(f);
            if (m && !Q.dead) Q.push(f, m, make_uint4(uint32_t((p + 4 * d - g.grid0) >> 2), c0, ev_state(h0, walked, owned), wd[d]));
        }
        if (Q.n >= Q.a.q_flush) Q.flush();
    }
    return cnt;
}

// The count walk with events.  Geometry as k_lw_count<8, kLwFull, CC> with one lane-chunk per count chunk
// (a.lanes_per_chunk == 1: counts[] is per lane-chunk).
template <bool CC>
__global__ __launch_bounds__(kEvBlock) void k_lw_count_ev(LwArgs a, ScanGeom g, uint32_t* __restrict__ counts, LwEvArgs ea) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLwLdsBytes];   // static, at LDS address 0: no base add per lookup
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.image);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (uint32_t i = threadIdx.x; i < a.image_bytes / 16; i += kEvBlock) dst[i] = src[i];
    }
    __syncthreads();
    constexpr int UP = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_id = uint64_t(blockIdx.x) * kEvWaves + wave;
    const uint64_t n_waves = uint64_t(gridDim.x) * kEvWaves;
    const uint32_t C = a.lane_chunk;
    const uint32_t warm_bytes = a.warm_pieces * 16;
    const uint32_t n_units = C / (16 * UP);
    const LwLds L{lds};
    LwCc cc;
    cc.add = uint32_t(a.cc_add); cc.hi = uint32_t(a.cc_hi);
    asm volatile("v_mov_b32 %0, %1" : "=v"(cc.v_lo) : "s"(a.cc_lo));
    LwEvQ Q;
    Q.q = lds + ea.q_off + uint32_t(wave) * ((ea.q_flush + kEvSlack) * 16);
    Q.n = 0; Q.pos = 0; Q.dead = false; Q.slab = ea.ev; Q.lane = lane; Q.a = ea;

    const uint64_t region_bytes = uint64_t(64) * C;
    auto is_interior = [&](uint64_t lo) {
        return lo >= g.emit_lo && lo + region_bytes <= g.emit_hi && lo >= g.cold_floor + warm_bytes;
    };
    uint4 ua[UP], ub[UP];
    bool have_ua = false;
    for (uint64_t task = wave_id; task < a.n_tasks; task += n_waves) {
        const uint64_t j0 = task * 64;
        const uint64_t region_lo = g.grid0 + j0 * C;
        const bool interior = is_interior(region_lo);
        const uint64_t next_lo = region_lo + n_waves * region_bytes;
        const bool next_interior = task + n_waves < a.n_tasks && is_interior(next_lo);
        uint32_t cnt = 0;
        Q.begin_task(task);
        if (interior) {
            const uint8_t* p_main = g.hay16 + region_lo + uint64_t(lane) * C;
            uint32_t gd = uint32_t((region_lo - g.grid0 + uint64_t(lane) * C) >> 2);   // dword index of the lane-chunk's first dword
            uint32_t h = a.start;
            auto ld = [&](const uint8_t* p) {
                ACGPU_HAY_CHECK(g, uint64_t(p - g.hay16), 16);
                return *reinterpret_cast<const uint4*>(p);
            };
            auto ld_unit_at = [&](uint4 (&u)[UP], const uint8_t* p) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < UP; k++) u[k] = ld(p + 16 * k);
            };
            auto warm_piece = [&](const uint4& q) __attribute__((always_inline)) {
                (void)ev_step4<CC, false>(L, cc, q.x, h);
                (void)ev_step4<CC, false>(L, cc, q.y, h);
                (void)ev_step4<CC, false>(L, cc, q.z, h);
                (void)ev_step4<CC, false>(L, cc, q.w, h);
            };
            auto do_unit = [&](const uint4 (&u)[UP]) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < UP; k++) {
                    ev_piece<CC>(L, cc, u[k], h, cnt, gd + 4 * k, Q);
                    if (__builtin_expect(Q.n >= ea.q_flush, 0)) Q.flush();
                }
                gd += 4 * UP;
            };
            {
                uint4 wq = make_uint4(0, 0, 0, 0);
                if (a.warm_pieces) wq = ld(p_main - 16 * a.warm_pieces);
                if (!have_ua) ld_unit_at(ua, p_main);
                for (uint32_t wp = a.warm_pieces; wp > 0; wp--) {
                    warm_piece(wq);
                    if (wp > 1) wq = ld(p_main - 16 * (wp - 1));
                }
            }
#pragma unroll 1
            for (uint32_t u0 = 0; u0 < n_units; u0 += 2) {
                ld_unit_at(ub, p_main + (16 * UP) * (u0 + 1 < n_units ? u0 + 1 : n_units - 1));
                do_unit(ua);
                {
                    const bool more = u0 + 2 < n_units;
                    const uint8_t* pn = more ? p_main + (16 * UP) * (u0 + 2)
                                             : next_interior ? g.hay16 + next_lo + uint64_t(lane) * C
                                                             : p_main + (16 * UP) * (n_units - 1);
                    ld_unit_at(ua, pn);
                }
                if (u0 + 1 < n_units) do_unit(ub);
            }
            have_ua = next_interior;
        } else {
            have_ua = false;
            const uint64_t j = j0 + uint64_t(lane);
            uint64_t w = 0, lo = 0, hi = 0;
            if (j < a.n_lane_chunks) {
                const uint64_t glo = g.grid0 + j * C, ghi = glo + C;
                lo = glo > g.emit_lo ? glo : g.emit_lo;
                hi = ghi < g.emit_hi ? ghi : g.emit_hi;
                if (hi > lo) {
                    w = lo >= g.halo ? lo - g.halo : 0;
                    if (w < g.cold_floor) w = g.cold_floor;
                } else hi = lo = 0;
            }
            cnt = ev_edge_walk<CC>(a, L, g, w, lo, hi, Q);   // (every lane: the loop inside is wave-uniform)
        }
        {
            const uint64_t j = j0 + uint64_t(lane);
            const bool in = j < a.n_lane_chunks;
            if (in) counts[j] = cnt;
            // the task's records and non-empty lane-chunks: what the one-workgroup scan (k_lw_task_scan) adds up
            uint32_t t = in ? cnt : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += uint32_t(__shfl_xor(int(t), o, 64));
            const uint32_t act = uint32_t(__popcll(__ballot(in && cnt != 0)));
            if (lane == 0) { ea.task_rec[task] = t; ea.task_rec[a.n_tasks + task] = act; }
        }
        Q.end_task(task);
    }
}

// Events -> records.  One wavefront per task, one event per lane; the image of the automaton in LDS (dynamic: small automata
// leave room for several workgroups per CU).
//
// A slab holds its events in the order the wave met them: one dword step after the other, the flagged lanes of each -- 64
// consecutive events belong to 64 different lane-chunks, their records to 64 different cache lines, and a line had to wait
// in L2 for the other four records it holds until the wave came round again: with 8 192 such waves the open lines did not
// survive (0.9 TB/s of record writes, profiles/r06_lw_events_ab.txt).  So the wave takes its events in windows of 512 and
// counting-sorts each window by lane-chunk in LDS (64 bins: one LDS atomic for the rank, a wave scan for the bin starts):
// the records of a lane-chunk's share of the window -- eight on average at full windows -- are one contiguous piece, written by
// neighbouring lanes in one instruction.  The order inside a bin is whatever the atomics made it: every event carries the
// rank of its first record, so the place of a record never depends on the order of the events.
constexpr int kEmBlock = 256;
constexpr uint32_t kEmWindow = 512;                              // events sorted at a time (kEmWindow / 64 per lane in registers)
constexpr uint32_t kEmWaveLds = kEmWindow * 16 + 2 * 64 * 4 + 64 * 8;   // sorted events | bin counters | bin cursors | record offsets of the task's lane-chunks
// An event in two halves, so that a lane can have the walks of several events in flight before it writes anything: em_walk is
// branch-free (four dependent LDS gathers; a byte that was not walked leaves the state alone, a byte that is not owned
// reports no match), em_write reads the match lists of the states entered and stores the records.  One event after the
// other -- walk, lists, stores, next event -- left a wave waiting out ~12 dependent LDS round trips per event: 96 us for
// 9.4 M events before the first store was issued (profiles/r06_lw_events_ab.txt).
// (off_tab: the record offsets of the task's 64 lane-chunks, in LDS -- not a global gather per event)
template <bool CC>
__device__ __forceinline__ uint4 em_walk(const LwArgs& a, const LwLds& L, const uint4 e) {
    const uint32_t walked = e.z & 0xFu, owned = (e.z >> 4) & 0xFu;
    uint32_t h = e.z, r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t nh = L.rd32((h >> 16) + lw_clsval_byte<CC, true>(L, a, (e.w >> (8 * k)) & 0xFFu));
        h = ((walked >> k) & 1u) ? nh : h;
        r[k] = ((walked & owned) >> k) & 1u ? h : 0u;
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}
__device__ __forceinline__ void em_write(const LwArgs& a, const LwLds& L, const ScanGeom& g, const uint4 e, const uint4 hs, const uint64_t* off_tab,
                                         acgpu_match* __restrict__ out, uint32_t chunk_shift) {
    const uint32_t gd = e.x;
    acgpu_match* dst = out + off_tab[(gd >> chunk_shift) & 63u] + e.y;
    uint64_t end = g.grid0 + (uint64_t(gd) << 2) - g.base_mis;   // haystack offset of the dword's first byte
    const uint32_t hk[4] = {hs.x, hs.y, hs.z, hs.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        end++;
        const uint32_t h = hk[k], len = h & kLwFullLenMask;
        if (len == 0) continue;
        const uint32_t list = L.rd32((h >> 16) + 4 * a.list_col);
        for (uint32_t r = 0; r < len; r++) {
            const uint32_t pid = L.rd32(list + 8 * r), plen = L.rd32(list + 8 * r + 4);
            const uint64_t start = end - plen;
            uint32_t* p = reinterpret_cast<uint32_t*>(dst + r);   // acgpu_match: {u32 pattern, u32 pad, u64 start, u64 end}
            *reinterpret_cast<uint4*>(p) = make_uint4(pid, 0u, uint32_t(start), uint32_t(start >> 32));
            *reinterpret_cast<uint2*>(p + 4) = make_uint2(uint32_t(end), uint32_t(end >> 32));
        }
        dst += len;
    }
}

template <bool CC>
__global__ __launch_bounds__(kEmBlock) void k_lw_ev_emit(LwArgs a, ScanGeom g, LwEvArgs ea, const uint32_t* __restrict__ counts,
                                                        const uint64_t* __restrict__ totals, uint64_t cap, acgpu_match* __restrict__ out,
                                                        uint32_t chunk_shift) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_dyn[];
    if (ea.report && blockIdx.x == 0 && threadIdx.x == 0) {
        ea.report[0] = totals[0];
        ea.report[1] = (ea.report_fail && *ea.overflow == ea.gen) ? ~uint64_t(0) : uint64_t(0);
    }
    if (*ea.overflow == ea.gen || totals[0] > cap || totals[0] == 0) return;   // (overflow: k_lw_fill serves)
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.image);
        uint4* dst = reinterpret_cast<uint4*>(lds_dyn);
        for (uint32_t i = threadIdx.x; i < a.image_bytes / 16; i += kEmBlock) dst[i] = src[i];
    }
    __syncthreads();
    const LwLds L{lds_dyn};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* wl = lds_dyn + ((a.image_bytes + 15u) & ~15u) + uint32_t(wave) * kEmWaveLds;
    uint4* sorted = reinterpret_cast<uint4*>(wl);
    uint32_t* bins = reinterpret_cast<uint32_t*>(wl + kEmWindow * 16);
    uint32_t* cursor = bins + 64;
    uint64_t* off_tab = reinterpret_cast<uint64_t*>(cursor + 64);
    const uint64_t n_waves = uint64_t(gridDim.x) * (kEmBlock / 64);
    for (uint64_t task = uint64_t(wave) * gridDim.x + blockIdx.x; task < a.n_tasks; task += n_waves) {
        const uint32_t n = ea.task_n[task];
        if (n == 0) continue;
        const uint4* slab = ea.ev + task * ea.slab_events;
        {   // record offsets of the task's lane-chunks: the task's own (k_lw_task_scan) + the counts in front of each inside it
            const uint64_t j = task * 64 + uint64_t(lane);
            const uint32_t c = j < a.n_lane_chunks ? counts[j] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = uint32_t(__shfl_up(int(incl), o, 64));
                if (lane >= o) incl += t;
            }
            off_tab[lane] = ea.task_off[task] + (incl - c);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        for (uint32_t base = 0; base < n; base += kEmWindow) {
            const uint32_t m = n - base < kEmWindow ? n - base : kEmWindow;
            if (m <= 64) {   // (nothing to gain from sorting one row)
                for (uint32_t i = uint32_t(lane); i < m; i += 64) {
                    const uint4 e = slab[base + i];
                    em_write(a, L, g, e, em_walk<CC>(a, L, e), off_tab, out, chunk_shift);
                }
                continue;
            }
            // (the window's events stay in registers between the two passes: compile-time indices -- a runtime-indexed array
            // of eight uint4 went to scratch memory, 128 bytes per lane, and the kernel took 96 us before its first store)
            static_assert(kEmWindow == 8 * 64, "eight events per lane");
            bins[lane] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            auto take = [&](uint32_t k) __attribute__((always_inline)) {
                const uint32_t i = k * 64 + uint32_t(lane);
                uint32_t x = 0, y = 0, z = 0, w = 0;
                if (i < m) {
                    const uint4 t = slab[base + i];
                    x = t.x; y = t.y; z = t.z; w = t.w;
                    atomicAdd(&bins[(x >> chunk_shift) & 63u], 1u);
                }
                return make_uint4(x, y, z, w);
            };
            const uint4 e0 = take(0), e1 = take(1), e2 = take(2), e3 = take(3), e4 = take(4), e5 = take(5), e6 = take(6), e7 = take(7);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            {
                const uint32_t c = bins[lane];
                uint32_t incl = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = uint32_t(__shfl_up(int(incl), o, 64));
                    if (lane >= o) incl += t;
                }
                cursor[lane] = incl - c;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            auto place = [&](const uint4 e, uint32_t k) __attribute__((always_inline)) {
                const uint32_t i = k * 64 + uint32_t(lane);
                if (i < m) sorted[atomicAdd(&cursor[(e.x >> chunk_shift) & 63u], 1u)] = e;
            };
            place(e0, 0); place(e1, 1); place(e2, 2); place(e3, 3); place(e4, 4); place(e5, 5); place(e6, 6); place(e7, 7);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            {   // eight walks in flight, then the records
                auto get = [&](uint32_t k) __attribute__((always_inline)) {
                    const uint32_t i = k * 64 + uint32_t(lane);
                    return i < m ? sorted[i] : make_uint4(0, 0, 0, 0);
                };
                const uint4 s0 = get(0), s1 = get(1), s2 = get(2), s3 = get(3), s4 = get(4), s5 = get(5), s6 = get(6), s7 = get(7);
                const uint4 w0 = em_walk<CC>(a, L, s0), w1 = em_walk<CC>(a, L, s1), w2 = em_walk<CC>(a, L, s2), w3 = em_walk<CC>(a, L, s3);
                const uint4 w4 = em_walk<CC>(a, L, s4), w5 = em_walk<CC>(a, L, s5), w6 = em_walk<CC>(a, L, s6), w7 = em_walk<CC>(a, L, s7);
                em_write(a, L, g, s0, w0, off_tab, out, chunk_shift); em_write(a, L, g, s1, w1, off_tab, out, chunk_shift);
                em_write(a, L, g, s2, w2, off_tab, out, chunk_shift); em_write(a, L, g, s3, w3, off_tab, out, chunk_shift);
                em_write(a, L, g, s4, w4, off_tab, out, chunk_shift); em_write(a, L, g, s5, w5, off_tab, out, chunk_shift);
                em_write(a, L, g, s6, w6, off_tab, out, chunk_shift); em_write(a, L, g, s7, w7, off_tab, out, chunk_shift);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Exclusive prefix of the tasks' record counts, the totals of the call: ONE workgroup (a 256 MiB span has 8 192 tasks, an 8 GiB
// one 262 144) in place of the three launches of the lane-chunk scan (kernels.hip: 14 us of a 125 us call).  totals = {records,
// non-empty lane-chunks}; host (page-locked, device-visible) also receives them, and *extra32 as host[2].
constexpr int kTsBlock = 1024;
__global__ __launch_bounds__(kTsBlock) void k_lw_task_scan(const uint32_t* __restrict__ task_rec, uint64_t n_tasks, uint64_t* __restrict__ task_off,
                                                          uint64_t* __restrict__ totals, uint64_t* __restrict__ host, const uint32_t* __restrict__ extra32) {
    __shared__ uint64_t s_sum[kTsBlock / 64], s_act[kTsBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t per = (n_tasks + kTsBlock - 1) / kTsBlock;
    const uint64_t b0 = uint64_t(threadIdx.x) * per, b1 = b0 + per < n_tasks ? b0 + per : n_tasks;
    // (slices of up to kTsRegs tasks are held in registers: independent loads, one memory round trip -- a loop of dependent
    // read-modify-writes took 16.6 us for 8 192 tasks)
    constexpr int kTsRegs = 16;
    const bool in_regs = per <= uint64_t(kTsRegs);
    uint32_t xr[kTsRegs], xa[kTsRegs];
    uint64_t ls = 0, la = 0;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kTsRegs; j++) {
            const bool ok = b0 + j < b1;
            xr[j] = ok ? task_rec[b0 + j] : 0u;
            xa[j] = ok ? task_rec[n_tasks + b0 + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < kTsRegs; j++) { ls += xr[j]; la += xa[j]; }
    } else {
        for (uint64_t t = b0; t < b1; t++) { ls += task_rec[t]; la += task_rec[n_tasks + t]; }
    }
    uint64_t is = ls, ia = la;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t ts = __shfl_up(is, o, 64), ta = __shfl_up(ia, o, 64);
        if (lane >= o) { is += ts; ia += ta; }
    }
    if (lane == 63) { s_sum[wave] = is; s_act[wave] = ia; }
    __syncthreads();
    uint64_t rs = is - ls, ra = ia - la;
    for (int k = 0; k < wave; k++) { rs += s_sum[k]; ra += s_act[k]; }
    if (threadIdx.x == kTsBlock - 1) {
        totals[0] = rs + ls; totals[1] = ra + la;
        if (host) { host[0] = rs + ls; host[1] = ra + la; if (extra32) host[2] = *extra32; }
    }
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kTsRegs; j++) {
            if (b0 + j < b1) task_off[b0 + j] = rs;
            rs += xr[j];
        }
    } else {
        for (uint64_t t = b0; t < b1; t++) { task_off[t] = rs; rs += task_rec[t]; }
    }
}

LwArgs ev_lw_args(const HotTables& h, const ScanGeom& g) {
    const LwHostTables& t = h.lw;
    LwArgs la{};
    la.image = h.lw_image;
    la.image_bytes = h.lw_image_bytes; la.row_bytes = t.row_bytes;
    la.start = t.start;
    la.cc_add = t.cc_add; la.cc_lo = t.cc_lo; la.cc_hi = t.cc_hi;
    la.list_col = t.classes;
    la.lanes_per_chunk = 1;
    la.lane_chunk = g.chunk;
    la.warm_pieces = (g.halo + 15) / 16;
    la.n_lane_chunks = g.n_chunks;
    la.n_tasks = (la.n_lane_chunks + 63) / 64;
    return la;
}

}  // namespace

// Lane-chunk of the event form: 512 bytes, more when the warm-up (max_pattern_len - 1 bytes rounded up to 16) would exceed
// an eighth of it; 0 = the form does not apply (very long patterns: the chunk fill serves).
uint32_t lw_events_chunk(const HotTables& h, uint32_t halo, uint64_t span_bytes) {
    if (!lw_fill_supported(h)) return 0;
    if (h.lw_image_bytes + kEvQueueBytesMin + 64 > kLwLdsBytes) return 0;   // the wave queues sit behind the image
    const uint32_t warm = (halo + 15) & ~15u;
    // small spans: smaller lane-chunks, down to 128 bytes -- the first and last regions of a shard are walked byte by byte, one
    // lane-chunk per lane, and a 4 KiB haystack in eight chunks of 512 bytes was a 50 us loop on eight lanes
    uint32_t c = span_bytes <= (uint64_t(64) << 10) ? 128u : span_bytes <= (uint64_t(1) << 20) ? 256u : kLwLaneChunk;
    const uint32_t share = c < kLwLaneChunk ? 2u : 8u;   // (warm-up at most an eighth of the walk; half of it on small spans)
    while (c < share * warm && c < 8192) c *= 2;
    return c < share * warm ? 0 : c;
}

// scratch of the event form for this geometry: the slabs (one event per 16 haystack bytes and task), the tasks' event counts
// and the overflow word (zero it once, when the buffer is made)
// flush threshold of the wave queues for this image: what LDS leaves, in steps of 64, at most 448
uint32_t ev_queue_flush(uint32_t image_bytes) {
    const uint32_t q_off = (image_bytes + 63u) & ~63u;
    const uint32_t per_wave = (kLwLdsBytes - q_off) / kEvWaves / 16;   // events
    uint32_t f = per_wave > kEvSlack ? (per_wave - kEvSlack) / 64 * 64 : 0;
    return f > 448 ? 448 : f;
}

LwEvSizes lw_events_sizes(const ScanGeom& g) {
    LwEvSizes z;
    z.n_tasks = (g.n_chunks + 63) / 64;
    z.slab_events = kEvSlabPerChunk(g.chunk);
    z.ev_bytes = size_t(z.n_tasks) * z.slab_events * 16;
    z.task_n_bytes = size_t(z.n_tasks) * (3 * sizeof(uint32_t) + sizeof(uint64_t)) + 8;   // events | records | non-empty lane-chunks | (8-byte aligned) record offsets
    return z;
}

// (task_n: the per-task words of lw_events_sizes -- events, then records and non-empty lane-chunks (task_rec), then the record offsets)
namespace {
uint32_t* ev_task_rec(uint32_t* task_n, uint64_t n_tasks) { return task_n + n_tasks; }
uint64_t* ev_task_off(uint32_t* task_n, uint64_t n_tasks) {
    return reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(task_n) + ((n_tasks * 3 * sizeof(uint32_t) + 7) & ~size_t(7)));
}
}  // namespace

hipError_t launch_lw_count_ev(const HotTables& h, const ScanGeom& g, uint32_t* counts, void* events, uint32_t* task_n, uint32_t* overflow,
                              uint32_t gen, hipStream_t s) {
    if (!lw_fill_supported(h) || g.chunk % 128 != 0 || (g.n_chunks * uint64_t(g.chunk)) >> 2 > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const LwArgs la = ev_lw_args(h, g);
    if (la.n_tasks == 0) return hipSuccess;
    uint64_t blocks = uint64_t(device_cus());
    const uint64_t need = (la.n_tasks + kEvWaves - 1) / kEvWaves;
    if (blocks > need) blocks = need;
    LwEvArgs ea;
    ea.ev = static_cast<uint4*>(events); ea.task_n = task_n; ea.overflow = overflow; ea.gen = gen; ea.slab_events = kEvSlabPerChunk(g.chunk);
    ea.task_rec = ev_task_rec(task_n, la.n_tasks); ea.task_off = nullptr; ea.report = nullptr; ea.report_fail = 0;
    ea.q_off = (h.lw_image_bytes + 63u) & ~63u;
    ea.q_flush = ev_queue_flush(h.lw_image_bytes);
    if (ea.q_flush < kEvFlushMin) return hipErrorInvalidValue;
    const dim3 grid{uint32_t(blocks)}, block{kEvBlock};
    if (h.lw.computed_cls) k_lw_count_ev<true><<<grid, block, 0, s>>>(la, g, counts, ea);
    else k_lw_count_ev<false><<<grid, block, 0, s>>>(la, g, counts, ea);
    return hipGetLastError();
}

hipError_t launch_lw_task_scan(const ScanGeom& g, uint32_t* task_n, uint64_t* totals, uint64_t* host_totals, const uint32_t* extra32, hipStream_t s) {
    const uint64_t n_tasks = (g.n_chunks + 63) / 64;
    if (n_tasks == 0) return hipSuccess;
    k_lw_task_scan<<<dim3(1), dim3(kTsBlock), 0, s>>>(ev_task_rec(task_n, n_tasks), n_tasks, ev_task_off(task_n, n_tasks), totals, host_totals, extra32);
    return hipGetLastError();
}

hipError_t launch_lw_ev_emit(const HotTables& h, const ScanGeom& g, const void* events, const uint32_t* task_n, const uint32_t* overflow,
                             uint32_t gen, const uint32_t* counts, const uint64_t* totals, uint64_t cap, acgpu_match* out, hipStream_t s,
                             uint64_t* report, bool report_fail) {
    if (!lw_fill_supported(h)) return hipErrorInvalidValue;
    const LwArgs la = ev_lw_args(h, g);
    uint32_t shift = 0;
    while ((1u << shift) < g.chunk / 4) shift++;
    if ((1u << shift) != g.chunk / 4) return hipErrorInvalidValue;   // lane-chunks are powers of two
    if (la.n_tasks == 0) return hipSuccess;
    constexpr uint64_t waves = kEmBlock / 64;
    const uint64_t blocks = std::min<uint64_t>((la.n_tasks + waves - 1) / waves, 8 * uint64_t(device_cus()));
    const void* fn = h.lw.computed_cls ? reinterpret_cast<const void*>(k_lw_ev_emit<true>) : reinterpret_cast<const void*>(k_lw_ev_emit<false>);
    if (hipError_t e = ensure_dynamic_lds(fn, int(kLwLdsBytes)); e != hipSuccess) return e;
    LwEvArgs ea;
    ea.ev = const_cast<uint4*>(static_cast<const uint4*>(events)); ea.task_n = const_cast<uint32_t*>(task_n);
    ea.overflow = const_cast<uint32_t*>(overflow); ea.gen = gen; ea.slab_events = kEvSlabPerChunk(g.chunk);
    ea.task_rec = nullptr; ea.task_off = ev_task_off(const_cast<uint32_t*>(task_n), la.n_tasks);
    ea.report = report; ea.report_fail = report_fail ? 1u : 0u;
    ea.q_flush = 0; ea.q_off = 0;
    const dim3 grid{uint32_t(blocks)}, block{kEmBlock};
    const uint32_t lds = ((h.lw_image_bytes + 15u) & ~15u) + (kEmBlock / 64) * kEmWaveLds;
    if (h.lw.computed_cls) k_lw_ev_emit<true><<<grid, block, lds, s>>>(la, g, ea, counts, totals, cap, out, shift);
    else k_lw_ev_emit<false><<<grid, block, lds, s>>>(la, g, ea, counts, totals, cap, out, shift);
    return hipGetLastError();
}

}  // namespace acgpu
