// LDS-resident DFA transition walk (engine "hot", k_lw_count) for the Standard/unanchored overlapping scan (gfx950).
//
// The per-byte primitive is the reference's   sid = trans[sid + classes[byte]]   (src/dfa.rs:218-226) inside the
// overlapping loop (src/automaton.rs:1491-1534), one haystack lane-chunk per wavefront lane, with the WHOLE automaton
// held in LDS.  Three table flavours (host/lw_tables.cpp), one kernel skeleton:
//
// kLwFull -- every state owns a class-compressed row (automata whose rows fit 64 KiB: the reference's small-set
//   definitions).  The handle of a state is   byte address of its row << 16 | match-list length,   class values are
//   premultiplied by four, and a step is
//       h = LDS[h.hi16 + class4];  sum += h
//   i.e. the reference's premultiplied-id walk word for word (src/dfa.rs:218-226), the `is_match` test (src/dfa.rs:229-241)
//   and the length of the match list (src/dfa.rs:275-279) folded into the state id: the address is an SDWA word select, the
//   count is the low half of the plain sum of the handles (taken out once per dword: four lengths of at most 4 095) -- no
//   flag, no second path, the same speed whatever the match density (round 4's walk re-walked every dword that held a
//   match: 0.65 TB/s on the reference's teddy/same definitions against 3.3 TB/s on match-free input).
//
// kLwNarrow / kLwWide -- the "default row + exception" form SURVEY.md section 7 asks for, which is the failure-link idea
//   of the contiguous NFA (src/nfa/contiguous.rs:186-247) folded back into a DFA that needs ONE LDS gather per byte:
//   * dense states (the start state, the states at distance 1, then -- while LDS lasts -- the shallowest states that
//     differ from their nearest dense fail-ancestor in two or more columns) keep a full class-compressed row of handles;
//   * every other state t is described by its handle alone:  {base: row of D(t), e: exception class, da}  where D(t)
//     is the nearest dense state on t's failure chain.  row(t) equals row(D(t)) except in column e (for the 1k-pattern
//     headline set 99.5 % of the non-dense states differ in at most one column: a trie node with one child);
//     deep[idx] holds the handle of that one exceptional successor and da = 4 * idx is its LDS byte address as it stands
//     (deep[] is the first table: da < 64 KiB, read out of the handle by an SDWA word select);
//   * a step is   next = LDS[(class == h.e) ? h.da : h.base * row_bytes + 4 * class_value]   -- the two candidate addresses
//     side by side and a select: 4 VALU + one ds_read_b32, uniform over the wave whatever mix of states its lanes are in
//     (class_value = class + the dword offset of row 0, so the row address needs no further add);
//   * states with k >= 2 exceptions that got no row ("multi") own k consecutive "virtual" slots behind the real states:
//     deep[slot j] = successor under exception j, nxt[slot j] = handle that tests exception j+1 (the last one falls
//     back to D's row).  The fast path does not test for them: a multi state's base is the POISON row, whose entries
//     are a self-perpetuating poison handle numbered above everything.  Match states are numbered last among the real
//     states (is_match <=> da >= fm_addr, the reference's `sid <= max_special_id` trick reversed) and virtual slots above
//     them, so ONE max per byte (v_max3_u16: two per dword) tells "match, multi or poison in these 4 bytes".  If it was
//     matches only (da < virt_addr) the four handles are exact and each matching byte costs one u16 gather of its
//     match-list length; only a multi state sends the lane back over its 4 bytes by the exact step -- entirely from LDS:
//     a global load on that path would make the compiler drain the haystack prefetch (s_waitcnt vmcnt(0)) at every dword.
//
// Class of a byte: read from a u16 map in LDS (one more gather per byte, conflict-free: 2 LDS cycles), or COMPUTED when
// the map is a clamp of the byte onto the range the patterns use -- the reference's ByteClasses are monotone step
// functions (src/util/alphabet.rs:235-250) and for sets whose classes hold one byte each that step function is
// med3(byte + add, lo, hi): v_add_u32_sdwa + v_med3_i32 and no LDS access.
//
// Haystack access: lane-chunks are 512 B (1 KiB on shards of 6 GiB and more; sub-divisions of the scan's count chunks),
// so a wavefront covers one contiguous 32 KiB region and a persistent workgroup of 16 waves 512 KiB at a time; each
// lane streams its chunk in whole 128-byte lines, double-buffered in registers (eight 16-byte loads back to back: the
// line is fetched once), the prefetch behind the last line of a chunk fetches the first line of the wavefront's next
// task; warm-up = the max_pattern_len-1 bytes before the chunk rounded up to 16.  No LDS staging: all of LDS belongs
// to the automaton.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../host/lw_tables.hpp"
#include "hot.hpp"
#include "launch_util.hpp"
#include "lw_dev.hpp"

namespace acgpu {

namespace {

using namespace lwdev;

constexpr int kLwBlock = 1024;
constexpr int kLwWaves = kLwBlock / 64;

// UP = 16-byte pieces per unit: 8 = one 128-byte cache line per visit (every line is fetched once), 4 = 64-byte units.
template <int UP, int FLAV, bool CC>
__global__ __launch_bounds__(kLwBlock) void k_lw_count(LwArgs a, ScanGeom g, uint32_t* __restrict__ counts) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLwLdsBytes];   // static, at LDS address 0: no base add per lookup
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.image);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (uint32_t i = threadIdx.x; i < a.image_bytes / 16; i += kLwBlock) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_id = uint64_t(blockIdx.x) * kLwWaves + wave;
    const uint64_t n_waves = uint64_t(gridDim.x) * kLwWaves;
    const uint32_t C = a.lane_chunk;
    const uint32_t warm_bytes = a.warm_pieces * 16;
    const uint32_t n_main = C / 16;   // pieces of an owned lane-chunk (multiple of 4)
    const LwLds L{lds};
    LwCc cc;
    cc.add = uint32_t(a.cc_add); cc.hi = uint32_t(a.cc_hi);
    asm volatile("v_mov_b32 %0, %1" : "=v"(cc.v_lo) : "s"(a.cc_lo));   // opaque: the compiler must keep it in a VGPR

    const uint64_t region_bytes = uint64_t(64) * C;
    auto is_interior = [&](uint64_t lo) {
        return lo >= g.emit_lo && lo + region_bytes <= g.emit_hi && lo >= g.cold_floor + warm_bytes;
    };
    uint4 ua[UP], ub[UP];
    bool have_ua = false;   // ua holds unit 0 of this task: the previous task of the wave loaded it under its last unit
    for (uint64_t task = wave_id; task < a.n_tasks; task += n_waves) {
        const uint64_t j0 = task * 64;                           // first lane-chunk of the wave
        const uint64_t region_lo = g.grid0 + j0 * C;
        const bool interior = is_interior(region_lo);
        const uint64_t next_lo = region_lo + n_waves * region_bytes;
        const bool next_interior = task + n_waves < a.n_tasks && is_interior(next_lo);
        uint32_t cnt = 0;
        if (interior) {
            const uint8_t* p_main = g.hay16 + region_lo + uint64_t(lane) * C;
            uint32_t h = a.start;
            auto ld = [&](const uint8_t* p) {
                ACGPU_HAY_CHECK(g, uint64_t(p - g.hay16), 16);
                return *reinterpret_cast<const uint4*>(p);
            };
            // The chunk is consumed in units of one 128-byte cache line (UP = 8 pieces), double-buffered in registers: the
            // 16-byte loads of a unit are issued back to back, so every line is requested ONCE (the other loads merge into
            // the pending miss) and never re-fetched.  Measured (profiles/r02_*): with a sliding 16-byte window each lane
            // came back to its line eight times, microseconds apart, and the 1024 open lines per CU did not survive in L2
            // between visits (2.08 TB/s); 64-byte units still fetch every line 1.8 times (3.2 TB/s, 15.6 GB of fabric reads).
            auto ld_unit_at = [&](uint4 (&u)[UP], const uint8_t* p) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < UP; k++) u[k] = ld(p + 16 * k);
            };
            auto piece = [&](const uint4& q, auto owned) __attribute__((always_inline)) {
                constexpr bool OW = decltype(owned)::value;
                lw_step4<FLAV, CC, OW>(a, L, cc, q.x, h, cnt);
                lw_step4<FLAV, CC, OW>(a, L, cc, q.y, h, cnt);
                lw_step4<FLAV, CC, OW>(a, L, cc, q.z, h, cnt);
                lw_step4<FLAV, CC, OW>(a, L, cc, q.w, h, cnt);
            };
            auto do_unit = [&](const uint4 (&u)[UP]) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < UP; k++) piece(u[k], std::true_type{});
            };
            const uint32_t n_units = n_main / UP;
            // warm-up pieces (processed first) and unit 0 in flight together
            {
                uint4 wq = make_uint4(0, 0, 0, 0);
                if (a.warm_pieces) wq = ld(p_main - 16 * a.warm_pieces);
                if (!have_ua) ld_unit_at(ua, p_main);
                for (uint32_t wp = a.warm_pieces; wp > 0; wp--) {
                    piece(wq, std::false_type{});
                    if (wp > 1) wq = ld(p_main - 16 * (wp - 1));
                }
            }
#pragma unroll 1
            for (uint32_t u0 = 0; u0 < n_units; u0 += 2) {
                // unconditional prefetches: a conditional load would force the compiler to s_waitcnt vmcnt(0) in front of
                // every use.  Behind the last unit of the chunk the prefetch fetches unit 0 of the wave's NEXT task (round 2
                // re-loaded the last unit there: one line of every four fetched twice, profiles/r02_hot_pmc.json 1.47x).
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
            if (j < a.n_lane_chunks) {
                const uint64_t glo = g.grid0 + j * C, ghi = glo + C;
                const uint64_t lo = glo > g.emit_lo ? glo : g.emit_lo;
                const uint64_t hi = ghi < g.emit_hi ? ghi : g.emit_hi;
                if (j == 0 && g.emit_start_matches) cnt += lw_match_len<FLAV>(a, L, a.start);   // empty patterns at span_start
                if (hi > lo) {
                    uint64_t w = lo >= g.halo ? lo - g.halo : 0;
                    if (w < g.cold_floor) w = g.cold_floor;
                    cnt = lw_edge_walk<FLAV, CC>(a, L, g, w, lo, hi, cnt);
                }
            }
        }
        // sum the lane-chunks of each count chunk (lanes_per_chunk consecutive lanes)
        {
            const uint64_t j = j0 + uint64_t(lane);
            uint32_t c = cnt;
            for (uint32_t o = 1; o < a.lanes_per_chunk; o <<= 1) c += __shfl_xor(c, int(o), 64);
            if ((uint32_t(lane) & (a.lanes_per_chunk - 1)) == 0 && j < a.n_lane_chunks) counts[j / a.lanes_per_chunk] = c;
        }
    }
}

// ---- record fill for the one-row-per-state form (k_lw_fill): the ordered match records of the non-empty count chunks.
// Same contract as k_hot_fill / k_walk_fill (one wavefront per non-empty chunk, its 64 lanes own consecutive sub-ranges, a
// counting walk, a prefix sum over the lanes, an emitting walk), but nothing of the walk leaves LDS: transitions, match-list
// lengths (in the handles) and the lists themselves {pattern id, pattern length} (src/dfa.rs:275-279) come from the image;
// k_hot_fill paid three dependent global gathers per match state and one per record (match list offset, pattern id,
// pattern length), 50 G records/s on the reference's match-dense definitions.  The haystack bytes of a sub-range are read
// in 16-byte pieces, two ahead.
template <bool CC, bool EMIT>
__device__ __forceinline__ uint32_t lw_range_walk(const LwArgs& a, const LwLds& L, const ScanGeom& g, uint64_t w, uint64_t lo, uint64_t hi,
                                                  bool start_matches, acgpu_match* dst) {
    uint32_t n = 0;
    auto emit = [&](uint32_t h, uint64_t end) {   // the records of the state behind h, all ending at haystack offset `end`
        const uint32_t len = h & kLwFullLenMask;
        if (!EMIT) { n += len; return; }
        if (len == 0) return;
        const uint32_t list = L.rd32((h >> 16) + 4 * a.list_col);
        for (uint32_t i = 0; i < len; i++) {
            const uint32_t pid = L.rd32(list + 8 * i), plen = L.rd32(list + 8 * i + 4);
            const uint64_t start = end - plen;
            uint32_t* p = reinterpret_cast<uint32_t*>(dst + n + i);   // acgpu_match: {u32 pattern, u32 pad, u64 start, u64 end}
            *reinterpret_cast<uint4*>(p) = make_uint4(pid, 0u, uint32_t(start), uint32_t(start >> 32));
            *reinterpret_cast<uint2*>(p + 4) = make_uint2(uint32_t(end), uint32_t(end >> 32));
        }
        n += len;
    };
    uint32_t h = a.start;
    if (start_matches) emit(h, g.cold_floor - g.base_mis);   // empty patterns at span_start
    if (hi <= lo) return n;
    const uint8_t* hay16 = g.hay16;
    const uint64_t p0 = w & ~uint64_t(15);
    auto ld = [&](uint64_t p) {
        if (p < hi) ACGPU_HAY_CHECK(g, p, 16);
        return p < hi ? *reinterpret_cast<const uint4*>(hay16 + p) : make_uint4(0, 0, 0, 0);
    };
    uint4 q0 = ld(p0), q1 = ld(p0 + 16);
    for (uint64_t p = p0; p < hi; p += 16) {
        const uint4 q = q0;
        q0 = q1;
        q1 = ld(p + 32);
        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t v = p + k;
            if (v >= w && v < hi) {
                h = L.rd32((h >> 16) + lw_clsval_byte<CC, true>(L, a, (wd[k >> 2] >> (8 * (k & 3))) & 0xFFu));
                if (v >= lo) emit(h, v + 1 - g.base_mis);
            }
        }
    }
    return n;
}

constexpr int kLfBlock = 512;   // 8 wavefronts: the image is small here and two workgroups share a CU's LDS when it is below 80 KiB
template <bool CC>
__global__ __launch_bounds__(kLfBlock) void k_lw_fill(LwArgs a, ScanGeom g, const uint64_t* __restrict__ active,
                                                      const uint64_t* __restrict__ totals, uint64_t cap, const uint64_t* __restrict__ aoff,
                                                      acgpu_match* __restrict__ out, const uint32_t* __restrict__ ev_overflow, uint32_t gen,
                                                      const uint64_t* __restrict__ fine_off, uint32_t stride, uint64_t n_fine) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_dyn[];
    // (fine_off: no list of active chunks -- every chunk of g, its record offset that of its first fine chunk: the event form's
    // scan runs over lane-chunks of a quarter of the fill's chunk)
    const uint64_t n_active = fine_off ? g.n_chunks : totals[1];
    if (totals[0] > cap || uint64_t(blockIdx.x) * (kLfBlock / 64) >= n_active) return;
    if (ev_overflow && *ev_overflow != gen) return;   // the events of the count walk serve (lds_emit.hip)
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.image);
        uint4* dst = reinterpret_cast<uint4*>(lds_dyn);
        for (uint32_t i = threadIdx.x; i < a.image_bytes / 16; i += kLfBlock) dst[i] = src[i];
    }
    __syncthreads();
    const LwLds L{lds_dyn};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t ai = uint64_t(blockIdx.x) * (kLfBlock / 64) + wave; ai < n_active; ai += uint64_t(gridDim.x) * (kLfBlock / 64)) {
        const uint64_t ci = fine_off ? ai : active[ai];
        uint64_t rec0 = 0;
        if (fine_off) {
            rec0 = fine_off[ci * stride];
            const uint64_t rec1 = (ci + 1) * stride < n_fine ? fine_off[(ci + 1) * stride] : totals[0];
            if (rec1 == rec0) continue;   // (wave-uniform)
        } else rec0 = aoff[ai];
        const ChunkRange r = chunk_range(g, ci);
        // split [r.lo, r.hi) into 64 sub-ranges of `sub` bytes (the last ones may be empty)
        const uint64_t len = r.hi - r.lo;
        const uint64_t sub = (len + 63) / 64;
        uint64_t lo = r.lo + uint64_t(lane) * sub, hi = lo + sub;
        if (lo > r.hi) lo = r.hi;
        if (hi > r.hi) hi = r.hi;
        uint64_t w = lo >= g.halo ? lo - g.halo : 0;
        if (w < g.cold_floor) w = g.cold_floor;
        const bool sm = ci == 0 && lane == 0 && g.emit_start_matches;
        const uint32_t c = lw_range_walk<CC, false>(a, L, g, w, lo, hi, sm, nullptr);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (c) (void)lw_range_walk<CC, true>(a, L, g, w, lo, hi, sm, out + rec0 + (incl - c));
    }
}

template <int UP, int FLAV>
void lw_launch_cls(bool cc, dim3 grid, hipStream_t s, const LwArgs& la, const ScanGeom& g, uint32_t* counts) {
    const dim3 block{kLwBlock};
    if (cc) k_lw_count<UP, FLAV, true><<<grid, block, 0, s>>>(la, g, counts);
    else k_lw_count<UP, FLAV, false><<<grid, block, 0, s>>>(la, g, counts);
}
template <int UP>
void lw_launch(uint32_t flavour, bool cc, dim3 grid, hipStream_t s, const LwArgs& la, const ScanGeom& g, uint32_t* counts) {
    if (flavour == kLwFull) lw_launch_cls<UP, kLwFull>(cc, grid, s, la, g, counts);
    else if (flavour == kLwWide) lw_launch_cls<UP, kLwWide>(cc, grid, s, la, g, counts);
    else lw_launch_cls<UP, kLwNarrow>(cc, grid, s, la, g, counts);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host: tables
// `order` = hid -> nnfa sid, `sid2hid` its inverse (hid_order); the tables themselves: host/lw_tables.cpp.
hipError_t build_lw_tables(const NNfa& n, const Dfa& d, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid,
                           uint32_t first_match, HotTables& out) {
    out.lw_ready = false;
    LwHostTables t;
    // variants lw_flavour / lw_cls (host/variants.hpp): the tables are refused when the automaton does not fit a forced form
    if (!build_lw_host(n, d, order, sid2hid, first_match, t, out.var.lw_flavour, out.var.lw_cls)) return hipSuccess;
    const uint32_t image_bytes = uint32_t(t.image.size() * 4);
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.lw_image), image_bytes)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.lw_image, t.image.data(), image_bytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.lw_image_bytes = image_bytes;
    {
        // What the walk costs in the prefix filter's routing rule (pf_scan.hip: 5000 X + E M > cb B + cr min(B, 256 M)).
        // A dword on the exact path holds up its whole wavefront: p = share of wave-dwords with at least one such lane;
        // measured, the walk then runs at 3 200 / (1 + 3.9 p) GB/s (headline set: p ~ 0; 1 000 a-z patterns: 2 % of the
        // dwords meet a chained state, p = 0.73, 830 GB/s).  cb = 1.25 * 650 000 / rate - 130 (hot.hpp: 124 at 3 200 GB/s).
        const double r = lw_estimate_redo(t);
        const double p = 1.0 - std::pow(1.0 - std::min(r, 1.0), 64.0);
        const double rate = 3200.0 / (1.0 + 3.9 * p);
        out.lw_route_cb = uint32_t(std::max(124.0, 812500.0 / rate - 130.0));
    }
    t.image.clear();
    t.image.shrink_to_fit();
    out.lw = t;   // offsets, flavour, class form (the image itself lives on the device)
    out.lw_ready = true;
    return hipSuccess;
}

namespace {
LwArgs lw_args(const HotTables& h) {
    const LwHostTables& t = h.lw;
    LwArgs la{};
    la.image = h.lw_image;
    la.image_bytes = h.lw_image_bytes; la.row_bytes = t.row_bytes;
    la.fm_addr = t.fm_addr; la.virt_addr = t.virt_addr;
    la.nxt_off = t.nxt_off; la.vhid_off = t.vhid_off; la.mlen_off = t.mlen_off;
    la.poison_base = t.poison_row; la.start = t.start;
    la.cc_add = t.cc_add; la.cc_lo = t.cc_lo; la.cc_hi = t.cc_hi;
    la.list_col = t.classes;
    return la;
}
}  // namespace

// The record fill of the one-row-per-state form: available when the match lists fit LDS beside the rows.
bool lw_fill_supported(const HotTables& h) { return h.lw_ready && h.lw.flavour == kLwFull && h.lw.mlist_off != 0; }

hipError_t launch_lw_fill(const HotTables& h, const ScanGeom& g, const uint64_t* active, const uint64_t* totals, uint64_t cap,
                          uint64_t max_waves, const uint64_t* aoff, acgpu_match* out, hipStream_t s,
                          const uint32_t* ev_overflow, uint32_t gen, const uint64_t* fine_off, uint32_t stride, uint64_t n_fine) {
    if (!lw_fill_supported(h)) return hipErrorInvalidValue;
    const LwArgs la = lw_args(h);
    uint64_t waves = max_waves < g.n_chunks ? max_waves : g.n_chunks;
    uint64_t blocks = (waves + kLfBlock / 64 - 1) / (kLfBlock / 64);
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const void* fn = h.lw.computed_cls ? reinterpret_cast<const void*>(k_lw_fill<true>) : reinterpret_cast<const void*>(k_lw_fill<false>);
    if (hipError_t e = ensure_dynamic_lds(fn, int(kLwLdsBytes)); e != hipSuccess) return e;
    const dim3 grid{uint32_t(blocks)}, block{kLfBlock};
    if (h.lw.computed_cls) k_lw_fill<true><<<grid, block, h.lw_image_bytes, s>>>(la, g, active, totals, cap, aoff, out, ev_overflow, gen, fine_off, stride, n_fine);
    else k_lw_fill<false><<<grid, block, h.lw_image_bytes, s>>>(la, g, active, totals, cap, aoff, out, ev_overflow, gen, fine_off, stride, n_fine);
    return hipGetLastError();
}

hipError_t launch_hot_count(const HotTables& h, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts, hipStream_t s) {
    if (!h.lw_ready) return hipErrorInvalidValue;
    const LwHostTables& t = h.lw;
    LwArgs la = lw_args(h);
    // lane-chunks: the count chunk split into a power-of-two number of pieces of >= kLwLaneChunk bytes (multiples of 64)
    // 512-byte lane-chunks by default; on the largest shards (from 6 GiB on) 1 024-byte ones halve the share of the
    // warm-up line (fabric reads 1.25x -> 1.13x the haystack, +2 % at 8 GiB: profiles/r03_hot_pmc.json, r03_hot_ab.jsonl).
    // Below that they lose to the coarser task grain: 1 GiB 0.41 vs 0.49 ms, 2 GiB 0.71 vs 0.76 ms, 4 GiB 1.33 vs 1.32 ms.
    const uint32_t target = h.var.lw_lane_chunk > 0 ? uint32_t(h.var.lw_lane_chunk) : (g.emit_hi - g.emit_lo >= (uint64_t(6) << 30) ? 2 * kLwLaneChunk : kLwLaneChunk);
    const int up = g.chunk % 128 == 0 ? 8 : 4;   // whole cache lines when the chunk grid allows
    uint32_t m = 1;
    const uint32_t want = std::max<uint32_t>(target, (8 * g.halo + 63) & ~63u);   // warm-up <= 1/8 of the walk
    const uint32_t unit = 16u * uint32_t(up);   // lane-chunks are whole units
    while (m < 64 && g.chunk % (2 * m * unit) == 0 && g.chunk / (2 * m) >= want) m *= 2;
    la.lanes_per_chunk = m;
    la.lane_chunk = g.chunk / m;
    la.warm_pieces = (g.halo + 15) / 16;
    la.n_lane_chunks = g.n_chunks * m;
    la.n_tasks = (la.n_lane_chunks + 63) / 64;
    if (la.n_tasks == 0) return hipSuccess;
    if (h.lw_image_bytes > kLwLdsBytes) return hipErrorInvalidValue;
    uint64_t blocks = uint64_t(device_cus());
    const uint64_t need = (la.n_tasks + kLwWaves - 1) / kLwWaves;
    if (blocks > need) blocks = need;
    const dim3 grid{uint32_t(blocks)};
    if (up == 4) lw_launch<4>(t.flavour, t.computed_cls, grid, s, la, g, counts);
    else lw_launch<8>(t.flavour, t.computed_cls, grid, s, la, g, counts);
    return hipGetLastError();
}

}  // namespace acgpu
