// LDS-resident DFA transition walk (engine "hot", k_lw_count) for the Standard/unanchored overlapping scan (gfx950).
//
// The per-byte primitive is the reference's   sid = trans[sid + classes[byte]]   (src/dfa.rs:218-226) inside the
// overlapping loop (src/automaton.rs:1491-1534), one haystack lane-chunk per wavefront lane, with the WHOLE automaton
// held in LDS in the "default row + exception" form SURVEY.md section 7 asks for -- which is the failure-link idea of
// the contiguous NFA (src/nfa/contiguous.rs:186-247) folded back into a DFA that needs ONE LDS gather per byte:
//
//   * dense states (the start state, the states at distance 1, then -- while LDS lasts -- the shallowest states that
//     differ from their nearest dense fail-ancestor in two or more columns) keep a full class-compressed row of
//     32-bit "handles";
//   * every other state t is described by its handle alone:  {base: row of D(t), e: exception class, idx}  where D(t)
//     is the nearest dense state on t's failure chain.  row(t) equals row(D(t)) except in column e (for the 1k-pattern
//     headline set 99.5 % of the non-dense states differ in at most one column: a trie node with one child);
//     deep[idx] holds the handle of that one exceptional successor;
//   * a step is   next = (class == h.e) ? deep[h.idx] : rows[h.base][class]   -- the two candidate LDS addresses are
//     computed side by side and selected, so the dependent chain per byte is 3 VALU ops + one ds_read_b32, uniform over
//     the wave whatever mix of dense / non-dense states its lanes are in;
//   * states with k >= 2 exceptions that got no row ("multi") own k consecutive "virtual" slots behind the real states:
//     deep[slot j] = successor under exception j, nxt[slot j] = handle that tests exception j+1 (the last one falls
//     back to D's row).  The fast path does not test for them: a multi state's base is the POISON row, whose entries
//     are a self-perpetuating poison handle numbered above everything.  Match states are numbered last among the real
//     states (is_match <=> idx >= first_match, the reference's `sid <= max_special_id` trick, src/dfa.rs:229-241,
//     reversed) and virtual slots above them, so ONE compare per byte (folded per dword with v_max3) tells "match, multi
//     or poison in these 4 bytes", and only then the 4 bytes are re-walked from the saved handle by the exact step, which
//     resolves chains and counts matches (match-list lengths as u16 in LDS) -- entirely from LDS: a global load on
//     that path would make the compiler drain the haystack prefetch (s_waitcnt vmcnt(0)) at every dword.
//
// The class map is the engine's own: bytes whose DFA columns are identical share a class (coarser than the
// reference's ByteClasses, src/util/alphabet.rs:224-250, e.g. both cases of a letter under ascii_case_insensitive),
// which keeps one trie edge = one exception.
//
// Haystack access: lane-chunks are 512 B (1 KiB on shards of 6 GiB and more; sub-divisions of the scan's count chunks),
// so a wavefront covers one contiguous 32 KiB region and a persistent workgroup of 16 waves 512 KiB at a time; each
// lane streams its chunk in whole 128-byte lines, double-buffered in registers (eight 16-byte loads back to back: the
// line is fetched once), the prefetch behind the last line of a chunk fetches the first line of the wavefront's next
// task; warm-up = the max_pattern_len-1 bytes before the chunk rounded up to 16.  No LDS staging: all of LDS belongs
// to the automaton.
//
// Handle layouts (host/lw_tables.cpp): base 8 | e 8 | idx 16 bits (SDWA byte selects, 4 VALU for the address), or -- for
// alphabets of at most 64 classes whose states want more than 254 rows -- base 10 | e 6 | idx 16 (WIDE, 6 VALU).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../host/lw_tables.hpp"
#include "hot.hpp"
#include "launch_util.hpp"

namespace acgpu {

namespace {

constexpr int kLwBlock = 1024;
constexpr int kLwWaves = kLwBlock / 64;
constexpr uint32_t kLwLdsBytes = kLwLdsBudget;
constexpr uint32_t kLwLaneChunk = 512;   // target bytes per lane-chunk

constexpr uint32_t kLwCls = kLwClsBytes;   // the class map occupies LDS bytes [0, 256); table addresses are relative to 256

struct LwArgs {
    const uint32_t* image;     // LDS image: class map | rows | deep | nxt | vhid | mlen
    uint32_t image_bytes;
    uint32_t row_bytes;        // bytes per row (an odd number of dwords: host/lw_tables.cpp)
    uint32_t deep_off;         // byte offset of deep[] behind the rows (relative to kLwCls)
    uint32_t fm_addr;          // deep_off + 4 * first_match: handles whose deep address is >= this are match / multi / poison
    uint32_t nxt_off;          // u32 [n_virtual]: next handle of an exception chain
    uint32_t vhid_off;         // u16 [n_virtual]: the real state behind the first slot of a multi state
    uint32_t mlen_off;         // u16 [n_states - first_match]: match-list lengths
    uint32_t poison_base;      // row index of the poison row
    uint32_t start;            // handle of the unanchored start state
    uint32_t first_match, n_states;
    uint32_t base_shift, e_mask;   // handle layout: base = h >> base_shift, e = (h >> 16) & e_mask (24 / 0xFF, wide: 22 / 0x3F)
    // lane-chunk geometry (sub-division of the scan's count chunks)
    uint32_t lane_chunk;       // bytes per lane-chunk (multiple of 64)
    uint32_t lanes_per_chunk;  // power of two <= 64: lane-chunks per count chunk
    uint32_t warm_pieces;      // ceil(halo / 16)
    uint64_t n_lane_chunks, n_tasks;
};

struct LwLds {
    const uint8_t* base;   // LDS byte 0 of the image
    __device__ __forceinline__ uint32_t cls(uint32_t byte) const { return base[byte]; }
    __device__ __forceinline__ uint32_t rd32(uint32_t table_addr) const {   // the constant lands in the DS offset field
        return *reinterpret_cast<const uint32_t*>(base + kLwCls + table_addr);
    }
    __device__ __forceinline__ uint32_t rd16(uint32_t table_addr) const {
        return *reinterpret_cast<const uint16_t*>(base + kLwCls + table_addr);
    }
};

// The exact step (any state kind): resolves exception chains.  Rare path; LDS only.
__device__ __forceinline__ uint32_t lw_careful_step(const LwArgs& a, const LwLds& L, uint32_t h, uint32_t byte) {
    const uint32_t c = L.cls(byte);
    for (int hop = 0; hop < 4096; hop++) {
        const uint32_t idx = h & 0xFFFFu;
        if (((h >> 16) & a.e_mask) == c) return L.rd32(a.deep_off + idx * 4);
        const uint32_t b = h >> a.base_shift;
        if (b != a.poison_base) return L.rd32(b * a.row_bytes + c * 4);
        h = L.rd32(a.nxt_off + (idx - a.n_states) * 4);   // multi state / chain link: idx is a virtual slot
    }
    return h;
}

// Number of matches of the state behind handle h (src/dfa.rs:275-279: the length of its match list).
__device__ __forceinline__ uint32_t lw_match_len(const LwArgs& a, const LwLds& L, uint32_t h) {
    uint32_t idx = h & 0xFFFFu;
    if (idx >= a.n_states) idx = L.rd16(a.vhid_off + (idx - a.n_states) * 2);   // first slot of a multi state
    return idx >= a.first_match ? L.rd16(a.mlen_off + (idx - a.first_match) * 2) : 0u;
}

// Re-walk of one dword from the saved handle, for the lanes whose fast walk met a match state or poison.
__device__ __forceinline__ uint32_t lw_redo4(const LwArgs& a, const LwLds& L, uint32_t h, uint32_t w, bool owned, uint32_t& cnt) {
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        h = lw_careful_step(a, L, h, (w >> (8 * k)) & 0xFFu);
        if (owned) cnt += lw_match_len(a, L, h);
    }
    return h;
}

// Generic (edge) walk of one lane-chunk -- the first and last wave regions of a shard, and every chunk of a small
// input: exact step, ownership from `lo`.  The bytes come in aligned 16-byte pieces (only pieces holding a live byte are
// touched), two pieces ahead: a dependent global load per byte cost ~0.5 us each, 0.25 ms for one 512-byte chunk.
__device__ __forceinline__ uint32_t lw_edge_walk(const LwArgs& a, const LwLds& L, const ScanGeom& g, uint64_t w, uint64_t lo,
                                                 uint64_t hi, uint32_t cnt) {
    const uint8_t* hay16 = g.hay16;
    uint32_t h = a.start;
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
                h = lw_careful_step(a, L, h, (wd[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                if (v >= lo) cnt += lw_match_len(a, L, h);
            }
        }
    }
    return cnt;
}

// ---- the fast step, hand-scheduled (gfx950).  State of a chain: handle h and da = LDS address of deep[h.idx].
//   a  = (class == h.e) ? da : rows + h.base * row_bytes + 4 * class          -- lw_addr: 4 VALU
//   h' = LDS[a]                                                                  -- ds_read_b32 (offset = kLwCls)
//   da' = deep_off + 4 * h'.idx                                                  -- lw_deep: 1 VALU (v_mad_u32_u16)
// The compare writes VCC and the select reads it two instructions later (the wait states gfx950 needs between a VALU
// write of VCC and a VALU read of it); SDWA operand selects pick h.e / h.base without separate shifts.
__device__ __forceinline__ uint32_t lw_addr(uint32_t h, uint32_t da, uint32_t c, uint32_t row_bytes) {
    uint32_t a, t;
    asm("v_cmp_eq_u32_sdwa vcc, %2, %4 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
        "v_mul_u32_u24_sdwa %1, %5, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "v_lshl_add_u32 %1, %4, 2, %1\n\t"
        "v_cndmask_b32_e32 %0, %1, %3, vcc"
        : "=&v"(a), "=&v"(t)
        : "v"(h), "v"(da), "v"(c), "s"(row_bytes)
        : "vcc");
    return a;
}
// The wide-base layout (base 10 | e 6 | idx 16 bits; small alphabets with more than 254 rows): no byte selects, 6 VALU.
__device__ __forceinline__ uint32_t lw_addr_wide(uint32_t h, uint32_t da, uint32_t c, uint32_t row_bytes) {
    const uint32_t e = __builtin_amdgcn_ubfe(h, 16, 6);
    const uint32_t ra = __umul24(h >> 22, row_bytes) + (c << 2);
    return e == c ? da : ra;
}
__device__ __forceinline__ uint32_t lw_deep(uint32_t h, uint32_t deep_off) {   // deep_off + 4 * (h & 0xFFFF)
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(r) : "v"(h), "s"(deep_off));
    return r;
}

// NCH independent chains per lane (lane-chunks j0 + lane + 64 i): instruction-level parallelism on top of the
// wave-level one, so that the LDS latency of one chain's lookup is covered by the other chain's address arithmetic.
// UP = 16-byte pieces per unit: 8 = one 128-byte cache line per visit (every line is fetched once), 4 = 64-byte units.
template <int NCH, int UP, bool WIDE>
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
    const uint32_t row_bytes = a.row_bytes, deep_off = a.deep_off, fm_addr = a.fm_addr;
    const LwLds L{lds};
    const uint32_t da_start = deep_off + ((a.start & 0xFFFFu) << 2);

    const uint64_t region_bytes = uint64_t(64 * NCH) * C;
    auto is_interior = [&](uint64_t lo) {
        return lo >= g.emit_lo && lo + region_bytes <= g.emit_hi && lo >= g.cold_floor + warm_bytes;
    };
    uint4 ua[UP][NCH], ub[UP][NCH];
    bool have_ua = false;   // ua holds unit 0 of this task: the previous task of the wave loaded it under its last unit
    for (uint64_t task = wave_id; task < a.n_tasks; task += n_waves) {
        const uint64_t j0 = task * (64 * NCH);                   // first lane-chunk of the wave
        const uint64_t region_lo = g.grid0 + j0 * C;
        const bool interior = is_interior(region_lo);
        const uint64_t next_lo = region_lo + n_waves * region_bytes;
        const bool next_interior = task + n_waves < a.n_tasks && is_interior(next_lo);
        uint32_t cnt[NCH];
#pragma unroll
        for (int i = 0; i < NCH; i++) cnt[i] = 0;
        if (interior) {
            const uint8_t* p_main[NCH];
            uint32_t h[NCH], da[NCH];
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                p_main[i] = g.hay16 + region_lo + (uint64_t(lane) + 64 * i) * C;
                h[i] = a.start; da[i] = da_start;
            }
            // one dword per chain = 4 steps on the fast path; flagged lanes (match or poison met) redo theirs exactly
            auto step4 = [&](const uint32_t (&w)[NCH], bool owned) {
                uint32_t h0[NCH], worst[NCH];
#pragma unroll
                for (int i = 0; i < NCH; i++) { h0[i] = h[i]; worst[i] = 0; }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t c[NCH];
#pragma unroll
                    for (int i = 0; i < NCH; i++) c[i] = L.cls(__builtin_amdgcn_ubfe(w[i], 8 * k, 8));
#pragma unroll
                    for (int i = 0; i < NCH; i++) {
                        h[i] = L.rd32(WIDE ? lw_addr_wide(h[i], da[i], c[i], row_bytes) : lw_addr(h[i], da[i], c[i], row_bytes));
                        da[i] = lw_deep(h[i], deep_off);
                        worst[i] = worst[i] > da[i] ? worst[i] : da[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < NCH; i++) {
                    const bool flag = worst[i] >= fm_addr;
                    if (__builtin_expect(__any(flag), 0)) {
                        if (flag) { h[i] = lw_redo4(a, L, h0[i], w[i], owned, cnt[i]); da[i] = deep_off + ((h[i] & 0xFFFFu) << 2); }
                    }
                }
            };
            auto piece = [&](const uint4 (&q)[NCH], bool owned) {
                uint32_t w[NCH];
#pragma unroll
                for (int i = 0; i < NCH; i++) w[i] = q[i].x;
                step4(w, owned);
#pragma unroll
                for (int i = 0; i < NCH; i++) w[i] = q[i].y;
                step4(w, owned);
#pragma unroll
                for (int i = 0; i < NCH; i++) w[i] = q[i].z;
                step4(w, owned);
#pragma unroll
                for (int i = 0; i < NCH; i++) w[i] = q[i].w;
                step4(w, owned);
            };
            auto ld = [&](const uint8_t* p) {
                ACGPU_HAY_CHECK(g, uint64_t(p - g.hay16), 16);
                return *reinterpret_cast<const uint4*>(p);
            };

            // The chunk is consumed in units of one 128-byte cache line (UP = 8 pieces), double-buffered in registers: the
            // 16-byte loads of a unit are issued back to back, so every line is requested ONCE (the other loads merge into
            // the pending miss) and never re-fetched.  Measured (profiles/r02_*): with a sliding 16-byte window each lane
            // came back to its line eight times, microseconds apart, and the 1024 open lines per CU did not survive in L2
            // between visits (2.08 TB/s); 64-byte units still fetch every line 1.8 times (3.2 TB/s, 15.6 GB of fabric reads).
            auto ld_unit_at = [&](uint4 (&u)[UP][NCH], const uint8_t* const (&p)[NCH]) {
#pragma unroll
                for (int k = 0; k < UP; k++)
#pragma unroll
                    for (int i = 0; i < NCH; i++) u[k][i] = ld(p[i] + 16 * k);
            };
            auto ld_unit = [&](uint4 (&u)[UP][NCH], uint32_t unit) {
                const uint8_t* p[NCH];
#pragma unroll
                for (int i = 0; i < NCH; i++) p[i] = p_main[i] + (16 * UP) * unit;
                ld_unit_at(u, p);
            };
            auto do_unit = [&](const uint4 (&u)[UP][NCH]) {
#pragma unroll
                for (int k = 0; k < UP; k++) {
                    uint4 q[NCH];
#pragma unroll
                    for (int i = 0; i < NCH; i++) q[i] = u[k][i];
                    piece(q, true);
                }
            };
            const uint32_t n_units = n_main / UP;
            // warm-up pieces (processed first) and unit 0 in flight together
            {
                uint4 wq[NCH];
                if (a.warm_pieces) {
#pragma unroll
                    for (int i = 0; i < NCH; i++) wq[i] = ld(p_main[i] - 16 * a.warm_pieces);
                }
                if (!have_ua) ld_unit(ua, 0);
                for (uint32_t wp = a.warm_pieces; wp > 0; wp--) {
                    piece(wq, false);
                    if (wp > 1) {
#pragma unroll
                        for (int i = 0; i < NCH; i++) wq[i] = ld(p_main[i] - 16 * (wp - 1));
                    }
                }
            }
#pragma unroll 1
            for (uint32_t u0 = 0; u0 < n_units; u0 += 2) {
                // unconditional prefetches: a conditional load would force the compiler to s_waitcnt vmcnt(0) in front of
                // every use.  Behind the last unit of the chunk the prefetch fetches unit 0 of the wave's NEXT task (round 2
                // re-loaded the last unit there: one line of every four fetched twice, profiles/r02_hot_pmc.json 1.47x).
                ld_unit(ub, u0 + 1 < n_units ? u0 + 1 : n_units - 1);
                do_unit(ua);
                {
                    const bool more = u0 + 2 < n_units;
                    const uint8_t* pn[NCH];
#pragma unroll
                    for (int i = 0; i < NCH; i++)
                        pn[i] = more ? p_main[i] + (16 * UP) * (u0 + 2)
                                     : next_interior ? g.hay16 + next_lo + (uint64_t(lane) + 64 * i) * C
                                                     : p_main[i] + (16 * UP) * (n_units - 1);
                    ld_unit_at(ua, pn);
                }
                if (u0 + 1 < n_units) do_unit(ub);
            }
            have_ua = next_interior;
        } else {
            have_ua = false;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const uint64_t j = j0 + uint64_t(lane) + 64 * i;
                if (j >= a.n_lane_chunks) continue;
                const uint64_t glo = g.grid0 + j * C, ghi = glo + C;
                const uint64_t lo = glo > g.emit_lo ? glo : g.emit_lo;
                const uint64_t hi = ghi < g.emit_hi ? ghi : g.emit_hi;
                if (j == 0 && g.emit_start_matches) cnt[i] += lw_match_len(a, L, a.start);   // empty patterns at span_start
                if (hi > lo) {
                    uint64_t w = lo >= g.halo ? lo - g.halo : 0;
                    if (w < g.cold_floor) w = g.cold_floor;
                    cnt[i] = lw_edge_walk(a, L, g, w, lo, hi, cnt[i]);
                }
            }
        }
        // sum the lane-chunks of each count chunk (lanes_per_chunk consecutive lanes)
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const uint64_t j = j0 + uint64_t(lane) + 64 * i;
            uint32_t c = cnt[i];
            for (uint32_t o = 1; o < a.lanes_per_chunk; o <<= 1) c += __shfl_xor(c, int(o), 64);
            if ((uint32_t(lane) & (a.lanes_per_chunk - 1)) == 0 && j < a.n_lane_chunks) counts[j / a.lanes_per_chunk] = c;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host: tables
// `order` = hid -> nnfa sid, `sid2hid` its inverse (hid_order); the tables themselves: host/lw_tables.cpp.
hipError_t build_lw_tables(const NNfa& n, const Dfa& d, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid,
                           uint32_t first_match, HotTables& out) {
    out.lw_ready = false;
    LwHostTables t;
    if (!build_lw_host(n, d, order, sid2hid, first_match, t)) return hipSuccess;
    const uint32_t image_bytes = uint32_t(t.image.size() * 4);
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.lw_image), image_bytes)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.lw_image, t.image.data(), image_bytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.lw_image_bytes = image_bytes;
    out.lw_row_bytes = t.row_bytes;
    out.lw_wide = t.wide;
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
    out.lw_deep_off = t.deep_off;
    out.lw_nxt_off = t.nxt_off; out.lw_vhid_off = t.vhid_off; out.lw_mlen_off = t.mlen_off;
    out.lw_fm_addr = t.fm_addr;
    out.lw_poison_row = t.poison_row;
    out.lw_start = t.start;
    out.lw_n_dense = t.n_dense; out.lw_n_multi = t.n_multi; out.lw_classes = t.classes;
    out.lw_ready = true;
    return hipSuccess;
}

hipError_t launch_hot_count(const HotTables& h, const DevAutomaton& a, const ScanGeom& g, uint32_t* counts, hipStream_t s) {
    if (!h.lw_ready) return hipErrorInvalidValue;
    LwArgs la{};
    la.image = h.lw_image;
    la.nxt_off = h.lw_nxt_off; la.vhid_off = h.lw_vhid_off; la.mlen_off = h.lw_mlen_off;
    la.image_bytes = h.lw_image_bytes; la.row_bytes = h.lw_row_bytes; la.deep_off = h.lw_deep_off;
    la.fm_addr = h.lw_fm_addr; la.poison_base = h.lw_poison_row; la.start = h.lw_start;
    la.first_match = h.first_match; la.n_states = h.n_states;
    // lane-chunks: the count chunk split into a power-of-two number of pieces of >= kLwLaneChunk bytes (multiples of 64)
    // 512-byte lane-chunks by default; on the largest shards (from 6 GiB on) 1 024-byte ones halve the share of the
    // warm-up line (fabric reads 1.25x -> 1.13x the haystack, +2 % at 8 GiB: profiles/r03_hot_pmc.json, r03_hot_ab.jsonl).
    // Below that they lose to the coarser task grain: 1 GiB 0.41 vs 0.49 ms, 2 GiB 0.71 vs 0.76 ms, 4 GiB 1.33 vs 1.32 ms.
    static const uint32_t target_env = [] { const char* e = std::getenv("ACGPU_LW_LANE_CHUNK"); return e ? uint32_t(std::atoi(e)) : 0u; }();
    const uint32_t target = target_env ? target_env : (g.emit_hi - g.emit_lo >= (uint64_t(6) << 30) ? 2 * kLwLaneChunk : kLwLaneChunk);
    static const int nch_env = [] { const char* e = std::getenv("ACGPU_LW_CHAINS"); return e ? std::atoi(e) : 1; }();
    const int nch = h.lw_wide ? 1 : nch_env;   // the two-chain variant exists for the narrow layout only
    static const int up_env = [] { const char* e = std::getenv("ACGPU_LW_UNIT"); return e ? std::atoi(e) / 16 : 8; }();
    const int up = (up_env == 8 && g.chunk % 128 == 0 && nch == 1) ? 8 : 4;   // whole cache lines when the chunk grid allows
    uint32_t m = 1;
    const uint32_t want = std::max<uint32_t>(target, (8 * g.halo + 63) & ~63u);   // warm-up <= 1/8 of the walk
    const uint32_t unit = 16u * uint32_t(up);   // lane-chunks are whole units
    while (m < 64 && g.chunk % (2 * m * unit) == 0 && g.chunk / (2 * m) >= want) m *= 2;
    la.lanes_per_chunk = m;
    la.lane_chunk = g.chunk / m;
    la.warm_pieces = (g.halo + 15) / 16;
    la.n_lane_chunks = g.n_chunks * m;
    la.n_tasks = (la.n_lane_chunks + 64 * nch - 1) / (64 * nch);
    if (la.n_tasks == 0) return hipSuccess;
    if (h.lw_image_bytes > kLwLdsBytes) return hipErrorInvalidValue;
    uint64_t blocks = uint64_t(device_cus());
    const uint64_t need = (la.n_tasks + kLwWaves - 1) / kLwWaves;
    if (blocks > need) blocks = need;
    la.base_shift = h.lw_wide ? 22 : 24;
    la.e_mask = h.lw_wide ? 0x3Fu : 0xFFu;
    const dim3 grid{uint32_t(blocks)}, block{kLwBlock};
    if (h.lw_wide) {
        if (up == 4) k_lw_count<1, 4, true><<<grid, block, 0, s>>>(la, g, counts);
        else k_lw_count<1, 8, true><<<grid, block, 0, s>>>(la, g, counts);
    } else if (nch == 2 && up == 4) k_lw_count<2, 4, false><<<grid, block, 0, s>>>(la, g, counts);
    else if (up == 4) k_lw_count<1, 4, false><<<grid, block, 0, s>>>(la, g, counts);
    else k_lw_count<1, 8, false><<<grid, block, 0, s>>>(la, g, counts);
    return hipGetLastError();
}

}  // namespace acgpu
