// Prefix-filter engine for the Standard/unanchored overlapping scan (gfx950).
//
// The overlapping search reports "every occurrence of every pattern" (src/automaton.rs:1491-1534 visits the match
// list of every entered match state; reference DESIGN.md:60-63), so the occurrences can be enumerated by START
// position instead of by carrying the automaton state from byte to byte -- which removes the serial dependency of
// the transition walk and lets every position be tested independently:
//
//   level 1  (every other haystack position q; 6 VALU ops + 1 LDS gather = 3 ops per haystack byte)
//       a 64 KiB LDS Bloom table addressed by a 24x24-bit multiplicative hash of b[q+1..q+3] (bits 16..31 of the
//       product: every key byte reaches them).  One gather serves both start positions q and q+1: the table holds
//       every pattern twice (hot.hpp / host/pf_tables.cpp) -- "type 0" = bytes 1..3 as key, byte 0 selects the bit, tested
//       with b[q]; "type 1" = bytes 0..2 as key, byte 3 selects the bit, tested with b[q+4].  ~0.77 % of the even
//       positions of a random haystack survive for the 1k-pattern set, almost all Bloom false positives.
//   survivors  (0.77 % of the probes) each lane peels its survivor bits, takes the window b[q..q+4] from its row
//       registers (the haystack is never re-read) and probes a SECOND, independently hashed 64 KiB Bloom table.
//       Up to 24 000 patterns that table has the construction of the first one and the probe is repeated under the
//       other hash (one gather; a false positive of the first table survives with the fill of the second, 0.8 %, and
//       both starts the probe stands for go on); for larger sets (HotTables::pf_exact2, kernel instance X2) it holds
//       one entry per pattern keyed by the true start and is probed once per candidate start (q: key b[q..q+2], bit
//       b[q+3]; q+1: key b[q+1..q+3], bit b[q+4]).  What is left -- ~0.03 probes per 1008-byte row on the headline
//       input, almost all true 4-byte prefix matches -- goes straight to level 3.
//   level 3  (dense batches of 64 from a per-wavefront LDS queue filled at __ballot/mbcnt ranks)   exact walk of the
//       trie-only (anchored) transition table in global memory from the start state.  Every pattern end is credited
//       to the chunk owning it (classic mode, counts for the scan + fill pipeline) or recorded as an event
//       {end, length, trie node} in a per-wavefront LDS buffer that is appended to the global event list with one
//       atomic per flush (event modes: k_ev_rank / k_ev_write below, or the radix sort of event_sort.hip for large
//       result sets, order the events and emit the records without re-walking the haystack).
//
// HBM is read exactly once, fully coalesced (lane l loads 16 B at row + 16 l, non-temporal; four row-pair register
// sets rotate so three pairs are in flight while one is filtered, carried from each task into the wave's next one).  The filter has no false negatives by construction
// and every survivor is verified exactly, so the result is exact for every input (tables: host/pf_tables.cpp; their
// completeness is checked on the CPU by tests/test_pf_tables.py).  Unavailable (the host uses the transition-walk engines)
// when a pattern is empty or the set exceeds 131 072 patterns / 2^20 states.
//
// Routing (PfArgs::route_*, drain_q2): a wavefront that has handed 1 024 starts to level 3 compares its own cost so far
// with the model of the alternative engine (LDS walk, large-set filter, or global DFA walk: capi_overlap.cpp::pf_alternative) and
// abandons the scan when it predicts the alternative to win; the host then repeats the search with that engine.
//
// VALU budget (measured, scripts/ubench/valu_rate.hip): integer shifts / mul24 / alignbit / perm / SDWA forms issue at
// 4 cycles per wavefront-instruction per SIMD on gfx950, add / xor / or / bitop3 at 2.  Level 1 per q is
// [alignbit] -> mul_u32_u24 -> and (WORD_1 SDWA) -> ds_read_b32 -> two SDWA byte-select shifts -> or -> alignbit; the
// kernel runs at ~85 % VALU issue and ~85 % of the practical streaming-read ceiling (profiles/r01_ubench_stream_ceiling.txt).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hot.hpp"
#include "launch_util.hpp"
#include "pf_common.hpp"

// PF_EXP: bit mask of timing experiments (scripts/pf_variants.sh); 0 in the product build.
//   1 = no survivor handling   2 = no LDS gathers   4 = conflict-free gathers   8 = no level 1 at all
//   32 = survivor loop with the second probe but no pushes / level 3   64 = survivor loop without the second probe
#ifndef PF_EXP
#define PF_EXP 0
#endif

namespace acgpu {

namespace {

using namespace pfdev;

constexpr unsigned long long kPfEventCost = 464000ull;   // 5000 * 130 / 1.4: an event in the units of the routing rule (drain_q2)

// Per-wavefront state of the filter pipeline.  X2: second table keyed by true starts, probed once per candidate start
// (HotTables::pf_exact2, large pattern sets).
template <bool X2, bool FOLD = false>   // FOLD: the row registers are or-ed with 0x20 per byte before the tables see them (HotTables::pf_fold)
struct PfWave {
    const PfArgs& a;
    const ScanGeom& g;
    uint32_t* counts;
    const uint32_t* s_bits;  // level-1 bit table (static LDS)
    const uint32_t* s_bits2; // second bit table (dynamic LDS)
    const uint8_t* s_acls;   // class map of the trie table (static LDS)
    uint64_t* q2;        // survivors of both tables: absolute (virtual) start positions, verified in batches of 64
    PfEvent* ebuf;       // per-wave event buffer + its fill counter (event modes)
    uint32_t* ecnt;
    uint32_t* rt;        // per-wave routing counters (LDS): [0] level-3 starts, [1] events flushed, [2] tasks started, [3] abandoned
    uint64_t task_base = 0;
    uint32_t q2count = 0;    // wave-uniform fill level; a batch = the LAST (up to) 64 entries (order is irrelevant)
    uint32_t xcount = 0;     // wave-uniform: starts handed to level 3 so far (the probe's measure)
    uint4 ra[kSets] = {}, rb[kSets] = {};   // row-pair register sets (rows 2i / 2i+1 of the pair in set i % kSets)
    uint32_t dummy = 0;      // PF_EXP experiments only
    bool carried = false;    // wave-uniform: sets 0..kSets-2 already receive the first pairs of the task to run
    int lane = 0;
    uint32_t amask = 0;

    __device__ __forceinline__ void drain_q2(uint32_t n) {
        pf_fence();
        q2count = uni(q2count - n);
        xcount += n;
        if (a.route_cb && uni(rt[3])) return;   // scan abandoned: its result is discarded, nothing left to verify
        if (a.skip_verify) return;              // (the probe only counts)
        uint64_t v = 0;
        if (uint32_t(lane) < n) v = q2[q2count + lane];
        pf_fence();
        bool buffered = false;
        bool go = uint32_t(lane) < n;
        if (a.bits3 && go && v + 4 <= g.emit_hi) {   // all four bytes inside the span: one gather decides most candidates
            uint32_t k4;
            ACGPU_HAY_CHECK(g, v, 4);
            __builtin_memcpy(&k4, g.hay16 + v, 4);
            const uint32_t h = pf_hash3(k4, a.bits3_log2);
            go = ((a.bits3[h >> 5] >> (h & 31)) & 1u) != 0;
        }
        if (go) buffered = pf_verify(a, g, counts, v, ebuf, ecnt, s_acls);
        if (__builtin_amdgcn_ballot_w64(buffered) != 0) flush_events(kEvFlush);
        if (a.route_cb) {
            pf_fence();
            const uint32_t cand = uni(rt[0]) + n;
            // (wave-local decision: polling a global flag here -- every wavefront loading ONE address past the caches
            // at every batch -- serialised in a single memory channel at ~200 ns per load and cost the headline scan 0.7 ms)
            bool stop = false;
            if (cand >= 1024) {
                // bytes this wavefront has scanned: its finished tasks + how far the newest candidate lies into the
                // current one (counting the whole 40 KB task from its first row on made a dense input wait for the end
                // of every wavefront's first task: 0.7 ms of a 1 GiB call)
                const uint64_t vlast = uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v)), int(n - 1)))) |
                                       (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v >> 32)), int(n - 1)))) << 32);
                const uint64_t tasks = uni(rt[2]);
                const uint64_t into = vlast > task_base ? vlast - task_base : 0;
                const uint64_t bytes = (tasks ? tasks - 1 : 0) * (uint64_t(kTaskRows) * kRowBytes) + into + kRowBytes;
                const uint64_t m_ev = uint64_t(uni(rt[1])) + uni(*ecnt);
                const uint64_t m256 = 256ull * m_ev;
                // (kPfEventCost: what recording an occurrence costs the filter -- the flushes of all wavefronts meet in two
                // global counters, ~1.4 G events/s chip-wide: a match-dense input never showed in X alone and ran 18 GB/s)
                stop = 5000ull * cand + kPfEventCost * m_ev > uint64_t(a.route_cb) * bytes + uint64_t(a.route_cr) * (m256 < bytes ? m256 : bytes);
                if (stop && lane == 0) atomicExch(&a.ev_ctr[2], 1ull);
            }
            if (lane == 0) { rt[0] = cand; rt[3] = stop ? 1u : 0u; }
            pf_fence();
        }
        // Retire this path's stores / atomics before going back to the row loop: with store-type operations still
        // pending the compiler can only order the next use of a prefetched row with s_waitcnt vmcnt(0), which would
        // also wait for the row pairs just issued and serialise every pair with the memory latency.
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
    }
    // appends the buffered events to the global list if at least `at_least` are waiting (wave-uniform)
    __device__ __forceinline__ void flush_events(uint32_t at_least) {
        pf_fence();
        uint32_t n = uni(*ecnt);
        if (n < at_least) return;
        if (n > uint32_t(kEvBuf)) n = kEvBuf;    // the lanes beyond the buffer appended their events themselves
        if (a.route_cb && lane == 0) rt[1] += n;
        uint64_t key = 0;
        uint32_t node = 0, cnt = 0;
        if (uint32_t(lane) < n) { key = ebuf[lane].key; node = ebuf[lane].node; cnt = ebuf[lane].cnt; }
        uint32_t recs = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) recs += __shfl_xor(recs, o, 64);
        unsigned long long base = 0;
        if (lane == 0) {
            base = atomicAdd(&a.ev_ctr[0], static_cast<unsigned long long>(n));
            atomicAdd(&a.ev_ctr[1], static_cast<unsigned long long>(recs));
            *ecnt = 0;
        }
        base = (static_cast<unsigned long long>(uni(uint32_t(base >> 32))) << 32) | uni(uint32_t(base));
        if (uint32_t(lane) < n && base + lane < a.ev_cap) {
            PfEvent* dst = a.events + (base + lane);
            dst->key = key; dst->node = node; dst->cnt = cnt;
            if (a.eo_bb) pf_eo_hist(a, key, cnt, base + lane);
        }
        pf_fence();
    }
    static __device__ __forceinline__ uint32_t uni(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }

    // level 1 over the 8 ODD offsets q = 1,3,..,15 of a lane's row (wd[0..4] = its 16 bytes + 4 look-ahead).
    // One gather per q serves the two start positions q and q+1 (hot.hpp): key = b[q+1..q+3], probe bits selected
    // by b[q] and b[q+4].  Odd q makes the key dword-aligned for every other lookup (b[q+1..q+3] is the low 24 bits
    // of a row dword when q % 4 == 3: mul_u32_u24 ignores the top byte) and every selector a plain byte of a row
    // dword (SDWA operand).  Start 0 of a lane is start 16 of its left neighbour (lane 63 / the previous row for
    // lane 0); the first start of the whole scan is handed to level 3 directly (k_pf_count).
    // `probe` issues the 8 LDS gathers of a row, `fold` appends the 8 survivor bits below `hits`.
    static __device__ __forceinline__ uint32_t key_at(const uint32_t (&wd)[5], int q) {   // low 24 bits = b[q+1..q+3], q odd
        const int i = (q + 1) >> 2;
        return ((q + 1) & 2) == 0 ? wd[i] : __builtin_amdgcn_alignbit(wd[i + 1], wd[i], 16);
    }
    // word << (b[o] & 31) for odd o: the shift amount is byte 1 or 3 of a row dword, selected by the SDWA operand
    // modifier (the hardware uses the low 5 bits of the selected byte) -- no separate shift to extract the byte
    template <int O>
    static __device__ __forceinline__ uint32_t shl_by_byte(uint32_t word, const uint32_t (&wd)[5]) {
        uint32_t r;
        if ((O & 2) == 0)
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
                : "=v"(r) : "v"(wd[O >> 2]), "v"(word));
        else
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
                : "=v"(r) : "v"(wd[O >> 2]), "v"(word));
        return r;
    }
    __device__ __forceinline__ void probe(const uint32_t (&wd)[5], uint32_t (&word)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t h = pf_hash(key_at(wd, 2 * j + 1)) & amask;
            if (PF_EXP & 2) word[j] = h;
            else if (PF_EXP & 4) word[j] = s_bits[((h & 0xFF00u) | (uint32_t(lane) << 2)) >> 2];
            else word[j] = s_bits[h >> 2];
        }
    }
    template <int J>
    __device__ __forceinline__ uint32_t fold1(uint32_t hits, const uint32_t (&wd)[5], const uint32_t (&word)[8]) const {
        const uint32_t t = shl_by_byte<2 * J + 1>(word[J], wd) | shl_by_byte<2 * J + 5>(word[J], wd);
        return __builtin_amdgcn_alignbit(hits, t, 31);
    }
    __device__ __forceinline__ uint32_t fold(uint32_t hits, const uint32_t (&wd)[5], const uint32_t (&word)[8]) const {
        hits = fold1<0>(hits, wd, word); hits = fold1<1>(hits, wd, word); hits = fold1<2>(hits, wd, word);
        hits = fold1<3>(hits, wd, word); hits = fold1<4>(hits, wd, word); hits = fold1<5>(hits, wd, word);
        hits = fold1<6>(hits, wd, word); hits = fold1<7>(hits, wd, word);
        return hits;
    }
    // rows w0, w1 -> 16 survivor bits: bit 15-i <=> row i >> 3, offset q = 2 * (i & 7) + 1
    __device__ __forceinline__ uint32_t level1_pair(const uint32_t (&w0)[5], const uint32_t (&w1)[5]) const {
        uint32_t A[8], B[8];
        probe(w0, A);
        probe(w1, B);
        __builtin_amdgcn_sched_barrier(0);
        return fold(fold(0u, w0, A), w1, B) & 0xFFFFu;
    }

    // the window b[k..k+3] of a lane's row registers, k = 0..15 dynamic (cndmask tree + funnel shift), and b[k+4]
    // (low byte of `next`)
    static __device__ __forceinline__ uint32_t window(const uint32_t (&wd)[5], uint32_t k, uint32_t& next) {
        const bool up = (k & 8u) != 0, mid = (k & 4u) != 0;
        const uint32_t c0 = up ? wd[2] : wd[0], c1 = up ? wd[3] : wd[1], c2 = up ? wd[4] : wd[2];
        const uint32_t hi = mid ? c2 : c1;
        next = hi >> (8u * (k & 3u));
        return __builtin_amdgcn_alignbit(hi, mid ? c1 : c0, 8u * (k & 3u));
    }

    // level-1 survivors of a pair of rows.  Each lane that has one peels its lowest survivor q, takes the window
    // b[q..q+4] from its row registers (the haystack is never re-read) and repeats the level-1 probe in the SECOND,
    // independently hashed Bloom table.  A false positive of the first table survives with the fill of the second
    // (0.8 %): what is left (~0.03 probes per row on the headline input, almost all true 4-byte prefix matches) hands
    // both start positions it stands for to the exact trie walk of level 3.
    __device__ __forceinline__ void survivors(uint32_t hits, const uint32_t (&w0)[5], const uint32_t (&w1)[5], uint32_t off) {
        while (__any(hits != 0)) {
            const bool has = hits != 0;
            const uint32_t i = (15u - uint32_t(__builtin_ctz(hits | 0x10000u))) & 15u;
            hits &= hits - 1;
            const bool second = (i & 8u) != 0;
            const uint32_t wd[5] = {second ? w1[0] : w0[0], second ? w1[1] : w0[1], second ? w1[2] : w0[2],
                                    second ? w1[3] : w0[3], second ? w1[4] : w0[4]};
            const uint32_t q = (i & 7u) * 2u + 1u;
            uint32_t next;
            const uint32_t win = window(wd, q, next);
            if (PF_EXP & 64) { dummy += has ? win + next : 0u; continue; }   // experiment: peel + window only
            // the same probe as level 1 (key b[q+1..q+3], bits selected by b[q] and b[q+4]) in the second table
            bool ok_a, ok_b;
            if (X2) {   // the two exact candidate starts (q: key b[q..q+2], bit b[q+3]; q+1: key b[q+1..q+3], bit b[q+4])
                const uint32_t wa = s_bits2[(pf_hash2(win) & (kPfBits2Bytes - 4)) >> 2];
                const uint32_t wb = s_bits2[(pf_hash2(win >> 8) & (kPfBits2Bytes - 4)) >> 2];
                ok_a = has && int32_t(wa << ((win >> 24) & 31)) < 0;
                ok_b = has && int32_t(wb << (next & 31)) < 0;
            } else {
                const uint32_t w2 = s_bits2[(pf_hash2(win >> 8) & (kPfBits2Bytes - 4)) >> 2];
                const bool ok = has && int32_t((w2 << (win & 31)) | (w2 << (next & 31))) < 0;
                ok_a = ok; ok_b = ok;   // either start may be the one: level 3 verifies both
            }
            if (PF_EXP & 32) { dummy += uint32_t(ok_a) + uint32_t(ok_b); continue; }   // experiment: no pushes / level 3
            if (__any(ok_a | ok_b)) {
                const uint64_t v = task_base + off + (second ? kRowBytes : 0u) + q;
                push_q2(ok_a && v >= a.scan_lo && v < g.emit_hi, v);
                push_q2(ok_b && v + 1 >= a.scan_lo && v + 1 < g.emit_hi, v + 1);
            }
        }
    }

    __device__ __forceinline__ void push_q2(bool ok, uint64_t v) {
        if (__any(ok)) {
            const unsigned long long m = __ballot(ok);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
            if (ok) q2[q2count + rank] = v;
            q2count = uni(q2count + uint32_t(__popcll(m)));
            if (q2count >= 64) drain_q2(64);
        }
    }

    // one task = kTaskRows rows, processed two rows per step.  GUARD = per-lane bounds / ownership checks
    // (only the first and last tasks of a scan need them).
    // `next_base` / `next_interior`: the task this wave runs next.  When it is an interior one, the loads that would
    // run past the end of this task fetch ITS first pairs instead, so the pipeline never drains between tasks.
    template <bool GUARD>
    __device__ __forceinline__ void run_task(uint64_t task_base, uint64_t next_base, bool next_interior) {
        // the haystack is read exactly once: non-temporal loads keep it from evicting the tables from L2
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        auto load_plain = [&](uint64_t p, uint4& w) {
            ACGPU_HAY_CHECK(g, p, 16);
            const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(g.hay16 + p));
            w = make_uint4(t.x, t.y, t.z, t.w);
        };
        auto load = [&](uint64_t p, uint4& w) {
            if (GUARD) {
                w = make_uint4(0, 0, 0, 0);
                if (p < a.hull_end) { ACGPU_HAY_CHECK(g, p, 16); w = *reinterpret_cast<const uint4*>(g.hay16 + p); }
            } else {
                load_plain(p, w);
            }
        };
        this->task_base = task_base;
        uint64_t p = task_base + uint64_t(lane) * 16;
        uint32_t off = uint32_t(lane) * 16;  // p - task_base
        // kSets register sets in rotation: while one pair of rows is filtered, the next kSets-1 pairs are in
        // flight (the wave needs ~50 KB per CU outstanding to cover HBM latency at full rate), and no register
        // copies are needed to free the load destinations
        if (!carried) {
#pragma unroll
            for (int j = 0; j < kSets - 1; j++) {
                load(p + uint64_t(2 * j) * kRowBytes, ra[j]);
                load(p + uint64_t(2 * j + 1) * kRowBytes, rb[j]);
            }
        }
        carried = false;
        auto pair = [&](const uint4& wa, const uint4& wb) {
            // 4-byte look-ahead = first dword of the right neighbour lane (DPP wave shift, no memory traffic)
            constexpr uint32_t fm = FOLD ? 0x20202020u : 0u;   // (levels 1 and 2 only: level 3 reads the haystack itself)
            const uint32_t w0[5] = {wa.x | fm, wa.y | fm, wa.z | fm, wa.w | fm, uint32_t(__builtin_amdgcn_update_dpp(0, int(wa.x), 0x130, 0xF, 0xF, false)) | fm};
            const uint32_t w1[5] = {wb.x | fm, wb.y | fm, wb.z | fm, wb.w | fm, uint32_t(__builtin_amdgcn_update_dpp(0, int(wb.x), 0x130, 0xF, 0xF, false)) | fm};
            uint32_t hits = (PF_EXP & 8) ? uint32_t((w0[0] ^ w1[1] ^ w0[2] ^ w1[3] ^ w0[4] ^ w1[4]) == 0x12345678u) : level1_pair(w0, w1);
            if (PF_EXP & 1) hits = (hits == 0xFFFFu && w0[0] == 0x12345678u) ? 1u : 0u;
            if (lane == 63) hits = 0;  // lane 63's 16 bytes are lane 0 of the next row
            survivors(hits, w0, w1, off);  // (start positions outside [scan_lo, emit_hi) are dropped before level 3)
            p += 2 * kRowBytes;
            off += 2 * kRowBytes;
        };
        static_assert(kTaskRows % (2 * kSets) == 0, "kSets row pairs per iteration");
        constexpr uint32_t kPairs = kTaskRows / 2;
        const uint64_t next_p = next_base + uint64_t(lane) * 16;
        bool completed = true;
#pragma unroll 1
        for (uint32_t r = 0; r < kTaskRows; r += 2 * kSets) {
            if (GUARD && task_base + uint64_t(r) * kRowBytes >= g.emit_hi) { completed = false; break; }  // wave-uniform
#pragma unroll
            for (int j = 0; j < kSets; j++) {
                constexpr int kAhead = kSets - 1;           // pairs between the load and its use
                const int n = (j + kAhead) % kSets;         // the set that was consumed last
                // Always issue the two loads: a conditional load would merge two different queues of outstanding loads
                // in front of pair(), and the compiler would have to order the rows with the strictest s_waitcnt
                // (vmcnt(1)/(0)), i.e. wait for the loads just issued.  Past the end of this task they fetch the first
                // pairs of the next one (or, when that one needs guarded loads, re-read the current pair).
                const uint32_t pi = r / 2 + uint32_t(j + kAhead);
                uint64_t src = p + uint64_t(2 * kAhead) * kRowBytes;
                if (pi >= kPairs) src = next_interior ? next_p + uint64_t(pi - kPairs) * (2 * kRowBytes) : p;
                if (GUARD && !(pi >= kPairs && next_interior)) {   // only the next task's rows are known to be in bounds
                    load(src, ra[n]);
                    load(src + kRowBytes, rb[n]);
                } else {
                    load_plain(src, ra[n]);
                    load_plain(src + kRowBytes, rb[n]);
                }
                pair(ra[j], rb[j]);
            }
        }
        carried = completed && next_interior;
    }
};

template <bool X2, bool FOLD = false>
__global__ __launch_bounds__(kPfBlock) void k_pf_count(PfArgs a, ScanGeom g, uint32_t* __restrict__ counts) {
    // LDS: static [bit table], dynamic [bigram table | per-wave level-3 queues]
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[kBitsBytes / 4];
    __shared__ uint8_t s_acls[256];
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_bits2 = reinterpret_cast<uint32_t*>(smem);
    uint64_t* s_q = reinterpret_cast<uint64_t*>(smem + kPfBits2Bytes);
    PfEvent* s_ev = reinterpret_cast<PfEvent*>(smem + kPfBits2Bytes + size_t(kPfWaves) * kQueue * sizeof(uint64_t));
    uint32_t* s_ecnt = reinterpret_cast<uint32_t*>(s_ev + kPfWaves * kEvBuf);
    uint32_t* s_rt = s_ecnt + kPfWaves;
    if (a.gate && *a.gate != a.gate_val) return;   // (the probe chose the other filter)
    if (threadIdx.x < kPfWaves) s_ecnt[threadIdx.x] = 0;
    if (threadIdx.x < 256) s_acls[threadIdx.x] = a.acls[threadIdx.x];
    if (threadIdx.x < kPfWaves * 4) s_rt[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < kBitsBytes / 4; i += kPfBlock) s_bits[i] = a.bits[i];
    for (uint32_t i = threadIdx.x; i < kPfBits2Bytes / 4; i += kPfBlock) s_bits2[i] = a.bits2[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PfWave<X2, FOLD> st{a, g, counts, s_bits, s_bits2, s_acls, s_q + wave * kQueue, s_ev + wave * kEvBuf, s_ecnt + wave, s_rt + wave * 4};
    st.lane = lane;
    st.amask = (kBitsBytes - 1) & ~3u;

    const uint64_t task_bytes = uint64_t(kTaskRows) * kRowBytes;
    const uint64_t wave_id = uint64_t(blockIdx.x) * kPfWaves + wave;
    const uint64_t n_waves = uint64_t(gridDim.x) * kPfWaves;
    // the very first start position has no left neighbour to cover it when the scan begins on a row boundary
    if (wave_id == 0 && a.n_tasks && a.scan_lo == a.row0) st.push_q2(lane == 0 && a.scan_lo < g.emit_hi, a.scan_lo);
    auto is_interior = [&](uint64_t tb) {
        // interior task: every start position is owned and every load (incl. 4-byte look-ahead) is in bounds
        return tb >= a.scan_lo && tb + task_bytes + 16 <= a.hull_end && tb + task_bytes <= g.emit_hi;  // (+16: lane 63 of the last row)
    };
    for (uint64_t task = wave_id; task < a.n_tasks; task += n_waves) {
        if (a.route_cb) {   // (once per 40 KB task)
            pf_fence();
            if (st.uni(s_rt[wave * 4 + 3])) break;
            if (lane == 0) s_rt[wave * 4 + 2] += 1;
        }
        const uint64_t task_base = a.row0 + task * task_bytes;
        const uint64_t next_base = a.row0 + (task + n_waves) * task_bytes;
        const bool next_interior = task + n_waves < a.n_tasks && is_interior(next_base);
        if (is_interior(task_base)) st.template run_task<false>(task_base, next_base, next_interior);
        else st.template run_task<true>(task_base, next_base, next_interior);
    }
    if (PF_EXP && st.dummy == 0x12345u) counts[0] = st.dummy;
    // final partial batch of level 3
    while (st.q2count) st.drain_q2(st.q2count < 64 ? st.q2count : 64);
    if (a.events) st.flush_events(1);
}


// Probe (hot.hpp: launch_pf_probe): every wavefront runs the filter -- all three levels -- over ONE 8 KB sample of the
// shard (samples `stride` bytes apart) and adds what it measured to the probe counters: pc[0] starts handed to level 3
// (counted, not verified: pc[5], the pattern ends, stays 0 -- a match-dense input the start count alone does not give
// away is still caught by the scan kernel's own rule), pc[1] bytes sampled.  The last wavefront to finish applies the
// routing rule of the scan kernel (drain_q2) to the totals and writes the decision.
template <bool X2, bool FOLD = false>
__global__ __launch_bounds__(kPfBlock) void k_pf_probe(PfArgs a, ScanGeom g, uint32_t n_samples, uint64_t stride,
                                                       uint32_t route_cb, uint32_t route_cr, uint32_t* __restrict__ decision,
                                                       unsigned long long* __restrict__ pc) {
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[kBitsBytes / 4];
    __shared__ uint8_t s_acls[256];
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_bits2 = reinterpret_cast<uint32_t*>(smem);
    uint64_t* s_q = reinterpret_cast<uint64_t*>(smem + kPfBits2Bytes);
    PfEvent* s_ev = reinterpret_cast<PfEvent*>(smem + kPfBits2Bytes + size_t(kPfWaves) * kQueue * sizeof(uint64_t));
    uint32_t* s_ecnt = reinterpret_cast<uint32_t*>(s_ev + kPfWaves * kEvBuf);
    uint32_t* s_rt = s_ecnt + kPfWaves;
    if (threadIdx.x < kPfWaves) s_ecnt[threadIdx.x] = 0;
    if (threadIdx.x < 256) s_acls[threadIdx.x] = a.acls[threadIdx.x];
    if (threadIdx.x < kPfWaves * 4) s_rt[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < kBitsBytes / 4; i += kPfBlock) s_bits[i] = a.bits[i];
    for (uint32_t i = threadIdx.x; i < kPfBits2Bytes / 4; i += kPfBlock) s_bits2[i] = a.bits2[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t sample = uint64_t(blockIdx.x) * kPfWaves + wave;
    if (sample >= n_samples) return;
    constexpr uint64_t kSampleBytes = uint64_t(2 * kSets) * kRowBytes;   // one iteration of the row loop
    const uint64_t ws = a.row0 + sample * stride;
    uint64_t bytes = 0;
    uint32_t x = 0;
    if (ws + kSampleBytes <= g.emit_hi) {
        PfArgs la = a;
        ScanGeom lg = g;
        la.scan_lo = ws; la.row0 = ws; la.n_tasks = 1; la.route_cb = 0; la.route_cr = 0; la.skip_verify = 1;
        la.events = reinterpret_cast<PfEvent*>(pc); la.ev_cap = 0; la.ev_ctr = pc + 4;   // (counted, never stored)
        lg.emit_lo = ws; lg.emit_hi = ws + kSampleBytes;
        const uint64_t he = (lg.emit_hi + 31) & ~uint64_t(15);
        la.hull_end = he < a.hull_end ? he : a.hull_end;
        PfWave<X2, FOLD> st{la, lg, nullptr, s_bits, s_bits2, s_acls, s_q + wave * kQueue, s_ev + wave * kEvBuf, s_ecnt + wave, s_rt + wave * 4};
        st.lane = lane;
        st.amask = (kBitsBytes - 1) & ~3u;
        st.template run_task<true>(ws, 0, false);
        while (st.q2count) st.drain_q2(st.q2count < 64 ? st.q2count : 64);
        st.flush_events(1);
        bytes = kSampleBytes;
        x = st.xcount;
    }
    if (lane == 0) {
        atomicAdd(&pc[0], static_cast<unsigned long long>(x));
        atomicAdd(&pc[1], static_cast<unsigned long long>(bytes));
        __threadfence();
        if (atomicAdd(&pc[2], 1ull) + 1 == n_samples) {   // the last sample: decide, and leave the counters zeroed
            __threadfence();
            const unsigned long long X = atomicAdd(&pc[0], 0ull), B = atomicAdd(&pc[1], 0ull), M = atomicAdd(&pc[5], 0ull);
            const unsigned long long m256 = 256ull * M;
            const bool stop = B > 0 && 5000ull * X + kPfEventCost * M > static_cast<unsigned long long>(route_cb) * B + static_cast<unsigned long long>(route_cr) * (m256 < B ? m256 : B);
            *decision = stop ? 1u : 0u;
            pc[0] = 0; pc[1] = 0; pc[2] = 0; pc[4] = 0; pc[5] = 0;
        }
    }
}

// ---- direct mode: ordered records from the level-3 events.
// The reference order (ascending end; inside one end the entered state's match list = own patterns of the longest
// suffix first, then ever shorter ones, src/nfa/noncontiguous.rs:466-523) is the order of the keys
// (end << 16 | 0xFFFF - length): all occurrences ending at one position have different lengths, and patterns that
// are the same string sit in one trie node whose own patterns lead the state's list in pattern-id order.
// rank(i) = number of records of the events with a smaller key, computed all-pairs over LDS tiles (n <= ev_cap).
constexpr int kEvTile = 128;    // j-tile per workgroup: 256 events x 128 events of the all-pairs comparison

__global__ __launch_bounds__(256) void k_ev_rank(const PfEvent* __restrict__ ev, const unsigned long long* __restrict__ ctr,
                                                 uint32_t cap, uint32_t* __restrict__ rank,
                                                 uint64_t* __restrict__ totals) {
    __shared__ uint64_t s_key[kEvTile];
    __shared__ uint32_t s_cnt[kEvTile];
    // an abandoned scan (PfArgs::route_*) reports "more events than any buffer holds": every consumer of the totals
    // already treats that as "repeat the search another way"
    const unsigned long long n64 = ctr[2] ? ~0ull : ctr[0];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { totals[0] = ctr[1]; totals[1] = n64; }
    if (n64 > cap) return;   // overflow: the host switches to the sorted-events mode or the classic pipeline
    const uint32_t n = uint32_t(n64);
    // grid-stride over the (i-block, j-tile) pairs: correct for any grid, sized by the host from the previous call
    for (uint32_t j0 = blockIdx.y * kEvTile; j0 < n; j0 += gridDim.y * kEvTile) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < uint32_t(kEvTile); t += 256) {
            const bool ok = j0 + t < n;
            s_key[t] = ok ? ev[j0 + t].key : ~0ull;
            s_cnt[t] = ok ? ev[j0 + t].cnt : 0u;
        }
        __syncthreads();
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            const uint64_t key = ev[i].key;
            uint32_t r = 0;
#pragma unroll 8
            for (int t = 0; t < kEvTile; t++) r += s_key[t] < key ? s_cnt[t] : 0u;
            if (r) atomicAdd(&rank[i], r);
        }
    }
}

// one thread per event: its records at out[rank ..); restores rank[] = 0 and the counter for the next call
__global__ __launch_bounds__(256) void k_ev_write(DfaEng eng, const uint32_t* __restrict__ hid2sid,
                                                  const PfEvent* __restrict__ ev, unsigned long long* __restrict__ ctr,
                                                  uint32_t cap, uint32_t* __restrict__ rank,
                                                  const uint64_t* __restrict__ totals, uint64_t out_cap,
                                                  acgpu_match* __restrict__ out, uint64_t* __restrict__ also_zero, uint32_t also_zero_words) {
    // (a region the next kernels on the stream want zeroed -- the order pass's bucket counters -- rides along)
    for (uint32_t z = blockIdx.x * 256 + threadIdx.x; z < also_zero_words; z += gridDim.x * 256) also_zero[z] = 0;
    const uint64_t n = totals[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctr[0] = 0ull; ctr[1] = 0ull; ctr[2] = 0ull; }   // (totals were copied out by k_ev_rank)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (n > cap || i >= n) return;
    const uint32_t r = rank[i];
    rank[i] = 0;
    if (totals[0] > out_cap) return;   // the host reports ACGPU_ERR_BUFFER_TOO_SMALL
    const PfEvent e = ev[i];
    const uint32_t sid = hid2sid[e.node];
    const uint64_t end = e.key >> 16, len = 0xFFFFull - (e.key & 0xFFFFull);
    for (uint32_t k = 0; k < e.cnt; k++) {
        acgpu_match m; m.pattern = eng.match_pattern(sid, k); m._pad = 0; m.start = end - len; m.end = end;
        out[r + k] = m;
    }
}

}  // namespace

hipError_t launch_pf_count(const HotTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s, void* events,
                           unsigned long long* ev_ctr, uint64_t ev_cap, PfRoute route) {
    PfArgs a{};
    a.events = static_cast<PfEvent*>(events); a.ev_ctr = ev_ctr; a.ev_cap = ev_cap;
    if (events) { a.route_cb = route.cb; a.route_cr = route.cr; }
    if (events) { a.eo_bb = route.hist.bb; a.eo_slot = route.hist.slot; a.eo_origin = route.hist.origin; a.eo_shift = route.hist.shift; }
    a.gate = route.gate; a.gate_val = route.gate_val;
    a.bits = h.pf_bits; a.bits2 = h.pf_bits2; a.atab = h.atab; a.acls = h.acls; a.ashift = h.ashift; a.own_cnt = h.own_cnt;
    a.bits3 = h.pf_bits3; a.bits3_log2 = h.pf_bits3_log2;
    a.bits_bytes = h.pf_bits_bytes; a.root = h.start;
    const uint64_t lo = g.emit_lo >= g.halo ? g.emit_lo - g.halo : 0;
    a.scan_lo = lo > g.cold_floor ? lo : g.cold_floor;
    a.row0 = a.scan_lo & ~uint64_t(15);
    a.hull_end = (g.emit_hi + 15) & ~uint64_t(15);
    const uint64_t task_bytes = uint64_t(kTaskRows) * kRowBytes;
    a.n_tasks = g.emit_hi > a.row0 ? (g.emit_hi - a.row0 + task_bytes - 1) / task_bytes : 0;
    hipError_t e = hipSuccess;
    if (!events) e = hipMemsetAsync(counts, 0, g.n_chunks * sizeof(uint32_t), s);   // direct mode has no chunk counters
    if (e != hipSuccess) return e;
    if (a.n_tasks == 0) return hipSuccess;
    if (a.bits_bytes != kBitsBytes) return hipErrorInvalidValue;
    const size_t smem = size_t(kPfBits2Bytes) + size_t(kPfWaves) * kQueue * sizeof(uint64_t) +
                        size_t(kPfWaves) * (kEvBuf * sizeof(PfEvent) + sizeof(uint32_t) + 4 * sizeof(uint32_t));
    typedef void (*CountKernel)(PfArgs, ScanGeom, uint32_t*);
    const CountKernel kern = h.pf_exact2 ? (h.pf_fold ? k_pf_count<true, true> : k_pf_count<true, false>)
                                         : (h.pf_fold ? k_pf_count<false, true> : k_pf_count<false, false>);
    e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - int(kBitsBytes) - 512);   // static: bit table
    if (e != hipSuccess) return e;
    const int cus = device_cus();
    const uint64_t blocks_per_cu = std::max<uint64_t>(1, std::min<uint64_t>(2, (160 * 1024) / (smem + kBitsBytes + 1024)));
    uint64_t blocks = uint64_t(cus) * blocks_per_cu;
    const uint64_t need = (a.n_tasks + kPfWaves - 1) / kPfWaves;
    if (blocks > need) blocks = need;
    kern<<<dim3(uint32_t(blocks)), dim3(kPfBlock), smem, s>>>(a, g, counts);
    return hipGetLastError();
}

hipError_t launch_pf_probe(const HotTables& h, const ScanGeom& g, PfRoute route, uint32_t* decision, unsigned long long* probe_ctr, hipStream_t s) {
    PfArgs a{};
    a.bits = h.pf_bits; a.bits2 = h.pf_bits2; a.atab = h.atab; a.acls = h.acls; a.ashift = h.ashift; a.own_cnt = h.own_cnt;
    a.bits3 = h.pf_bits3; a.bits3_log2 = h.pf_bits3_log2;
    a.bits_bytes = h.pf_bits_bytes; a.root = h.start;
    const uint64_t lo = g.emit_lo >= g.halo ? g.emit_lo - g.halo : 0;
    a.scan_lo = lo > g.cold_floor ? lo : g.cold_floor;
    a.row0 = (a.scan_lo + 15) & ~uint64_t(15);
    a.hull_end = (g.emit_hi + 15) & ~uint64_t(15);
    if (a.bits_bytes != kBitsBytes) return hipErrorInvalidValue;
    constexpr uint32_t kSamples = 256;
    const uint64_t span = g.emit_hi > a.row0 ? g.emit_hi - a.row0 : 0;
    const uint64_t stride = (span / kSamples) & ~uint64_t(15);
    if (stride < uint64_t(2 * kSets) * kRowBytes) return hipErrorInvalidValue;   // (callers probe large shards only)
    const size_t smem = size_t(kPfBits2Bytes) + size_t(kPfWaves) * kQueue * sizeof(uint64_t) +
                        size_t(kPfWaves) * (kEvBuf * sizeof(PfEvent) + sizeof(uint32_t) + 4 * sizeof(uint32_t));
    typedef void (*ProbeKernel)(PfArgs, ScanGeom, uint32_t, uint64_t, uint32_t, uint32_t, uint32_t*, unsigned long long*);
    const ProbeKernel kern = h.pf_exact2 ? (h.pf_fold ? k_pf_probe<true, true> : k_pf_probe<true, false>)
                                         : (h.pf_fold ? k_pf_probe<false, true> : k_pf_probe<false, false>);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - int(kBitsBytes) - 512);
    if (e != hipSuccess) return e;
    const dim3 grid((kSamples + kPfWaves - 1) / kPfWaves);
    kern<<<grid, dim3(kPfBlock), smem, s>>>(a, g, kSamples, stride, route.cb, route.cr, decision, probe_ctr);
    return hipGetLastError();
}

size_t pf_event_bytes() { return sizeof(PfEvent); }

// totals[0] <- records, totals[1] <- events; rank[] must be all zero on entry (k_ev_write restores that)
hipError_t launch_pf_event_rank(const void* events, const unsigned long long* ev_ctr, uint32_t ev_cap, uint32_t* rank,
                                uint64_t* totals, uint32_t n_hint, hipStream_t s) {
    // the kernel is grid-stride in both dimensions; n_hint (events of the previous call) only sizes the grid
    const uint32_t h = std::min(ev_cap, std::max<uint32_t>(n_hint + n_hint / 4, 4096));
    const dim3 grid((h + 255) / 256, (h + kEvTile - 1) / kEvTile);
    k_ev_rank<<<grid, dim3(256), 0, s>>>(static_cast<const PfEvent*>(events), ev_ctr, ev_cap, rank, totals);
    return hipGetLastError();
}

hipError_t launch_pf_event_write(const HotTables& h, const DevAutomaton& a, const void* events, unsigned long long* ev_ctr,
                                 uint32_t ev_cap, uint32_t* rank, const uint64_t* totals, uint64_t out_cap,
                                 acgpu_match* out, hipStream_t s, void* also_zero, size_t also_zero_bytes) {
    DfaEng eng; eng.d = a.dfa; eng.cls = a.dfa.classes;
    if ((also_zero_bytes & 7) || also_zero_bytes / 8 > 0xFFFFFFFFull) return hipErrorInvalidValue;
    k_ev_write<<<dim3((ev_cap + 255) / 256), dim3(256), 0, s>>>(eng, h.hid2sid, static_cast<const PfEvent*>(events), ev_ctr,
                                                               ev_cap, rank, totals, out_cap, out,
                                                               static_cast<uint64_t*>(also_zero), uint32_t(also_zero_bytes / 8));
    return hipGetLastError();
}

}  // namespace acgpu
