// Prefix filter for LARGE pattern sets (thousands to 131 072 patterns, every pattern >= 4 bytes) and for inputs the
// two-type filter of pf_scan.hip abandons: k_pfx_count (gfx950).  Tables: host/pf_tables.cpp.
//
// Same idea as pf_scan.hip -- the overlapping result is "every occurrence of every pattern" (src/automaton.rs:1491-1534,
// reference DESIGN.md:60-63), so occurrences are enumerated by START position: a cheap filter over every position, an
// exact trie walk for what survives -- but both halves are rebuilt for sets where the 3-byte-key tables of pf_scan.hip
// saturate (100 000 patterns cover every trigram of printable ASCII ten times over):
//
//   level 1 (producer wavefronts, 12 of the 16 in a workgroup)   every position is tested against a 1 Mi-bit "blocked"
//       Bloom table in LDS (128 KiB) keyed by the FOUR bytes b[q..q+3]: one 32-bit multiplicative hash picks the word
//       (the hash folded once, as a byte address) and three bits inside it (selected by bytes 0, 2 and 3 of the hash:
//       SDWA byte operands of the shifts), i.e. ONE LDS gather tests three hash bits.  10 VALU ops + 1 gather per
//       position; 3 % of the positions of a random haystack survive at 100 000 patterns.  The haystack is streamed
//       exactly like in pf_scan.hip (63 x 16 B rows, 4-byte look-ahead from the neighbour lane through DPP, four row-pair
//       register sets in rotation).
//   hand-off   survivors are written -- with their 4-byte window -- to a per-producer LDS ring (ballot / mbcnt ranks, 256
//       entries; the verifier's head is cached, the tail published once per row pair).  Producers never touch global
//       memory beyond their row loads -- no stores, no dependent gathers -- so their s_waitcnt vmcnt(N) software
//       pipeline never drains.  (In pf_scan.hip every batch of 64 survivors is verified inline and ends with
//       s_waitcnt vmcnt(0): fine at 0.03 survivors per row, the whole cost at 40 per row: 17 ms for 8 GiB.)
//   level 2 (verifier wavefronts, 4 per workgroup, each serving three producers)   pop up to 256 survivors at a time
//       (four per lane, their gathers in flight together) and look the exact prefix up in a hash map: the first four
//       bytes -> trie node at depth 4 (HotTables::pfx_map; MODE 0), or, when every pattern has at least five bytes, the
//       first min(8, shortest pattern) bytes -> trie node at that depth (pfx_map8; bytes 4.. fetched from the haystack;
//       8 producers + 8 verifiers).  The 4-byte map (8 MB at 100 000 patterns) misses L2, so in front of it sits an
//       L2-resident exact-prefix BIT table (kGate; HotTables::pf_bits3, 64 bits per pattern): only what passes it is
//       looked up in the map, over dense batches (pfx_resolve).
//   level 3   the trie walk from that node (trie-only transition table), recording pattern ends as events / chunk credits
//       like pf_scan.hip's level 3 (pf_common.hpp): inline over dense batches of 64 queued hits, or -- while hits are at
//       least 1/8 of the survivors -- handed to a second pass over a global hit list (k_pfx_scan_segments, k_pfx_verify:
//       one hit per lane at full occupancy).
//   8-byte level 1 (kKey8: sets whose shortest pattern has 8 bytes -- dictionaries over natural text)   the producers hash
//       the whole 8-byte window (12.6 VALU per position; when every pattern has nine bytes only every other position is
//       probed -- level1_key8x2: one hash and one gather for two starts, 8.25 VALU per position); 0.4 % of the positions of
//       prose survive and 94 % of those are true prefixes, so level 3 is the verifiers' work.  Its cost is the dependent gathers of the DEEPEST walk of a batch (~1.4 us each: the clock
//       profile of -DPFX_PROF), hence: a prefix node below which the trie is one chain to a leaf carries
//       a chain-tail record (HotTables::pfx_tails; level 3 = one round of independent gathers + a masked compare,
//       pfx_verify_tails), and hits are queued by kind -- tail compares in s_hitq, walks in s_slowq -- and verified in
//       batches of their own, between the rounds.
//
// No false negatives by construction (every pattern's first four bytes are in the Bloom table, its exact prefix in the
// map: tests/test_pf_tables.py replays these decisions on the CPU) and every survivor is verified exactly, so the result
// is exact for any input.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "hot.hpp"
#include "launch_util.hpp"
#include "pf_common.hpp"

namespace acgpu {

namespace {

using namespace pfdev;

#ifndef PFX_EXP
#define PFX_EXP 0   // timing experiments only (scripts/pfx_variants.sh): 1 = verifiers drop the survivors, 2 = drop the level-2 hits, 4 = hand-off without its stores (second pass sees no hits), 8 = the 8-byte level 1 loads its rows with plain (temporal) loads, 16 = the long key is read back, the map is not consulted
#endif
#ifndef PFX_PRODUCERS
#define PFX_PRODUCERS 12
#define PFX_VERIFIERS 4
#endif
// wave roles per workgroup (template parameters of the kernel): 12 streaming producers + 4 verifiers for the 4-byte level 2
// (random text against 100 000 patterns: 3 % of the positions survive, 4 % of those hit); 8 + 8 for the long-prefix
// level 2, whose verifiers do two dependent gathers per survivor on inputs where 7 % of the positions survive
// (measured on English text: 12+4 1.77 ms, 10+5 1.63 ms, 8+8 1.28 ms per GiB)
#ifndef PFX_LONG_PRODUCERS
#define PFX_LONG_PRODUCERS 8
#define PFX_LONG_VERIFIERS 8
#endif
// -DPFX_PROF=1 (timing experiments, lib/exp only): where the wavefronts of k_pfx_count spend their clocks.  g_pfx_prof:
// [0] producer clocks, [1] of which waiting for ring room, [2] such waits; [3] verifier clocks, [4] idle (nothing to pop),
// [5] levels 1-2 of its rounds, [6] level 3 (drain_hits), [7] rounds, [8] survivors popped, [9] level-3 batches, [10] hits verified,
// [11] clocks in the event flushes, [12] trips of the walk loop (the lane with the most), [13] clocks / [14] batches / [15] hits
// of the slow queue's walks
#ifdef PFX_PROF
__device__ unsigned long long g_pfx_prof[16];
#define PFX_CLOCK() clock64()
#define PFX_PROF_ADD(i, v) atomicAdd(&g_pfx_prof[i], static_cast<unsigned long long>(v))
#else
#define PFX_CLOCK() 0ull
#define PFX_PROF_ADD(i, v) ((void)(v))
#endif
#ifndef PFX_RING
#define PFX_RING 256
#endif
constexpr int kXQueue = PFX_RING;                             // ring entries per producer: {4-byte window, position}, 8 bytes; the position alone (4 bytes) under the 8-byte level 1
constexpr int kXBatch = 4;                                 // survivors per verifier lane per round

// LDS words shared between wavefronts (ring indices, done flags).  Explicit address space: through a generic `volatile`
// pointer the compiler emits flat_load/flat_store (VMEM-counted), which drains the producers' row prefetch at every use.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint64_t lds_u64;
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p) {   // wave-uniform read of a word another wavefront writes
    return uint32_t(__builtin_amdgcn_readfirstlane(int(*(volatile lds_u32*)(p))));
}
__device__ __forceinline__ void lds_poke(uint32_t* p, uint32_t v) { *(volatile lds_u32*)(p) = v; }

// Per-wavefront producer state.
template <int kQ>   // ring entries
struct PfxProducer {
    const PfArgs& a;
    const ScanGeom& g;
    const uint32_t* s_bits;        // 128 KiB blocked Bloom table (static LDS at offset 0)
    uint64_t* ring;                // this producer's ring: {4-byte window, task sequence << 16 | offset in the task}
    uint32_t* tail;                // entries published (written by this wave, read by its verifier)
    uint32_t* head;                // entries consumed (written by the verifier)
    uint32_t* task_seq_pub;        // LDS word where the current task sequence number is published
    uint32_t task_seq = 0;         // k: this producer's k-th task
    uint32_t tail_local = 0;       // wave-uniform copy of *tail
    unsigned long long prof_stall = 0, prof_stalls = 0;   // (PFX_PROF)
    uint32_t head_cached = 0;      // wave-uniform: the last value of *head this wave has read (<= the real one)
    uint64_t task_base = 0;
    uint4 ra[kSets] = {}, rb[kSets] = {};
    bool carried = false;
    int lane = 0;

    static __device__ __forceinline__ uint32_t uni(uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); }

    // word << (byte B of h & 31): the shift amount is an SDWA byte operand
    template <int B>
    static __device__ __forceinline__ uint32_t shl_by_byte(uint32_t word, uint32_t h) {
        uint32_t r;
        if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(h), "v"(word));
        else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(h), "v"(word));
        else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(h), "v"(word));
        return r;
    }
    // 16 survivor bits of one row: bit 15-q <=> start position q of this lane's 16 bytes (wd[4] = look-ahead dword).
    // Per position: window (alignbit), hash (mul_u24 + mad_u24), word address (lshr + bitop3), gather, three SDWA shifts,
    // and3, alignbit into the mask: 10 VALU operations.
    __device__ __forceinline__ uint32_t level1(const uint32_t (&wd)[5]) const {
        uint32_t hits = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t word[8], hh[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int q = half * 8 + j, i = q >> 2, r = q & 3;
                const uint32_t w = r == 0 ? wd[i] : __builtin_amdgcn_alignbit(wd[i + 1], wd[i], 8 * r);
                const uint32_t h = pfx_hash(w);
                hh[j] = h;
                word[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s_bits) + pfx_word_addr(h));
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                // the three selected bits moved to bit 31
                const uint32_t t = shl_by_byte<0>(word[j], hh[j]) & shl_by_byte<2>(word[j], hh[j]) & shl_by_byte<3>(word[j], hh[j]);
                hits = __builtin_amdgcn_alignbit(hits, t, 31);
            }
        }
        return hits;
    }

    // The same test with the EIGHT-byte key (wd[4], wd[5] = the neighbour lane's first two dwords): two windows and a
    // four-operation hash per position instead of one and two -- 13 VALU operations -- for sets whose 4-byte prefixes are
    // everywhere in the text and whose 8-byte prefixes are not (English prose against a dictionary: 7 % of the positions
    // survive the 4-byte key).  Bytes past the span read as whatever the hull holds: a survivor is verified exactly.
    __device__ __forceinline__ uint32_t level1_key8(const uint32_t (&wd)[6]) const {
        uint32_t hits = 0;
        const uint32_t himask = a.xdepth >= 8 ? 0xFFFFFFFFu : (1u << (8 * (a.xdepth - 4))) - 1u;   // prefixes of 5..7 bytes: the rest of the window is not key
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t word[8], hh[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int q = half * 8 + j, i = q >> 2, r = q & 3;
                const uint32_t lo = r == 0 ? wd[i] : __builtin_amdgcn_alignbit(wd[i + 1], wd[i], 8 * r);
                const uint32_t hi = (r == 0 ? wd[i + 1] : __builtin_amdgcn_alignbit(wd[i + 2], wd[i + 1], 8 * r)) & himask;
                const uint32_t h = pfx_hash8(lo, hi);
                hh[j] = h;
                word[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s_bits) + pfx_word_addr(h));
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t t = shl_by_byte<0>(word[j], hh[j]) & shl_by_byte<2>(word[j], hh[j]) & shl_by_byte<3>(word[j], hh[j]);
                hits = __builtin_amdgcn_alignbit(hits, t, 31);
            }
        }
        return hits;
    }

    // ---- the 8-byte level 1 probing only the ODD offsets p = 1, 3, .., 15 of a lane's 16 bytes (hot.hpp: pfx_x2_mask; sets
    // whose shortest pattern has nine bytes): key = b[p+1..p+8]; the entry's two bits selected by b[p] admit a start at p
    // ("type 0"), the two selected by b[p+9] a start at p + 1 ("type 1").  Per probe: at most one new window (alignbit by 16:
    // the keys begin at even offsets), the four-operation hash, two operations for the word address, the gather, and per
    // type one SDWA shift by the selector byte, one SDWA byte add (selector + a byte of the hash), one shift by the sum,
    // one and, one alignbit into the mask: 16.5 VALU for two start positions where level1_key8 spends 25.
    // Returns 16 bits: bit 16-q <=> start position q = 1..16 of this lane's row (16 = position 0 of the next lane).
    template <int B>
    static __device__ __forceinline__ uint32_t shl_by_byte_of(uint32_t word, uint32_t reg) {   // word << (byte B of reg & 31)
        uint32_t r;
        if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(reg), "v"(word));
        else if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(reg), "v"(word));
        else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(reg), "v"(word));
        else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(reg), "v"(word));
        return r;
    }
    template <int BA, int BB>
    static __device__ __forceinline__ uint32_t add_bytes(uint32_t ra, uint32_t rb) {   // byte BA of ra + byte BB of rb
        uint32_t r;
#define ACGPU_ADD_SDWA(SA, SB) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" SA " src1_sel:" SB : "=v"(r) : "v"(ra), "v"(rb))
        if (BB == 0) {
            if (BA == 0) ACGPU_ADD_SDWA("BYTE_0", "BYTE_0"); else if (BA == 1) ACGPU_ADD_SDWA("BYTE_1", "BYTE_0");
            else if (BA == 2) ACGPU_ADD_SDWA("BYTE_2", "BYTE_0"); else ACGPU_ADD_SDWA("BYTE_3", "BYTE_0");
        } else {
            if (BA == 0) ACGPU_ADD_SDWA("BYTE_0", "BYTE_2"); else if (BA == 1) ACGPU_ADD_SDWA("BYTE_1", "BYTE_2");
            else if (BA == 2) ACGPU_ADD_SDWA("BYTE_2", "BYTE_2"); else ACGPU_ADD_SDWA("BYTE_3", "BYTE_2");
        }
#undef ACGPU_ADD_SDWA
        return r;
    }
    template <int J>
    __device__ __forceinline__ void x2_probe(const uint32_t (&wd)[7], uint32_t& h, uint32_t& word) const {
        constexpr int k0 = 2 * J + 2, i = k0 >> 2;   // the key begins at the even offset k0 = p + 1
        const uint32_t lo = (k0 & 2) ? __builtin_amdgcn_alignbit(wd[i + 1], wd[i], 16) : wd[i];
        const uint32_t hi = (k0 & 2) ? __builtin_amdgcn_alignbit(wd[i + 2], wd[i + 1], 16) : wd[i + 1];
        h = pfx_hash8(lo, hi);
        word = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s_bits) + pfx_word_addr(h));
    }
    template <int J>
    __device__ __forceinline__ uint32_t x2_fold(uint32_t hits, const uint32_t (&wd)[7], uint32_t h, uint32_t word) const {
        constexpr int p = 2 * J + 1, s1 = p + 9;   // selector offsets: b[p] (byte 1 or 3 of its dword), b[p + 9] (byte 0 or 2)
        const uint32_t t0 = shl_by_byte_of<p & 3>(word, wd[p >> 2]) & (word << (add_bytes<p & 3, 0>(wd[p >> 2], h) & 31u));
        const uint32_t t1 = shl_by_byte_of<s1 & 3>(word, wd[s1 >> 2]) & (word << (add_bytes<s1 & 3, 2>(wd[s1 >> 2], h) & 31u));
        hits = __builtin_amdgcn_alignbit(hits, t0, 31);
        return __builtin_amdgcn_alignbit(hits, t1, 31);
    }
    __device__ __forceinline__ uint32_t level1_key8x2(const uint32_t (&wd)[7]) const {
        uint32_t h[8], word[8];
        x2_probe<0>(wd, h[0], word[0]); x2_probe<1>(wd, h[1], word[1]); x2_probe<2>(wd, h[2], word[2]); x2_probe<3>(wd, h[3], word[3]);
        x2_probe<4>(wd, h[4], word[4]); x2_probe<5>(wd, h[5], word[5]); x2_probe<6>(wd, h[6], word[6]); x2_probe<7>(wd, h[7], word[7]);
        uint32_t hits = 0;
        hits = x2_fold<0>(hits, wd, h[0], word[0]); hits = x2_fold<1>(hits, wd, h[1], word[1]); hits = x2_fold<2>(hits, wd, h[2], word[2]);
        hits = x2_fold<3>(hits, wd, h[3], word[3]); hits = x2_fold<4>(hits, wd, h[4], word[4]); hits = x2_fold<5>(hits, wd, h[5], word[5]);
        hits = x2_fold<6>(hits, wd, h[6], word[6]); hits = x2_fold<7>(hits, wd, h[7], word[7]);
        return hits;
    }

    // the window b[k..k+3] of a lane's row registers, k = 0..15 dynamic (cndmask tree + funnel shift)
    static __device__ __forceinline__ uint32_t window(const uint32_t (&wd)[5], uint32_t k) {
        const bool up = (k & 8u) != 0, mid = (k & 4u) != 0;
        const uint32_t c0 = up ? wd[2] : wd[0], c1 = up ? wd[3] : wd[1], c2 = up ? wd[4] : wd[2];
        return __builtin_amdgcn_alignbit(mid ? c2 : c1, mid ? c1 : c0, 8u * (k & 3u));
    }

    // publishes the start positions whose bit is set in hits32 (rows 2k, 2k+1 of a pair: bits 31..16, 15..0), each with
    // its 4-byte window (taken from the row registers: the verifier's exact test needs no look at the haystack).
    // The verifier's `head` is cached and re-read only when the ring looks full, and the new tail is published once per
    // call (and before every wait for room): the LDS round trip and fence per survivor iteration were a third of the loop.
    // SHORT (short mode, hot.hpp): the task sequence takes 15 bits of an entry, bit 31 says "a straggler's first bytes begin
    // here" (flag) -- level 2 compares the stragglers instead of looking the prefix up
    template <bool GUARD, bool KEY8, bool X2 = false, bool SHORT = false>   // X2: bit 15-i of a row's mask stands for start position i + 1 (level1_key8x2)
    __device__ __forceinline__ void push(uint32_t hits32, uint32_t off, const uint32_t (&w0)[6], const uint32_t (&w1)[6], bool flag = false) {
        // (Round 4 tried a wave-level compaction here -- survivor counts prefix-summed by three ballots, one room check, each
        // lane storing its own entries -- on the premise that this loop's ballot / rank / room check per trip was a third of
        // the producers' work: config 4 4.468 vs 4.463 ms, natural text 0.707 vs 0.692 ms per GiB: nothing.  It also needed
        // room for up to 256 entries at once, which the verifier -- it waits for batches of 64 -- would never make: removed.)
        bool dirty = false;   // wave-uniform: entries written since the tail was last published
        while (__any(hits32 != 0)) {
            const bool has = hits32 != 0;
            const uint32_t b = uint32_t(__builtin_ctz(hits32 | 0x80000000u));   // lowest set bit first
            hits32 &= hits32 - 1;
            const uint32_t idx = 31u - b;   // bit b of hits32 <=> row idx >> 4 of the pair, start position idx & 15
            const bool second = (idx >> 4) != 0;
            const uint32_t toff = off + (second ? kRowBytes : 0u) + (idx & 15u) + (X2 ? 1u : 0u);   // <= 40 320: fits 16 bits
            bool ok = has;
            if (GUARD) {   // (an interior task owns every start position of its rows)
                const uint64_t v = task_base + toff;
                ok = has && v >= a.scan_lo && v < g.emit_hi;
            }
            // (8-byte level 1: level 2 reads the whole prefix back from the haystack, so the entry is the position alone --
            // four bytes, 256 entries per ring in the LDS the 128 eight-byte ones took: room for the bursts of natural text.
            // Round 6 queued the 8-byte key instead, taken from the row registers: fabric traffic 1.59x -> 1.33x of the haystack,
            // kernel 3-4 % SLOWER -- docs/experiments/r06_pfx_key_in_ring.md)
            uint64_t entry = uint64_t((task_seq << 16) | toff) << 32;
            if (SHORT) entry = uint64_t((((task_seq & 0x7FFFu) | (flag ? 0x8000u : 0u)) << 16) | toff) << 32;
            if (!KEY8) {
                const uint32_t wd[5] = {second ? w1[0] : w0[0], second ? w1[1] : w0[1], second ? w1[2] : w0[2],
                                        second ? w1[3] : w0[3], second ? w1[4] : w0[4]};
                entry |= uint64_t(window(wd, idx & 15u));
            }
            const unsigned long long m = __ballot(ok);
            if (m == 0) continue;
            const uint32_t n = uint32_t(__popcll(m));
            // room for n entries?  (the verifier advances *head; spin while the ring is full)
            if (tail_local + n - head_cached > uint32_t(kQ)) {
                if (dirty) { pf_fence(); if (lane == 0) lds_poke(tail, tail_local); dirty = false; }   // let the verifier see what is there
                const unsigned long long w0 = PFX_CLOCK();
                while (tail_local + n - (head_cached = lds_peek(head)) > uint32_t(kQ)) __builtin_amdgcn_s_sleep(2);
                prof_stall += PFX_CLOCK() - w0; prof_stalls++;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
            if (KEY8) { if (ok) *(lds_u32*)(reinterpret_cast<uint32_t*>(ring) + ((tail_local + rank) & uint32_t(kQ - 1))) = uint32_t(entry >> 32); }
            else if (ok) *(lds_u64*)(&ring[(tail_local + rank) & uint32_t(kQ - 1)]) = entry;
            tail_local += n;
            dirty = true;
        }
        if (dirty) {
            pf_fence();                                  // entries visible ...
            if (lane == 0) lds_poke(tail, tail_local);   // ... before the new tail
        }
    }

    // the scan's very first start position has no probe in front of it under level1_key8x2: handed to level 2 as it is
    __device__ __forceinline__ void push_first(bool with_short = false) {
        if (lane == 0) *(lds_u32*)(reinterpret_cast<uint32_t*>(ring) + (tail_local & uint32_t(kQ - 1))) = (task_seq << 16);
        tail_local += 1;
        if (with_short) {   // (the same position as a straggler's start)
            if (lane == 0) *(lds_u32*)(reinterpret_cast<uint32_t*>(ring) + (tail_local & uint32_t(kQ - 1))) = (task_seq << 16) | 0x80000000u;
            tail_local += 1;
        }
        pf_fence();
        if (lane == 0) lds_poke(tail, tail_local);
    }

    // the start positions 1..16 of a lane's row registers where a straggler's first min(len, 4) bytes stand (short mode):
    // bit 15 - i <=> position i + 1, the layout of level1_key8x2's masks
    __device__ __forceinline__ uint32_t short_starts(const uint32_t (&w)[6]) const {
        const uint32_t k0 = a.short_lo[0] & (a.short_len[0] >= 4 ? 0xFFFFFFFFu : 0xFFFFFFu), m0 = a.short_len[0] >= 4 ? 0xFFFFFFFFu : 0xFFFFFFu;
        uint32_t m = 0;
        if (a.short_n == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int p = i + 1, d = p >> 2, r = p & 3;
                const uint32_t win = r ? __builtin_amdgcn_alignbit(w[d + 1], w[d], 8 * r) : w[d];
                m = (m << 1) | ((win & m0) == k0 ? 1u : 0u);
            }
        } else {
            const uint32_t m1 = a.short_len[1] >= 4 ? 0xFFFFFFFFu : 0xFFFFFFu, k1 = a.short_lo[1] & m1;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int p = i + 1, d = p >> 2, r = p & 3;
                const uint32_t win = r ? __builtin_amdgcn_alignbit(w[d + 1], w[d], 8 * r) : w[d];
                m = (m << 1) | (((win & m0) == k0 || (win & m1) == k1) ? 1u : 0u);
            }
        }
        return m;
    }

    template <bool GUARD, bool KEY8, bool X2 = false, bool SHORT = false>
    __device__ __forceinline__ void run_task(uint64_t tb, uint64_t next_base, bool next_interior) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        auto load_plain = [&](uint64_t p, uint4& w) {
            ACGPU_HAY_CHECK(g, p, 16);
            if ((PFX_EXP & 8) && KEY8) { w = *reinterpret_cast<const uint4*>(g.hay16 + p); return; }
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
        task_base = tb;
        if (lane == 0) lds_poke(task_seq_pub, task_seq);   // (entries of this task are published after this word)
        uint64_t p = tb + uint64_t(lane) * 16;
        uint32_t off = uint32_t(lane) * 16;
        if (!carried) {
#pragma unroll
            for (int j = 0; j < kSets - 1; j++) {
                load(p + uint64_t(2 * j) * kRowBytes, ra[j]);
                load(p + uint64_t(2 * j + 1) * kRowBytes, rb[j]);
            }
        }
        carried = false;
        auto pair = [&](const uint4& wa, const uint4& wb) {
            const uint32_t w0[6] = {wa.x, wa.y, wa.z, wa.w, uint32_t(__builtin_amdgcn_update_dpp(0, int(wa.x), 0x130, 0xF, 0xF, false)),
                                    KEY8 ? uint32_t(__builtin_amdgcn_update_dpp(0, int(wa.y), 0x130, 0xF, 0xF, false)) : 0u};
            const uint32_t w1[6] = {wb.x, wb.y, wb.z, wb.w, uint32_t(__builtin_amdgcn_update_dpp(0, int(wb.x), 0x130, 0xF, 0xF, false)),
                                    KEY8 ? uint32_t(__builtin_amdgcn_update_dpp(0, int(wb.y), 0x130, 0xF, 0xF, false)) : 0u};
            uint32_t hits32;
            if (X2) {
                const uint32_t x0[7] = {w0[0], w0[1], w0[2], w0[3], w0[4], w0[5], uint32_t(__builtin_amdgcn_update_dpp(0, int(wa.z), 0x130, 0xF, 0xF, false))};
                const uint32_t x1[7] = {w1[0], w1[1], w1[2], w1[3], w1[4], w1[5], uint32_t(__builtin_amdgcn_update_dpp(0, int(wb.z), 0x130, 0xF, 0xF, false))};
                hits32 = (level1_key8x2(x0) << 16) | (level1_key8x2(x1) & 0xFFFFu);
            } else if (KEY8) {
                hits32 = (level1_key8(w0) << 16) | (level1_key8(w1) & 0xFFFFu);
            } else {
                const uint32_t v0[5] = {w0[0], w0[1], w0[2], w0[3], w0[4]}, v1[5] = {w1[0], w1[1], w1[2], w1[3], w1[4]};
                hits32 = (level1(v0) << 16) | (level1(v1) & 0xFFFFu);
            }
            if (lane == 63) hits32 = 0;   // lane 63's 16 bytes are lane 0 of the next row
            push<GUARD, KEY8, X2, SHORT>(hits32, off, w0, w1);
            if (SHORT) {
                uint32_t shorts32 = (short_starts(w0) << 16) | short_starts(w1);
                if (lane == 63) shorts32 = 0;
                push<GUARD, KEY8, X2, SHORT>(shorts32, off, w0, w1, true);
            }
            p += 2 * kRowBytes;
            off += 2 * kRowBytes;
        };
        constexpr uint32_t kPairs = kTaskRows / 2;
        const uint64_t next_p = next_base + uint64_t(lane) * 16;
        bool completed = true;
#pragma unroll 1
        for (uint32_t r = 0; r < kTaskRows; r += 2 * kSets) {
            if (GUARD && tb + uint64_t(r) * kRowBytes >= g.emit_hi) { completed = false; break; }
#pragma unroll
            for (int j = 0; j < kSets; j++) {
                constexpr int kAhead = kSets - 1;
                const int n = (j + kAhead) % kSets;
                const uint32_t pi = r / 2 + uint32_t(j + kAhead);
                uint64_t src = p + uint64_t(2 * kAhead) * kRowBytes;
                if (pi >= kPairs) src = next_interior ? next_p + uint64_t(pi - kPairs) * (2 * kRowBytes) : p;
                if (GUARD && !(pi >= kPairs && next_interior)) {
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
        task_seq++;
    }
};

// One buffered event in LDS: the key and the trie node (its own count is looked up by the flush) -- 12 bytes, so that 128 of
// them per verifier leave room for the second hit queue.
struct PfxEv {
    uint32_t key_lo, key_hi, node;
    __device__ __forceinline__ void set(uint64_t key, uint32_t n) { key_lo = uint32_t(key); key_hi = uint32_t(key >> 32); node = n; }
    __device__ __forceinline__ uint64_t key() const { return uint64_t(key_lo) | (uint64_t(key_hi) << 32); }
};

// Hit-queue / hit-list entry (64 bits): low 32 bits of rel = v - row0; high word: value (20 bits) | own flag << 20 | tail
// flag << 21 | high 10 bits of rel << 22.  value = the trie node at the prefix depth, or -- tail flag -- the index of the
// node's chain-tail record (hot.hpp).
__device__ __forceinline__ uint64_t pfx_hit_entry(uint64_t rel, uint32_t value20, bool own, bool tail) {
    return uint64_t(uint32_t(rel)) |
           (uint64_t((value20 & 0xFFFFFu) | (own ? 1u << 20 : 0u) | (tail ? 1u << 21 : 0u) | (uint32_t(rel >> 32) << 22)) << 32);
}
__device__ __forceinline__ void pfx_hit_decode(const PfArgs& a, uint64_t ent, uint64_t& v, uint32_t& node, uint32_t& tail1) {
    const uint32_t hi = uint32_t(ent >> 32);
    v = a.row0 + (uint64_t(uint32_t(ent)) | (uint64_t(hi >> 22) << 32));
    const bool is_tail = (hi >> 21) & 1u;
    node = is_tail ? 0u : (hi & 0xFFFFFu) | (((hi >> 20) & 1u) << 31);
    tail1 = is_tail ? (hi & 0xFFFFFu) + 1u : 0u;   // index + 1 of the tail record, 0 = walk from `node`
}

// a pattern of trie node `node` (own count cnt) ends with byte `end_at` of a match that starts at v: event or chunk credit
template <int kCap>
__device__ __forceinline__ bool pfx_record(const PfArgs& a, const ScanGeom& g, uint32_t* counts, uint64_t v, uint64_t end_at, uint32_t node,
                                           uint32_t cnt, PfxEv* ebuf, uint32_t* ecnt) {
    if (end_at < g.emit_lo || end_at >= g.emit_hi) return false;
    if (a.events) {
        const uint64_t key = ((end_at + 1 - g.base_mis) << 16) | (0xFFFFull - (end_at + 1 - v));
        const uint32_t slot = atomicAdd(ecnt, 1u);
        if (slot < uint32_t(kCap)) { ebuf[slot].set(key, node); return true; }
        pf_append_event(a, key, node, cnt);
        return false;
    }
    atomicAdd(&counts[(end_at - g.grid0) / g.chunk], cnt);
    return false;
}

// level 3 from depth a.xdepth (4, or up to 8 with the long-prefix map): `node` = trie node reached by b[v..v+depth-1]
// (bit 31: a pattern ends there); same bookkeeping as pf_verify (pf_common.hpp)
template <bool kWide = false, int kCap = kEvBuf>   // kCap: entries of the wavefront's LDS event buffer
__device__ __forceinline__ bool pfx_verify_from(const PfArgs& a, const ScanGeom& g, uint32_t* counts, uint64_t v, uint32_t node,
                                                PfxEv* ebuf, uint32_t* ecnt, const uint8_t* s_acls) {
    uint32_t s = node & 0x7FFFFFFFu;
    bool buffered = false;
    auto record = [&](uint64_t at) {   // a pattern ends with byte `at`
        if (at < g.emit_lo || at >= g.emit_hi) return;
        if (a.events) {
            // (the node's own count is looked up by the flush, for all buffered events at once: read here it was a second
            // dependent gather in every step of the walk that ends a pattern -- with 64-128 walks in lockstep, nearly every step)
            const uint64_t key = ((at + 1 - g.base_mis) << 16) | (0xFFFFull - (at + 1 - v));
            const uint32_t slot = atomicAdd(ecnt, 1u);
            if (slot < uint32_t(kCap)) { ebuf[slot].set(key, s); buffered = true; }
            else pf_append_event(a, key, s, a.own_cnt[s]);
        } else {
            atomicAdd(&counts[(at - g.grid0) / g.chunk], a.own_cnt[s]);
        }
    };
    if (node >> 31) record(v + a.xdepth - 1);
    uint64_t at = v + a.xdepth;
    if (kWide && at + 16 <= g.emit_hi) {   // the next 16 haystack bytes in ONE gather (the walk rarely needs more)
        uint32_t w[4];
        ACGPU_HAY_CHECK(g, at, 16);
        __builtin_memcpy(w, g.hay16 + at, 16);
#pragma unroll
        for (int k = 0; k < 16; k++, at++) {
            const uint32_t e = a.atab[(s << a.ashift) | s_acls[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu]];
            if (e == 0) return buffered;
            s = e & 0x7FFFFFFFu;
            if (e >> 31) record(at);
        }
    }
    for (; at < g.emit_hi; at++) {
        ACGPU_HAY_CHECK(g, at, 1);
        const uint32_t e = a.atab[(s << a.ashift) | s_acls[g.hay16[at]]];
        if (e == 0) break;
        s = e & 0x7FFFFFFFu;
        if (e >> 31) record(at);
    }
    return buffered;
}

// Level 3 through a chain-tail record (hot.hpp): below the prefix node the trie is one chain down to a leaf, so the walk's
// dependent trie-row gathers are replaced by ONE gather of the chain's bytes beside the 16 haystack bytes, and a masked
// compare.  tail1 = record index + 1.
template <int kCap, bool kWide = false>
__device__ __forceinline__ bool pfx_verify_tail(const PfArgs& a, const ScanGeom& g, uint32_t* counts, uint64_t v, uint32_t tail1,
                                                PfxEv* ebuf, uint32_t* ecnt, const uint8_t* s_acls) {
    const uint4* rec = reinterpret_cast<const uint4*>(a.tails + size_t(tail1 - 1) * kPfxTailWords);
    uint4 t0 = rec[0], t1 = rec[1];   // bytes | pattern-end node, length | count << 8, prefix node, records that follow
    const uint64_t at = v + a.xdepth;
    if (at + 16 > g.emit_hi)   // the last bytes of the span: the walk from the prefix node
        return pfx_verify_from<kWide, kCap>(a, g, counts, v, t1.z, ebuf, ecnt, s_acls);
    uint32_t h[4];
    ACGPU_HAY_CHECK(g, at, 16);
    __builtin_memcpy(h, g.hay16 + at, 16);
    bool buffered = false;
    for (;;) {   // the node's records: neighbours in memory (the next one is on its way while this one is compared)
        const bool more = t1.w != 0;
        uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
        if (more) { n0 = rec[2]; n1 = rec[3]; }
        const uint32_t tl = t1.y & 0xFFu;
        const uint32_t tb[4] = {t0.x, t0.y, t0.z, t0.w};
        uint32_t diff = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t nb = tl > 4u * d ? (tl - 4u * d < 4u ? tl - 4u * d : 4u) : 0u;
            const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (1u << (8 * nb)) - 1u;
            diff |= (h[d] ^ tb[d]) & m;
        }
        if (diff == 0) buffered |= pfx_record<kCap>(a, g, counts, v, at + tl - 1, t1.x, t1.y >> 8, ebuf, ecnt);
        if (!more) break;
        t0 = n0; t1 = n1; rec += 2;
    }
    return buffered;
}

// N tail compares per lane (the fast queue of the 8-byte level 1): every gather of the batch -- N tail records, N x 16
// haystack bytes -- is independent of the others, so a batch costs ONE memory latency where the walks of
// pfx_verify_n_from cost one per trie level of their deepest lane.
template <int N, int kCap>
__device__ __forceinline__ bool pfx_verify_tails(const PfArgs& a, const ScanGeom& g, uint32_t* counts, const uint64_t (&v)[N],
                                                 const uint32_t (&tail1)[N], PfxEv* ebuf, uint32_t* ecnt, const uint8_t* s_acls) {
    bool buffered = false;
    uint4 t0[N], t1[N], h[N];
    bool have[N], wide[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        have[i] = tail1[i] != 0;
        wide[i] = have[i] && v[i] + a.xdepth + 16 <= g.emit_hi;
        t0[i] = t1[i] = h[i] = make_uint4(0, 0, 0, 0);
        if (have[i]) {
            const uint4* rec = reinterpret_cast<const uint4*>(a.tails + size_t(tail1[i] - 1) * kPfxTailWords);
            t0[i] = rec[0]; t1[i] = rec[1];
        }
        if (wide[i]) {
            ACGPU_HAY_CHECK(g, v[i] + a.xdepth, 16);
            __builtin_memcpy(&h[i], g.hay16 + v[i] + a.xdepth, 16);
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (!have[i]) continue;
        if (!wide[i]) {   // the last bytes of the span: the walk from the prefix node (no pattern ends there itself)
            buffered |= pfx_verify_from<true, kCap>(a, g, counts, v[i], t1[i].z, ebuf, ecnt, s_acls);
            continue;
        }
        const uint32_t tl = t1[i].y & 0xFFu;
        const uint32_t hb[4] = {h[i].x, h[i].y, h[i].z, h[i].w}, tb[4] = {t0[i].x, t0[i].y, t0[i].z, t0[i].w};
        uint32_t diff = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t nb = tl > 4u * d ? (tl - 4u * d < 4u ? tl - 4u * d : 4u) : 0u;
            const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (1u << (8 * nb)) - 1u;
            diff |= (hb[d] ^ tb[d]) & m;
        }
        if (diff == 0) buffered |= pfx_record<kCap>(a, g, counts, v[i], v[i] + a.xdepth + tl - 1, t1[i].x, t1[i].y >> 8, ebuf, ecnt);
    }
    return buffered;
}

// N level-3 walks per lane in lockstep (the inline level 3 under the 8-byte level 1, where nineteen survivors out of
// twenty are true prefixes with a walk ahead of them): the walks are chains of dependent gathers (haystack bytes, then one
// trie row per byte), so a verifier wavefront's throughput is the number of walks it keeps in flight.
template <int N, int kCap>
__device__ __forceinline__ bool pfx_verify_n_from(const PfArgs& a, const ScanGeom& g, uint32_t* counts, const uint64_t (&v)[N],
                                                  const uint32_t (&node)[N], PfxEv* ebuf, uint32_t* ecnt, const uint8_t* s_acls,
                                                  unsigned long long* prof_steps = nullptr) {
    bool buffered = false;
    uint32_t s[N];
    uint64_t at[N];
    bool live[N];
    auto record = [&](int i, uint64_t end_at) {   // a pattern ends with byte `end_at` (bookkeeping of pfx_verify_from)
        if (end_at < g.emit_lo || end_at >= g.emit_hi) return;
        if (a.events) {   // (own count: see pfx_verify_from)
            const uint64_t key = ((end_at + 1 - g.base_mis) << 16) | (0xFFFFull - (end_at + 1 - v[i]));
            const uint32_t slot = atomicAdd(ecnt, 1u);
            if (slot < uint32_t(kCap)) { ebuf[slot].set(key, s[i]); buffered = true; }
            else pf_append_event(a, key, s[i], a.own_cnt[s[i]]);
        } else {
            atomicAdd(&counts[(end_at - g.grid0) / g.chunk], a.own_cnt[s[i]]);
        }
    };
    // the 16 bytes behind the prefix in ONE gather per walk (as two 64-bit halves: the byte of step k is picked by shifts, so
    // the step loop stays rolled -- unrolled, with the bookkeeping of `record` inlined 32 times, the kernel grew to 67 000
    // instructions and lost more in the instruction cache than the lockstep gained); steps beyond them, and walks that
    // begin within 16 bytes of the span's end, read single bytes
    uint64_t wlo[N], whi[N];
    bool wide[N];
    bool any_live = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
        wlo[i] = 0; whi[i] = 0;
        live[i] = node[i] != 0;
        s[i] = node[i] & 0x7FFFFFFFu;
        at[i] = v[i] + a.xdepth;
        wide[i] = live[i] && at[i] + 16 <= g.emit_hi;
        if (wide[i]) {
            ACGPU_HAY_CHECK(g, at[i], 16);
            uint64_t t[2];
            __builtin_memcpy(t, g.hay16 + at[i], 16);
            wlo[i] = t[0]; whi[i] = t[1];
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (live[i] && (node[i] >> 31)) record(i, v[i] + a.xdepth - 1);
        any_live |= live[i];
    }
#pragma unroll 1
    for (uint32_t k = 0; any_live; k++) {
#ifdef PFX_PROF
        if (prof_steps && __builtin_amdgcn_ballot_w64(any_live) != 0) prof_steps[0]++;   // (trips of the wavefront)
#endif
        uint32_t e[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            e[i] = 0;
            if (!live[i]) continue;
            uint32_t byte;
            if (wide[i] && k < 16) byte = uint32_t(((k & 8u) ? whi[i] : wlo[i]) >> (8u * (k & 7u))) & 0xFFu;
            else if (at[i] + k < g.emit_hi) { ACGPU_HAY_CHECK(g, at[i] + k, 1); byte = g.hay16[at[i] + k]; }
            else { live[i] = false; continue; }
            e[i] = a.atab[(s[i] << a.ashift) | s_acls[byte]];
        }
        any_live = false;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (!live[i]) continue;
            if (e[i] == 0) { live[i] = false; continue; }
            s[i] = e[i] & 0x7FFFFFFFu;
            if (e[i] >> 31) record(i, at[i] + k);
            any_live = true;
        }
    }
    return buffered;
}

// Level 2 of a start the bit-table gate let through (k_pfx_count<false, .., kGate = true>): the exact first four bytes ->
// trie node at depth 4 from the hash map (0 = a false positive of the bit table).  The key is read back from the haystack:
// the gate's hand-off carries the position only.
__device__ __forceinline__ uint32_t pfx_resolve(const PfArgs& a, const ScanGeom& g, uint64_t v) {
    if (v + 4 > g.emit_hi) return 0;   // (every pattern has at least four bytes)
    uint32_t key;
    ACGPU_HAY_CHECK(g, v, 4);
    __builtin_memcpy(&key, g.hay16 + v, 4);
    const uint32_t mask = (1u << a.xmap_log2) - 1;
    for (uint32_t bk = pfx_map_bucket(key, a.xmap_log2);; bk = (bk + 1) & mask) {
        const uint4 q = a.xmap[bk];
        if (q.y && q.x == key) return q.y & ~kPfxMapOverflow;
        if (q.w && q.z == key) return q.w;
        if (!(q.y & kPfxMapOverflow)) return 0;
    }
}

// appends the verifier's buffered events to the global list if at least `at_least` are waiting (wave-uniform).  ONE pair of
// global atomics per flush: every flush of every wavefront adds to the same two words, which one L2 channel serialises at
// ~7 ns each -- a million events in batches of 24-48 cost more than the scan (natural text: k_pfx_verify 0.84 ms per GiB
// of which ~0.5 ms atomics), hence buffers of kCap >= 192 entries where LDS allows.
// The order pass's histogram on the way (PfEoHist): the bucket atomics of a flush return the events' arrival slots a few
// microseconds later; the wavefront does not wait for them -- the slots are stored by its NEXT flush (or pfx_flush_slots at its
// end).  (Measured: the scan takes the same 17-20 us longer either way -- it is the rate of the atomics, not their latency.)
template <int kCap>
struct PfxEoPend {
    static constexpr int kSlices = (kCap + 63) / 64;
    uint32_t r[kSlices] = {};
    unsigned long long base = 0;
    uint32_t n = 0;   // wave-uniform: events of the last flush whose slots are not stored yet
};
template <int kCap>
__device__ __forceinline__ void pfx_flush_slots(const PfArgs& a, int lane, PfxEoPend<kCap>& pend) {
#pragma unroll
    for (int k = 0; k < PfxEoPend<kCap>::kSlices; k++) {
        const uint32_t i = uint32_t(k) * 64 + uint32_t(lane);
        if (i < pend.n && pend.base + i < a.ev_cap) a.eo_slot[pend.base + i] = pend.r[k];
    }
    pend.n = 0;
}
template <int kCap = kEvBuf>
__device__ __forceinline__ void pfx_flush_events(const PfArgs& a, int lane, PfxEv* ebuf, uint32_t* ecnt, uint32_t at_least, PfxEoPend<kCap>& pend) {
    pf_fence();
    uint32_t n = uint32_t(__builtin_amdgcn_readfirstlane(int(*ecnt)));
    if (n < at_least) return;
    if (n > uint32_t(kCap)) n = kCap;
    if (a.eo_bb && pend.n) pfx_flush_slots<kCap>(a, lane, pend);
    constexpr int kSlices = (kCap + 63) / 64;
    uint64_t key[kSlices];
    uint32_t node[kSlices], cnt[kSlices];
    uint32_t recs = 0;
#pragma unroll
    for (int k = 0; k < kSlices; k++) {
        const uint32_t i = uint32_t(k) * 64 + uint32_t(lane);
        key[k] = 0; node[k] = 0;
        if (i < n) { key[k] = ebuf[i].key(); node[k] = ebuf[i].node; }
    }
#pragma unroll
    for (int k = 0; k < kSlices; k++) {   // the own counts of all buffered events in one round of gathers
        cnt[k] = uint32_t(k) * 64 + uint32_t(lane) < n ? a.own_cnt[node[k]] : 0u;
        recs += cnt[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) recs += __shfl_xor(recs, o, 64);
    unsigned long long base = 0;
    if (lane == 0) {
        base = atomicAdd(&a.ev_ctr[0], static_cast<unsigned long long>(n));
        atomicAdd(&a.ev_ctr[1], static_cast<unsigned long long>(recs));
        *ecnt = 0;
    }
    base = (static_cast<unsigned long long>(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(base >> 32))))) << 32) |
           uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(base))));
#pragma unroll
    for (int k = 0; k < kSlices; k++) {
        const uint32_t i = uint32_t(k) * 64 + uint32_t(lane);
        if (i < n && base + i < a.ev_cap) {
            PfEvent* dst = a.events + (base + i);
            dst->key = key[k]; dst->node = node[k]; dst->cnt = cnt[k];
            if (a.eo_bb) {
                const uint64_t b = ((key[k] >> 16) - 1 - a.eo_origin) >> a.eo_shift;
                pend.r[k] = uint32_t(atomicAdd(&a.eo_bb[b], (1ull << 32) | cnt[k]) >> 32);
            }
        }
    }
    if (a.eo_bb) { pend.base = base; pend.n = n; }
    pf_fence();
}

// Level-3 hand-off to a second pass (launch_pfx_count with a PfxHitList): each verifier wavefront owns one segment of the
// global hit list and appends its level-2 hits there, 64 at a time (its stores cost the producers nothing); k_pfx_verify
// then walks all hits of all segments with every CU and full occupancy.  hits == nullptr: level 3 inline (dense batches).
struct PfxHits {
    uint64_t* hits;       // [n_seg][seg_cap]: {low 32 bits of v - row0, node | high bits} (the hit-queue encoding)
    uint32_t* seg_n;      // [n_seg] hits appended per segment
    uint32_t seg_cap;     // entries per segment = segment stride
};

// kLong: level 2 compares a.xdepth = 5..8 prefix bytes (HotTables::pfx_map8) instead of four.
// kGate (4-byte level 2 only): the verifiers first test the survivors against the L2-resident exact-prefix BIT table
// (HotTables::pf_bits3, 64 bits per pattern: one 4-byte gather that hits L2); what passes -- true prefixes plus ~1 % of
// the rest -- is queued with its position only, and the hash-map lookup (key read back from the haystack) and the trie walk
// run over dense batches of 64 (or in the second pass when passes are at least 1/8 of the survivors).  Without the gate
// every survivor (3 % of the positions of random text at 100 000 patterns) costs a 16-byte gather from the 8 MB map, which
// misses L2: profiles/r03_pfx_pmc.json, 6 TB/s of fabric reads for a 1.8 TB/s scan.  Measured on config 4, 8 GiB
// (gpurun_out r03a/r03b): no gate 4.72 ms; gate with every pass handed to the second pass 4.23 + 1.21 ms (k_pfx_verify:
// three dependent gathers per entry); gate with inline batches 4.36 ms.
// kKey8 (long-prefix level 2 only, a.xdepth == 8): level 1 tests the whole 8-byte prefix (a.bits = HotTables::pfx_bits8).
// kShort (with kX2): short mode -- the producers also compare one or two stragglers of 3..8 bytes at every position (PfArgs::short_*).
template <bool kLong, int kXProducers, int kXVerifiers, bool kGate = false, bool kKey8 = false, bool kX2 = false, bool kShort = false>
__global__ __launch_bounds__(kPfBlock) void k_pfx_count(PfArgs a, ScanGeom g, uint32_t* __restrict__ counts, PfxHits hl) {
    static_assert(kX2 || !kShort, "short mode: the every-other-position level 1 only");
    static_assert(kXProducers + kXVerifiers <= kPfWaves && kXProducers % kXVerifiers == 0, "wave roles");
    static_assert(!(kLong && kGate), "the gate fronts the 4-byte map");
    static_assert(kLong || !kKey8, "the 8-byte level 1 goes with the long-prefix level 2");
    static_assert(kKey8 || !kX2, "every other position: the 8-byte level 1 only");
    if (a.gate && *a.gate != a.gate_val) return;   // (the probe chose the other filter)
    constexpr int kXPerVerifier = kXProducers / kXVerifiers;   // producers served by one verifier wave
    // survivors per verifier lane per round: four; two under the 8-byte level 1, whose rings hold 128 (and whose level 3 wants the registers)
    constexpr int kRB = kKey8 ? 2 : kXBatch;
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[kPfxBitsBytes / 4];
    constexpr int kQ = kXQueue;   // entries per ring: 8 bytes each, 4 under the 8-byte level 1 (the position alone)
    __shared__ __attribute__((aligned(16))) uint64_t s_ring[kXProducers][kKey8 ? kQ / 2 : kQ];
    // per-verifier event buffer: large where LDS has room (the 8-byte level 1 with 4 verifiers or fewer: its rings are half the size)
    constexpr int kEvX = (kKey8 && kXVerifiers <= 2) ? 256 : kKey8 ? 128 : kEvBuf, kEvXFlush = kEvX == kEvBuf ? kEvFlush : kEvX - 64;
    __shared__ PfxEv s_ev[kXVerifiers][kEvX];
    __shared__ uint8_t s_acls[256];
    // hits per level-3 batch: one per lane; two under the 8-byte level 1 (three and four were measured: nothing / slower)
    constexpr uint32_t kDrain = kKey8 ? 128 : 64;
    constexpr int kWalks = int(kDrain / 64);
    __shared__ uint64_t s_hitq[kXVerifiers][kDrain + 64];   // level-2 hits awaiting level 3 (a round adds at most 64 per slot, drained in between)
    // 8-byte level 1: hits whose prefix node has a chain-tail record (hot.hpp) wait in s_hitq -- a batch of them is one
    // round of independent gathers --, the others (the trie branches below the prefix, a pattern ends inside the chain) in
    // s_slowq for a walk: ONE walking lane makes its whole batch wait for its trie levels, so the walks get batches of their own
    constexpr uint32_t kSlowDrain = 64;
    __shared__ uint64_t s_slowq[kKey8 ? kXVerifiers : 1][kKey8 ? 2 * kSlowDrain : 1];
    __shared__ uint32_t s_tail[kXProducers], s_head[kXProducers], s_done[kXProducers], s_task[kXProducers], s_ecnt[kXVerifiers];
    if (threadIdx.x < kXProducers) { s_tail[threadIdx.x] = 0; s_head[threadIdx.x] = 0; s_done[threadIdx.x] = 0; s_task[threadIdx.x] = 0; }
    if (threadIdx.x < kXVerifiers) s_ecnt[threadIdx.x] = 0;
    if (threadIdx.x < 256) s_acls[threadIdx.x] = a.acls[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < kPfxBitsBytes / 4; i += kPfBlock) s_bits[i] = a.bits[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t task_bytes = uint64_t(kTaskRows) * kRowBytes;
    if (wave >= kXProducers + kXVerifiers) return;   // (role experiments with fewer than 16 active wavefronts)
    if (wave < kXProducers) {
        // ---------------------------------------------------------------- producer
        PfxProducer<kQ> st{a, g, s_bits, s_ring[wave], &s_tail[wave], &s_head[wave], &s_task[wave]};
        [[maybe_unused]] const unsigned long long prof_t0 = PFX_CLOCK();
        st.lane = lane;
        const uint64_t prod_id = uint64_t(blockIdx.x) * kXProducers + wave;
        const uint64_t n_prod = uint64_t(gridDim.x) * kXProducers;
        auto is_interior = [&](uint64_t tb) {
            return tb >= a.scan_lo && tb + task_bytes + 16 <= a.hull_end && tb + task_bytes <= g.emit_hi;
        };
        for (uint64_t task = prod_id; task < a.n_tasks; task += n_prod) {
            const uint64_t tb = a.row0 + task * task_bytes;
            const uint64_t next_base = a.row0 + (task + n_prod) * task_bytes;
            const bool next_interior = task + n_prod < a.n_tasks && is_interior(next_base);
            if (kX2 && task == 0 && a.row0 >= a.scan_lo) st.push_first(kShort);
            if (is_interior(tb)) st.template run_task<false, kKey8, kX2, kShort>(tb, next_base, next_interior);
            else st.template run_task<true, kKey8, kX2, kShort>(tb, next_base, next_interior);
        }
        pf_fence();
        if (lane == 0) lds_poke(&s_done[wave], 1u);   // (LDS executes a wavefront's operations in order: after its last tail)
        if (lane == 0) { PFX_PROF_ADD(0, PFX_CLOCK() - prof_t0); PFX_PROF_ADD(1, st.prof_stall); PFX_PROF_ADD(2, st.prof_stalls); }
        return;
    }
    // -------------------------------------------------------------------- verifier
    const int vw = wave - kXProducers;
    PfxEv* ebuf = s_ev[vw];
    uint32_t* ecnt = &s_ecnt[vw];
    // (the consumed counts live in s_head -- this wavefront is their only writer -- so the loop over its producers stays
    // rolled: unrolled, the level-2 / hand-off code was there once per producer, 35 000 instructions with three of them)
    uint64_t* hitq = s_hitq[vw];
    uint32_t hit_n = 0;   // wave-uniform
    const uint32_t seg = blockIdx.x * kXVerifiers + uint32_t(vw);
    uint32_t seg_fill = 0;   // wave-uniform: hits this wavefront has appended to its segment of the global list
    // wave-uniform.  Level 3 is handed to the second pass while level-2 hits are at least 1/8 of the survivors lately
    // (cand_acc / hit_acc: decayed counts): then it is level 3 that holds the verifier up (natural text against a
    // dictionary, 4-byte level 2).  Random text against 100 000 patterns also keeps the verifiers busy, but with level 2
    // (3 % hits): their few walks hide behind it, while a second pass pays for the same random HBM gathers on its own
    // (+1 ms on 5 ms).  (Requiring a nearly full ring as well changed nothing measurable.)
    uint32_t cand_acc = 0, hit_acc = 0;
    [[maybe_unused]] unsigned long long prof_v0 = PFX_CLOCK(), prof_idle = 0, prof_l2 = 0, prof_l3 = 0, prof_rounds = 0, prof_surv = 0, prof_batches = 0, prof_hits = 0, prof_steps = 0, prof_flush = 0, prof_slow = 0, prof_slow_batches = 0, prof_slow_hits = 0;
    uint64_t* slowq = s_slowq[kKey8 ? vw : 0];
    uint32_t slow_n = 0;   // wave-uniform
    PfxEoPend<kEvX> eo_pend;
    auto flush_if = [&](bool buffered) {
        const unsigned long long f0 = PFX_CLOCK();
        if (a.events && __builtin_amdgcn_ballot_w64(buffered) != 0) pfx_flush_events<kEvX>(a, lane, ebuf, ecnt, kEvXFlush, eo_pend);
        prof_flush += PFX_CLOCK() - f0;
    };
    auto drain_hits = [&](uint32_t n) {   // level 3 for the LAST n queued hits (order is irrelevant)
        const unsigned long long d0 = PFX_CLOCK(); prof_batches++; prof_hits += n;
        pf_fence();
        hit_n -= n;
        uint64_t e[kWalks];
#pragma unroll
        for (int w = 0; w < kWalks; w++) e[w] = uint32_t(lane) + 64u * w < n ? hitq[hit_n + 64 * w + lane] : 0;
        pf_fence();
        if (hl.hits && hit_acc * 8 >= cand_acc && seg_fill + n <= hl.seg_cap) {   // second pass will walk them (k_pfx_verify)
#pragma unroll
            for (int w = 0; w < kWalks; w++)
                if (uint32_t(lane) + 64u * w < n && !(PFX_EXP & 4)) hl.hits[uint64_t(seg) * hl.seg_cap + seg_fill + 64 * w + lane] = e[w];
            seg_fill += n;
            return;
        }
        bool buffered = false;
        if constexpr (kKey8) {
            // tail compares (inline level 3) or, with the tails switched off / in two passes' overflow, walks
            uint64_t v[kWalks];
            uint32_t node[kWalks], tail1[kWalks];
            uint32_t any_node = 0, any_tail = 0;
#pragma unroll
            for (int w = 0; w < kWalks; w++) {
                v[w] = 0; node[w] = 0; tail1[w] = 0;
                if (uint32_t(lane) + 64u * w < n) pfx_hit_decode(a, e[w], v[w], node[w], tail1[w]);
                any_node |= node[w]; any_tail |= tail1[w];
            }
            if (any_tail) buffered |= pfx_verify_tails<kWalks, kEvX>(a, g, counts, v, tail1, ebuf, ecnt, s_acls);
            if (any_node) buffered |= pfx_verify_n_from<kWalks, kEvX>(a, g, counts, v, node, ebuf, ecnt, s_acls, &prof_steps);
        } else if (uint32_t(lane) < n) {
            uint64_t v;
            uint32_t node, tail1;
            pfx_hit_decode(a, e[0], v, node, tail1);
            if (kGate) node = pfx_resolve(a, g, v);   // (its segment of the hit list is full: level 2 and 3 here)
            if (tail1) buffered = pfx_verify_tail<kEvX>(a, g, counts, v, tail1, ebuf, ecnt, s_acls);
            else if (node) buffered = pfx_verify_from<false, kEvX>(a, g, counts, v, node, ebuf, ecnt, s_acls);
        }
        flush_if(buffered);
        prof_l3 += PFX_CLOCK() - d0;
    };
    auto drain_slow = [&](uint32_t n) {   // the walks: the LAST n <= 64 entries of the slow queue, one per lane
        const unsigned long long d0 = PFX_CLOCK(); prof_slow_batches++; prof_slow_hits += n;
        pf_fence();
        slow_n -= n;
        const uint64_t e = uint32_t(lane) < n ? slowq[slow_n + lane] : 0;
        pf_fence();
        uint64_t v[1] = {0};
        uint32_t node[1] = {0u}, tail1 = 0;
        if (uint32_t(lane) < n) pfx_hit_decode(a, e, v[0], node[0], tail1);
        bool buffered = false;
        if (tail1) buffered = pfx_verify_tail<kEvX, true>(a, g, counts, v[0], tail1, ebuf, ecnt, s_acls);   // a node with several tail records
        if (__any(node[0] != 0)) buffered |= pfx_verify_n_from<1, kEvX>(a, g, counts, v, node, ebuf, ecnt, s_acls, &prof_steps);
        flush_if(buffered);
        prof_slow += PFX_CLOCK() - d0;
    };
    // levels 2 and 3 for the survivors of one round: ent = ring entries (the 4-byte window in the low word), rel = v - row0
    const uint64_t n_prod = uint64_t(gridDim.x) * kXProducers;
    auto rel_of = [&](uint64_t e, uint64_t prod_id, uint32_t seq_cur) {   // v - row0 of a ring entry (< 2^43)
        const uint32_t pos = uint32_t(e >> 32);
        const uint32_t seq = kShort ? seq_cur - ((seq_cur - ((pos >> 16) & 0x7FFFu)) & 0x7FFFu)   // (bit 31: a straggler's start)
                                    : seq_cur - ((seq_cur - (pos >> 16)) & 0xFFFFu);
        return (prod_id + uint64_t(seq) * n_prod) * task_bytes + (pos & 0xFFFFu);
    };
    auto process = [&](const uint64_t (&ent)[kRB], bool (&go)[kRB], auto&& rel_fn) {   // rel_fn(b) = v - row0 of slot b
        if (PFX_EXP & 1) return;
        [[maybe_unused]] bool short_buffered = false;
        uint64_t rel[kRB];   // (the 4-byte level 2 needs it for its hits only: computed on demand there)
        if constexpr (kLong) {
#pragma unroll
            for (int b = 0; b < kRB; b++) rel[b] = rel_fn(b);
        }
        // level 2: the exact first four bytes -> trie node at depth 4 (one 16-byte gather per survivor from the
        // L2-resident hash map, all of a round in flight together; the key came with the ring entry)
        uint4 q[kRB];
        uint32_t bk[kRB];
        uint32_t node[kRB];
        bool more[kRB];
        bool any_more = false;
        if constexpr (kLong) {
            // the survivor's bytes 4..depth-1 from the haystack (one 8-byte gather; the line was streamed by the
            // producer a moment ago), then ONE exact lookup of the whole prefix.  A start closer than `depth` bytes to
            // the end of the span cannot begin a pattern (depth <= shortest pattern).
            uint32_t khi[kRB], klo[kRB];
            const uint32_t himask = a.xdepth >= 8 ? 0xFFFFFFFFu : (1u << (8 * (a.xdepth - 4))) - 1u;
#pragma unroll
            for (int b = 0; b < kRB; b++) {
                const uint64_t v = a.row0 + rel[b];
                // short mode: an entry with bit 31 names a position where a straggler's first bytes stand
                const bool is_short = kShort && go[b] && (uint32_t(ent[b] >> 32) >> 31) != 0;
                go[b] = go[b] && (is_short ? v + 3 <= g.emit_hi : v + a.xdepth <= g.emit_hi);
                uint32_t w[2] = {0u, 0u};
                if (go[b]) {   // (carrying bytes 4..7 in the ring entry instead was measured in round 4: no gain, 6 KiB of LDS; the whole key in round 6: slower)
                    ACGPU_HAY_CHECK(g, v, v + 8 <= g.emit_hi ? 8 : g.emit_hi - v);
                    if (v + 8 <= g.emit_hi) __builtin_memcpy(w, g.hay16 + v, 8);
                    else if (kShort && is_short) {   // (a straggler may end with the span: byte by byte)
                        for (uint32_t i = 0; v + i < g.emit_hi; i++) w[i >> 2] |= uint32_t(g.hay16[v + i]) << (8 * (i & 3));
                    } else {   // the last bytes of the span (a prefix of 5..7 bytes still fits): bytes 0..3, then one by one
                        __builtin_memcpy(&w[0], g.hay16 + v, 4);
                        for (uint32_t i = 4; v + i < g.emit_hi; i++) w[1] |= uint32_t(g.hay16[v + i]) << (8 * (i - 4));
                    }
                }
                if constexpr (kShort) {
                    if (is_short && go[b]) {   // every byte of every straggler, here; no lookup for this entry
                        for (uint32_t i = 0; i < a.short_n; i++) {
                            const uint32_t sl = a.short_len[i];
                            const uint32_t mlo = sl >= 4 ? 0xFFFFFFFFu : (1u << (8 * sl)) - 1u;
                            const uint32_t mhi = sl >= 8 ? 0xFFFFFFFFu : sl > 4 ? (1u << (8 * (sl - 4))) - 1u : 0u;
                            if (v + sl <= g.emit_hi && ((w[0] ^ a.short_lo[i]) & mlo) == 0 && ((w[1] ^ a.short_hi[i]) & mhi) == 0)
                                short_buffered |= pfx_record<kEvX>(a, g, counts, v, v + sl - 1, a.short_node[i], a.own_cnt[a.short_node[i]], ebuf, ecnt);
                        }
                    }
                    if (is_short) go[b] = false;
                }
                khi[b] = w[1] & himask;
                klo[b] = kKey8 ? w[0] : uint32_t(ent[b]);   // (8-byte level 1: the ring entry carries no window)
                if (PFX_EXP & 16) go[b] = go[b] && w[0] == 0x12345678u && w[1] == 0x9ABCDEF0u;   // (timing: the key is read, nothing is looked up)
            }
#pragma unroll
            for (int b = 0; b < kRB; b++) {
                bk[b] = pfx_map8_bucket(klo[b], khi[b], a.xmap_log2);
                q[b] = go[b] ? a.xmap[bk[b]] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < kRB; b++) {
                const uint32_t val = q[b].z & ~kPfxMapOverflow;
                node[b] = (val && q[b].x == klo[b] && q[b].y == khi[b]) ? val : 0u;
                more[b] = go[b] && !node[b] && (q[b].z & kPfxMapOverflow);
                any_more |= more[b];
            }
            while (__any(any_more)) {   // rare: the next buckets of all slots together
                any_more = false;
#pragma unroll
                for (int b = 0; b < kRB; b++) {
                    bk[b] = (bk[b] + 1) & ((1u << a.xmap_log2) - 1);
                    if (more[b]) q[b] = a.xmap[bk[b]];
                }
#pragma unroll
                for (int b = 0; b < kRB; b++) {
                    if (!more[b]) continue;
                    const uint32_t val = q[b].z & ~kPfxMapOverflow;
                    node[b] = (val && q[b].x == klo[b] && q[b].y == khi[b]) ? val : 0u;
                    more[b] = !node[b] && (q[b].z & kPfxMapOverflow);
                    any_more |= more[b];
                }
            }
        } else if constexpr (kGate) {
            // node[b] = 1: the exact-prefix bit table has the window (the map lookup is the second pass's)
            uint32_t bw[kRB];
#pragma unroll
            for (int b = 0; b < kRB; b++) {
                bk[b] = pf_hash3(uint32_t(ent[b]), a.bits3_log2);
                bw[b] = go[b] ? a.bits3[bk[b] >> 5] : 0u;
            }
#pragma unroll
            for (int b = 0; b < kRB; b++) node[b] = (bw[b] >> (bk[b] & 31u)) & 1u;
        } else {
#pragma unroll
        for (int b = 0; b < kRB; b++) {
            bk[b] = pfx_map_bucket(uint32_t(ent[b]), a.xmap_log2);
            q[b] = go[b] ? a.xmap[bk[b]] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < kRB; b++) {
            const uint32_t key = uint32_t(ent[b]);
            node[b] = 0;
            if (q[b].y && q[b].x == key) node[b] = q[b].y & ~kPfxMapOverflow;
            else if (q[b].w && q[b].z == key) node[b] = q[b].w;
            more[b] = go[b] && !node[b] && (q[b].y & kPfxMapOverflow);
            any_more |= more[b];
        }
        while (__any(any_more)) {   // rare (0.1 % of the buckets overflow): the next buckets of all slots together
            any_more = false;
#pragma unroll
            for (int b = 0; b < kRB; b++) {
                bk[b] = (bk[b] + 1) & ((1u << a.xmap_log2) - 1);
                if (more[b]) q[b] = a.xmap[bk[b]];
            }
#pragma unroll
            for (int b = 0; b < kRB; b++) {
                if (!more[b]) continue;
                const uint32_t key = uint32_t(ent[b]);
                if (q[b].y && q[b].x == key) node[b] = q[b].y & ~kPfxMapOverflow;
                else if (q[b].w && q[b].z == key) node[b] = q[b].w;
                more[b] = !node[b] && (q[b].y & kPfxMapOverflow);
                any_more |= more[b];
            }
        }
        }
        // level 2 hits (~3 % of the survivors: true 4-byte prefix matches) go to this wavefront's hit queue; level 3
        // runs over DENSE batches of 64 -- its dependent HBM gathers (haystack byte -> trie row) cost microseconds
        // whatever the number of active lanes, and verifying the handful of hits of every round on the spot made
        // the round four times longer
#pragma unroll
        for (int b = 0; b < kRB; b++) {
            const bool hit = node[b] != 0 && !(PFX_EXP & 2);
            const unsigned long long m = __ballot(hit);
            if (m == 0) continue;
            const uint64_t relb = kLong ? rel[b] : rel_fn(b);   // v - row0 < 2^43
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
            uint32_t tail1 = 0;   // chain-tail record of the prefix node (index + 1), if it has one
            if constexpr (kLong) tail1 = a.tails ? q[b].w : 0u;
            const bool multi = (tail1 & kPfxTailMulti) != 0;   // several records: compared in the batches of the walks
            tail1 &= ~kPfxTailMulti;
            const uint64_t entry = tail1 ? pfx_hit_entry(relb, tail1 - 1, false, true)
                                         : pfx_hit_entry(relb, kGate ? 0u : node[b], !kGate && (node[b] >> 31), false);
            const uint32_t nh = uint32_t(__popcll(m));
            hit_acc += nh;
            if (hl.hits && hit_acc * 8 >= cand_acc && seg_fill + nh <= hl.seg_cap) {
                // handed to the second pass straight from the registers: the stores of a whole round retire together
                // with its level-2 gathers (through the hit queue every 64 hits waited for their own store: +1 ms per GiB)
                if (hit && !(PFX_EXP & 4)) hl.hits[uint64_t(seg) * hl.seg_cap + seg_fill + rank] = entry;
                seg_fill += nh;
                continue;
            }
            if constexpr (kKey8) {
                // two queues: tail compares and walks are verified in batches of their own (see s_slowq)
                const bool slow = hit && (tail1 == 0 || multi);
                const unsigned long long ms = __ballot(slow), mf = m & ~ms;
                if (hit && !slow) hitq[hit_n + __builtin_amdgcn_mbcnt_hi(uint32_t(mf >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mf), 0u))] = entry;
                if (slow) slowq[slow_n + __builtin_amdgcn_mbcnt_hi(uint32_t(ms >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(ms), 0u))] = entry;
                hit_n += uint32_t(__popcll(mf));
                slow_n += uint32_t(__popcll(ms));
                // (level 3 runs between rounds, with nothing of a round live in registers; a round takes no more survivors
                // than both queues have room for)
            } else {
                if (hit) hitq[hit_n + rank] = entry;
                hit_n += nh;
                if (hit_n >= kDrain) drain_hits(kDrain);
            }
        }
        if constexpr (kShort) flush_if(short_buffered);
    };
    for (;;) {
        bool all_done = true, any_work = false;
        cand_acc -= cand_acc >> 2; hit_acc -= hit_acc >> 2;
#pragma unroll 1
        for (int k = 0; k < kXPerVerifier; k++) {
            const int pw = vw * kXPerVerifier + k;
            const uint32_t head_k = lds_peek(&s_head[pw]);
            // read `done` BEFORE `tail`: a producer publishes its last entries before it raises done
            const uint32_t done = lds_peek(&s_done[pw]);
            pf_fence();
            const uint32_t tail = lds_peek(&s_tail[pw]);
            uint32_t avail = tail - head_k;
            if (!done) all_done = false;
            if (avail == 0) continue;
            if (avail < uint32_t(64) && !done) continue;   // let batches fill up (a finished producer's rest is taken as is; from 96 on: slower)
            any_work = true;
            if (avail > uint32_t(64 * kRB)) avail = 64 * kRB;
            if constexpr (kKey8) {
                if (hit_n >= kDrain) drain_hits(kDrain);
                if (slow_n >= kSlowDrain) drain_slow(kSlowDrain);
                avail = std::min<uint32_t>({avail, kDrain + 64u - hit_n, 2 * kSlowDrain - slow_n});   // >= 64: a finished producer's rest, or 65
            }
            cand_acc += avail;
            uint64_t ent[kRB];
            bool go[kRB];
#pragma unroll
            for (int b = 0; b < kRB; b++) {
                const uint32_t e = uint32_t(b) * 64 + uint32_t(lane);
                go[b] = e < avail;
                if (kKey8) ent[b] = go[b] ? uint64_t(*(volatile lds_u32*)(reinterpret_cast<uint32_t*>(s_ring[pw]) + ((head_k + e) & uint32_t(kQ - 1)))) << 32 : 0;
                else ent[b] = go[b] ? *(volatile lds_u64*)(&s_ring[pw][(head_k + e) & uint32_t(kQ - 1)]) : 0;   // (written by another wavefront)
            }
            pf_fence();
            const uint32_t seq_cur = lds_peek(&s_task[pw]);   // >= the sequence number of every entry read above
            if (lane == 0) lds_poke(&s_head[pw], head_k + avail);   // the producer may reuse the slots
            const uint64_t prod_id = uint64_t(blockIdx.x) * kXProducers + uint32_t(pw);
            const unsigned long long r0 = PFX_CLOCK(), l3_before = prof_l3; prof_rounds++; prof_surv += avail;
            process(ent, go, [&](int b) { return rel_of(ent[b], prod_id, seq_cur); });
            prof_l2 += (PFX_CLOCK() - r0) - (prof_l3 - l3_before);
        }
        if (!any_work && hit_n) drain_hits(hit_n < kDrain ? hit_n : kDrain);   // idle: verify what is queued
        if (!any_work && slow_n) drain_slow(slow_n < kSlowDrain ? slow_n : kSlowDrain);
        if (all_done && !any_work) {
            // every producer raised done before its tail was read above: nothing can arrive any more
            bool empty = true;
#pragma unroll
            for (int k = 0; k < kXPerVerifier; k++)
                empty = empty && lds_peek(&s_tail[vw * kXPerVerifier + k]) == lds_peek(&s_head[vw * kXPerVerifier + k]);
            if (empty) break;
        }
        if (!any_work) { const unsigned long long i0 = PFX_CLOCK(); __builtin_amdgcn_s_sleep(8); prof_idle += PFX_CLOCK() - i0; }
    }
    while (hit_n) drain_hits(hit_n < kDrain ? hit_n : kDrain);
    while (slow_n) drain_slow(slow_n < kSlowDrain ? slow_n : kSlowDrain);
    if (hl.hits && lane == 0) hl.seg_n[seg] = (PFX_EXP & 4) ? 0u : seg_fill;
    if (a.events) {
        pfx_flush_events<kEvX>(a, lane, ebuf, ecnt, 1, eo_pend);
        if (a.eo_bb && eo_pend.n) pfx_flush_slots<kEvX>(a, lane, eo_pend);
    }
#ifdef PFX_PROF
    {   // walk-loop trips of the wavefront = the longest-living lane's count, batch by batch; summed per lane here, so take the maximum
        unsigned long long m = prof_steps;
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
        if (lane == 0) PFX_PROF_ADD(12, m);
    }
#endif
    if (lane == 0) {
        PFX_PROF_ADD(3, PFX_CLOCK() - prof_v0); PFX_PROF_ADD(4, prof_idle); PFX_PROF_ADD(5, prof_l2); PFX_PROF_ADD(6, prof_l3);
        PFX_PROF_ADD(11, prof_flush); PFX_PROF_ADD(13, prof_slow); PFX_PROF_ADD(14, prof_slow_batches); PFX_PROF_ADD(15, prof_slow_hits);
        PFX_PROF_ADD(7, prof_rounds); PFX_PROF_ADD(8, prof_surv); PFX_PROF_ADD(9, prof_batches); PFX_PROF_ADD(10, prof_hits);
    }
}

// ---- second pass: level 3 over the global hit list.  One hit per lane, every CU, 32 wavefronts per CU: the dependent
// gathers of the trie walk (haystack byte -> trie row, microseconds each) are hidden by occupancy instead of stalling four
// verifier wavefronts per CU.  seg_off = exclusive prefix of seg_n (k_pfx_scan_segments); a flat hit index is mapped to
// its segment by binary search in LDS.
constexpr int kVfBlock = 256;
constexpr int kVfEvBuf = 256, kVfEvFlush = 192;   // per-wave event buffer of the second pass (16 KiB per workgroup)
__global__ __launch_bounds__(1024) void k_pfx_scan_segments(const uint32_t* __restrict__ seg_n, uint32_t n_seg, uint32_t* __restrict__ seg_off) {
    __shared__ uint32_t s_part[1024];
    // thread t sums a contiguous slice, then one thread scans the partials (n_seg <= a few thousand)
    const uint32_t per = (n_seg + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per, b1 = b0 + per < n_seg ? b0 + per : n_seg;
    uint32_t sum = 0;
    for (uint32_t i = b0; i < b1; i++) sum += seg_n[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 1024; t++) { const uint32_t x = s_part[t]; s_part[t] = run; run += x; }
        seg_off[n_seg] = run;
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t i = b0; i < b1; i++) { seg_off[i] = run; run += seg_n[i]; }
}

__global__ __launch_bounds__(kVfBlock) void k_pfx_verify(PfArgs a, ScanGeom g, uint32_t* __restrict__ counts, PfxHits hl,
                                                         const uint32_t* __restrict__ seg_off, uint32_t n_seg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* s_off = reinterpret_cast<uint32_t*>(smem);                         // n_seg + 1
    PfxEv* s_ev = reinterpret_cast<PfxEv*>(smem + ((size_t(n_seg) + 1) * 4 + 15 & ~size_t(15)));
    uint32_t* s_ecnt = reinterpret_cast<uint32_t*>(s_ev + (kVfBlock / 64) * kVfEvBuf);
    uint8_t* s_acls = reinterpret_cast<uint8_t*>(s_ecnt + kVfBlock / 64);
    for (uint32_t i = threadIdx.x; i <= n_seg; i += kVfBlock) s_off[i] = seg_off[i];
    if (threadIdx.x < kVfBlock / 64) s_ecnt[threadIdx.x] = 0;
    s_acls[threadIdx.x] = a.acls[threadIdx.x];   // (kVfBlock == 256)
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PfxEv* ebuf = s_ev + wave * kVfEvBuf;
    uint32_t* ecnt = s_ecnt + wave;
    const uint32_t total = s_off[n_seg];
    PfxEoPend<kVfEvBuf> eo_pend;
    // whole wavefronts iterate together (the event flush is wave-collective): round the trip count up per wave
    for (uint64_t base = (uint64_t(blockIdx.x) * kVfBlock + uint64_t(wave) * 64); base < total; base += uint64_t(gridDim.x) * kVfBlock) {
        const uint64_t i = base + uint32_t(lane);
        bool buffered = false;
        if (i < total) {
            uint32_t lo = 0, hi = n_seg;   // largest seg with s_off[seg] <= i
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_off[mid] <= uint32_t(i)) lo = mid; else hi = mid; }
            const uint64_t e = hl.hits[uint64_t(lo) * hl.seg_cap + (uint32_t(i) - s_off[lo])];
            uint64_t v;
            uint32_t node, tail1;
            pfx_hit_decode(a, e, v, node, tail1);
            if (tail1) buffered = pfx_verify_tail<kVfEvBuf, true>(a, g, counts, v, tail1, ebuf, ecnt, s_acls);
            else {
                if (node == 0) node = pfx_resolve(a, g, v);   // handed over by the bit-table gate: level 2 is still to do
                if (node) buffered = pfx_verify_from<true, kVfEvBuf>(a, g, counts, v, node, ebuf, ecnt, s_acls);
            }
        }
        if (a.events && __builtin_amdgcn_ballot_w64(buffered) != 0) pfx_flush_events<kVfEvBuf>(a, lane, ebuf, ecnt, kVfEvFlush, eo_pend);
    }
    if (a.events) {
        pfx_flush_events<kVfEvBuf>(a, lane, ebuf, ecnt, 1, eo_pend);
        if (a.eo_bb && eo_pend.n) pfx_flush_slots<kVfEvBuf>(a, lane, eo_pend);
    }
}

}  // namespace

#ifdef PFX_PROF
extern "C" int acgpu_debug_pfx_prof(unsigned long long* out16, int reset) {   // lib/exp builds only
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pfx_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pfx_prof), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

bool pf_uses_large_set(const HotTables& h, const PfRoute& route) {
    const uint32_t min_patterns = h.var.pfx_min_patterns >= 0 ? uint32_t(h.var.pfx_min_patterns) : kPfxMinPatterns;   // (variant)
    return h.pfx_ready && (h.n_patterns >= min_patterns || route.force_pfx);
}

hipError_t launch_pf_any(const HotTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s, void* events,
                         unsigned long long* ev_ctr, uint64_t ev_cap, PfRoute route) {
    // large pattern sets: the 4-byte-key filter with verifier wavefronts; else the two-type 3-byte-key filter
    if (pf_uses_large_set(h, route))
        return launch_pfx_count(h, g, counts, s, events, ev_ctr, ev_cap, route.hit_work, route.hit_work_bytes, route.gate, route.gate_val, route.hist);
    return launch_pf_count(h, g, counts, s, events, ev_ctr, ev_cap, route);
}

size_t pfx_hit_work_bytes(uint64_t span_bytes) {
    // hit list: one 8-byte entry per 16 haystack bytes (a segment that fills up verifies inline), per-segment counts and
    // their prefix (at most 4 096 segments: 1 024 CUs x 4 verifier wavefronts)
    const uint64_t entries = std::min<uint64_t>(uint64_t(1) << 28, std::max<uint64_t>(uint64_t(1) << 19, (span_bytes / 16 + 63) & ~uint64_t(63)));
    return size_t(entries * 8 + 2 * 4100 * 4 + 256);
}

hipError_t launch_pfx_count(const HotTables& h, const ScanGeom& g, uint32_t* counts, hipStream_t s, void* events,
                            unsigned long long* ev_ctr, uint64_t ev_cap, void* hit_work, size_t hit_work_bytes,
                            const uint32_t* gate, uint32_t gate_val, PfEoHist hist) {
    PfArgs a{};
    a.gate = gate; a.gate_val = gate_val;
    if (events) { a.eo_bb = hist.bb; a.eo_slot = hist.slot; a.eo_origin = hist.origin; a.eo_shift = hist.shift; }
    a.events = static_cast<PfEvent*>(events); a.ev_ctr = ev_ctr; a.ev_cap = ev_cap;
    a.bits = h.pfx_bits;   // (the 8-byte-key table below when that level 1 runs)
    a.bits2 = nullptr; a.atab = h.atab; a.acls = h.acls; a.ashift = h.ashift; a.own_cnt = h.own_cnt;
    const bool long_key = h.pfx_map8 != nullptr;
    const bool gate_on = h.var.pfx_gate != 0;   // the bit-table gate (variant pfx_gate)
    const bool use_gate = !long_key && h.pf_bits3 != nullptr && gate_on;
    a.bits3 = use_gate ? h.pf_bits3 : nullptr; a.bits3_log2 = use_gate ? h.pf_bits3_log2 : 0;
    a.xmap = long_key ? h.pfx_map8 : h.pfx_map; a.xmap_log2 = long_key ? h.pfx_map8_log2 : h.pfx_map_log2;
    a.xdepth = long_key ? h.pfx_depth : 4;
    a.tails = long_key && h.var.pfx_tails ? h.pfx_tails : nullptr;   // (variant: also read when the tables are built)
    a.bits_bytes = kPfxBitsBytes; a.root = h.start;
    const uint64_t lo = g.emit_lo >= g.halo ? g.emit_lo - g.halo : 0;
    a.scan_lo = lo > g.cold_floor ? lo : g.cold_floor;
    a.row0 = a.scan_lo & ~uint64_t(15);
    a.hull_end = (g.emit_hi + 15) & ~uint64_t(15);
    const uint64_t task_bytes = uint64_t(kTaskRows) * kRowBytes;
    a.n_tasks = g.emit_hi > a.row0 ? (g.emit_hi - a.row0 + task_bytes - 1) / task_bytes : 0;
    hipError_t e = hipSuccess;
    if (!events) e = hipMemsetAsync(counts, 0, g.n_chunks * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (a.n_tasks == 0) return hipSuccess;
    uint64_t blocks = uint64_t(device_cus());
    // the long-key level 1 (variant pfx_key8 = 0 switches it off); its wave roles: variant pfx_key8_roles = producers of
    // 12 | 14 (the verifiers see 0.4 % of the positions of English text instead of 7 %, so nearly every wavefront can stream)
    const bool key8 = long_key && h.pfx_bits8 != nullptr && h.pfx_depth >= 5 && (h.var.pfx_key8 != 0 || h.pfx_short_n > 0);
    int roles = h.var.pfx_key8_roles;   // (per GiB of prose, with the two hit queues: 10 + 5 0.645 ms, 12 + 4 0.575, 14 + 2 0.70)
    if (roles != 14) roles = 12;   // (8 + 8 was measured -- 0.91 ms -- and no longer fits LDS beside the two hit queues)
    // short mode (the long-key tables hold the long patterns only): the one kernel that compares the stragglers, whatever the variants say
    const bool short_mode = h.pfx_short_n > 0 && long_key && h.pfx_bits8x2 != nullptr;
    if (short_mode) roles = 12;
    a.short_n = short_mode ? h.pfx_short_n : 0;
    for (uint32_t i = 0; i < kPfxShortMax; i++) {
        const uint32_t j = i < h.pfx_short_n ? i : 0;   // (an unused slot repeats the first straggler)
        a.short_lo[i] = h.pfx_short_lo[j]; a.short_hi[i] = h.pfx_short_hi[j]; a.short_len[i] = h.pfx_short_len[j]; a.short_node[i] = h.pfx_short_node[j];
    }
    const int kXProducers = key8 ? roles : long_key ? PFX_LONG_PRODUCERS : PFX_PRODUCERS;
    const int kXVerifiers = key8 ? 16 - roles : long_key ? PFX_LONG_VERIFIERS : PFX_VERIFIERS;
    const uint64_t need = (a.n_tasks + kXProducers - 1) / kXProducers;
    if (blocks > need) blocks = need;
    PfxHits hl{nullptr, nullptr, 0};
    const uint32_t n_seg = uint32_t(blocks) * kXVerifiers;
    uint32_t* seg_off = nullptr;
    // (the 8-byte level 1 verifies inline: what survives it is a true prefix nineteen times out of twenty, the verifiers
    // have little else to do, and a second pass over 4 M hits per GiB costs more than the whole first one -- measured on
    // sherlock / words-5000, 1 GiB: 0.84 ms in two passes, 0.66 ms inline)
    if (hit_work && !key8 && n_seg <= 4096 && hit_work_bytes >= size_t(2 * 4100 * 4 + 256 + 64 * 8 * n_seg)) {
        uint8_t* w = static_cast<uint8_t*>(hit_work);
        hl.seg_n = reinterpret_cast<uint32_t*>(w);
        seg_off = hl.seg_n + 4100;
        hl.hits = reinterpret_cast<uint64_t*>(w + 2 * 4100 * 4 + 256 - ((2 * 4100 * 4) % 256));
        const uint64_t entries = (hit_work_bytes - size_t(reinterpret_cast<uint8_t*>(hl.hits) - w)) / 8;
        // (segment stride = an odd number of 128-byte lines: with a power-of-two stride the verifier wavefronts, which fill
        // their segments at the same pace, would all be storing into the same memory channel)
        hl.seg_cap = uint32_t(std::min<uint64_t>(((entries / n_seg - 64) & ~uint64_t(63)) + 16, 0x7FFFFFC0u));
        if ((e = hipMemsetAsync(hl.seg_n, 0, size_t(n_seg) * 4, s)) != hipSuccess) return e;
    }
    // ... probed at every other position when every pattern has nine bytes (variant pfx_key8_x2 = 0 switches it off)
    const bool x2 = key8 && h.pfx_bits8x2 != nullptr && roles == 12 && (h.var.pfx_key8_x2 != 0 || short_mode);
    if (x2 && short_mode) {
        a.bits = h.pfx_bits8x2;
        k_pfx_count<true, 12, 4, false, true, true, true><<<dim3(uint32_t(blocks)), dim3(kPfBlock), 0, s>>>(a, g, counts, hl);
    } else if (x2) {
        a.bits = h.pfx_bits8x2;
        k_pfx_count<true, 12, 4, false, true, true><<<dim3(uint32_t(blocks)), dim3(kPfBlock), 0, s>>>(a, g, counts, hl);
    } else if (key8) {
        a.bits = h.pfx_bits8;
        const dim3 grid{uint32_t(blocks)}, block{kPfBlock};
        if (roles == 14) k_pfx_count<true, 14, 2, false, true><<<grid, block, 0, s>>>(a, g, counts, hl);
        else k_pfx_count<true, 12, 4, false, true><<<grid, block, 0, s>>>(a, g, counts, hl);   // (14 + 2 and 15 + 1 were measured and lose: profiles/r04_key8_steps.jsonl)
    } else if (long_key) k_pfx_count<true, PFX_LONG_PRODUCERS, PFX_LONG_VERIFIERS><<<dim3(uint32_t(blocks)), dim3(kPfBlock), 0, s>>>(a, g, counts, hl);
    else if (use_gate) k_pfx_count<false, PFX_PRODUCERS, PFX_VERIFIERS, true><<<dim3(uint32_t(blocks)), dim3(kPfBlock), 0, s>>>(a, g, counts, hl);
    else k_pfx_count<false, PFX_PRODUCERS, PFX_VERIFIERS><<<dim3(uint32_t(blocks)), dim3(kPfBlock), 0, s>>>(a, g, counts, hl);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (hl.hits) {
        k_pfx_scan_segments<<<dim3(1), dim3(1024), 0, s>>>(hl.seg_n, n_seg, seg_off);
        const size_t smem = ((size_t(n_seg) + 1) * 4 + 15 & ~size_t(15)) + size_t(kVfBlock / 64) * (kVfEvBuf * sizeof(PfxEv) + 4) + 256;
        k_pfx_verify<<<dim3(uint32_t(device_cus()) * 8), dim3(kVfBlock), smem, s>>>(a, g, counts, hl, seg_off, n_seg);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace acgpu
