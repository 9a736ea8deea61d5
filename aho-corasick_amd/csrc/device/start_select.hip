// Leftmost find_iter for occurrence-DENSE inputs (gfx950): the non-overlapping matches are selected from a per-START table
// instead of from the materialised occurrence stream.
//
// The reference's FindIter (src/automaton.rs:857-936) restarts a leftmost search at the end of the previous match
// (src/automaton.rs:1285-1420).  Without an empty pattern its result is a function of the occurrence set (select.hpp):
//   LeftmostFirst    next = the occurrence with the smallest start >= pos; ties: lowest pattern id
//   LeftmostLongest  next = the occurrence with the smallest start >= pos; ties: greatest length, then lowest id
// -- so only ONE occurrence per start position can ever be reported: cand(i) = the winner among the occurrences that start
// at i.  select.hip needs every occurrence as a 24-byte record first; with an occurrence per haystack byte (the
// reference's same/*, teddy1-*, jetscii benchmarks) that stream is 24x the haystack and its ordering costs more than
// the scan.  Here:
//   k_ss_block   one WAVEFRONT per block of 1 024 start positions: every position walks the TRIE from the root (the trie-only
//                transition table of the prefix filters, HotTables::atab; haystack bytes and the root row from LDS) and
//                keeps the winner along its path -- cand[i] = pattern id + 1, 4 bytes per position, nothing else is
//                stored.  FindIter's chain is  pos -> T(pos) = i* + len(cand(i*)),  i* = first candidate >= pos;  the
//                block collapses it by pointer doubling (inside rows of 64 positions across lanes, from row to row through
//                LDS) to "offset at which a chain that enters this block at offset o leaves it" for the o < L that can
//                occur (a match overshoots a block by less than the longest pattern, L <= 1 024): a function
//                [0, L) -> [0, L) per block.
//   k_ss_group / k_ss_top / k_ss_entries   the chain across blocks is the composition of those functions: composed per
//                group of 256 blocks in parallel, one short serial pass over the groups (from LDS), then every block's
//                true entry offset.  No step of this depends on how long the chain is.
//   k_ss_mark    every block marks the positions its part of the chain visits (the rows it enters, then top-down over the
//                doubling levels inside each), hence the selected candidates: a 1 024-bit mask and a count per block;
//                launch_scan orders the counts.
//   k_ss_emit    records {pattern, start, start + len} at their rank.
// Spans are processed in windows (variant ss_window_kib, default 256 Mi positions) chained through the exit offset.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "start_select.hpp"

namespace acgpu {

namespace {

constexpr uint32_t kB = kSsBlock;       // start positions per block
constexpr uint32_t kG = 256;            // blocks per group

struct SsArgs {
    const uint8_t* hay;        // hay[i] = haystack byte i
    uint64_t span_end;
    uint64_t win_lo;           // absolute position of window offset 0
    uint64_t win_n;            // positions in the window
    const uint32_t* atab;      // trie-only transitions (HotTables::atab), root row = root << ashift
    const uint8_t* acls;
    const uint32_t* own_pid;   // [hid] lowest pattern id that ends exactly in this trie node
    const uint32_t* plens;
    uint32_t ashift, root, L, Lc;
    uint32_t n_states;         // trie nodes (hids); small tries are walked from LDS (trie_lds_words != 0)
    uint32_t trie_lds_words;   // (n_states << ashift) if the whole trie table is staged in LDS, else 0
    int32_t longest;           // 1: LeftmostLongest
    uint32_t* cand;            // [win_n] pattern id + 1, 0 = no occurrence starts here
    uint16_t* e1;              // [nblk][Lc] exit offset per entry offset
    uint64_t nblk;
};

// T[p] for the block's positions, from the next candidate and the candidates' lengths in LDS: the offset (relative to the
// block) at which the search continues after the match chosen from position p; kB + x = the chain leaves the block and
// enters the next one at offset x
__device__ __forceinline__ uint32_t jump_of(uint32_t nc, const uint16_t* s_len) { return nc >= kB ? kB : nc + s_len[nc]; }

// ---- wave-synchronous block kernels: ONE wavefront owns a block of 1 024 start positions, lane l holds positions
// 64 i + l (i = 0..15, "row" i) in registers.  No workgroup barrier anywhere: the first form of these kernels (1 024
// threads per block, pointer doubling through LDS with two barriers per round) spent its time in the barriers -- 2.3 +
// 2.8 ms per 256 Mi positions.  Here the chain inside a row of 64 positions is collapsed by cross-lane reads
// (ds_bpermute: no LDS storage, no bank conflicts), only the hops from row to row (at most 16) go through LDS.
constexpr uint32_t kRows = kB / 64;
constexpr int kSsWaves = 4;            // wavefronts (= blocks of positions) per workgroup

__device__ __forceinline__ void ss_fence() {   // LDS executes a wavefront's operations in order: drain, then keep the compiler from moving across
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// nc[i] = smallest position q >= 64 i + lane of this block that holds a candidate (kB: none), from has[i]
__device__ __forceinline__ void next_candidates(const bool (&has)[kRows], int lane, uint32_t (&nc)[kRows]) {
    uint32_t rs[kRows];
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) {
        uint32_t v = has[i] ? 64 * i + uint32_t(lane) : kB;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_down(v, o, 64);
            if (lane + o < 64) v = v < t ? v : t;
        }
        rs[i] = v;
    }
    uint32_t later = kB;
#pragma unroll
    for (int i = int(kRows) - 1; i >= 0; i--) {
        nc[i] = rs[i] < later ? rs[i] : later;
        const uint32_t rm = uint32_t(__builtin_amdgcn_readfirstlane(int(rs[i])));   // lane 0 holds the row's minimum
        later = later < rm ? later : rm;
    }
}

// x[i] = first position >= the end of row i that the chain from position 64 i + lane reaches (>= kB: it left the block),
// from t[i] = T(position): six rounds of "read my target's value" inside the row
__device__ __forceinline__ void row_exits(const uint32_t (&t)[kRows], uint32_t (&x)[kRows]) {
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) {
        const uint32_t row_end = 64 * (i + 1);
        uint32_t v = t[i];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const uint32_t u = uint32_t(__shfl(int(v), int(v & 63u), 64));   // (v in this row <=> v < row_end: its lane is v & 63)
            v = v < row_end ? u : v;
        }
        x[i] = v;
    }
}

__global__ __launch_bounds__(kSsWaves * 64) void k_ss_block(SsArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_hay_all[kSsWaves][kB + 1024 + 16];
    __shared__ uint16_t s_len_all[kSsWaves][kB], s_x_all[kSsWaves][2][kB];
    __shared__ uint32_t s_root[256];
    __shared__ uint8_t s_acls[256];
    // small tries (the reference's teddy / same / jetscii sets: a few hundred nodes) are walked entirely from LDS: the whole
    // trie table and the nodes' lowest pattern ids, staged once per workgroup (dynamic LDS; nothing for larger automata)
    extern __shared__ uint32_t s_trie[];   // [trie_lds_words] atab | [n_states] own_pid
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s_acls[threadIdx.x] = a.acls[threadIdx.x];                                                   // (kSsWaves * 64 == 256)
    s_root[threadIdx.x] = threadIdx.x < (1u << a.ashift) ? a.atab[(a.root << a.ashift) | threadIdx.x] : 0u;
    const bool small = a.trie_lds_words != 0;
    if (small) {
        for (uint32_t i = threadIdx.x; i < a.trie_lds_words; i += kSsWaves * 64) s_trie[i] = a.atab[i];
        for (uint32_t i = threadIdx.x; i < a.n_states; i += kSsWaves * 64) s_trie[a.trie_lds_words + i] = a.own_pid[i];
    }
    __syncthreads();   // (the only one: the tables are shared, everything below is per wavefront)
    const uint64_t blk = uint64_t(blockIdx.x) * kSsWaves + wave;
    if (blk >= a.nblk) return;
    uint8_t* s_hay = s_hay_all[wave];
    uint16_t* s_len = s_len_all[wave];
    const uint64_t base = a.win_lo + blk * kB;
    for (uint32_t off = 0; off < kB + a.L; off += 1024) {   // the block's bytes and the longest pattern's look-ahead
        const uint32_t idx = off + 16 * uint32_t(lane);
        uint4 q = make_uint4(0, 0, 0, 0);
        if (base + idx + 16 <= a.span_end) {
            __builtin_memcpy(&q, a.hay + base + idx, 16);
        } else {
            uint8_t b[16];
#pragma unroll
            for (int k = 0; k < 16; k++) b[k] = base + idx + k < a.span_end ? a.hay[base + idx + k] : uint8_t(0);
            __builtin_memcpy(&q, b, 16);
        }
        *reinterpret_cast<uint4*>(s_hay + idx) = q;
    }
    ss_fence();
    // (the sixteen walks of a lane one after the other: issuing their steps in lockstep -- sixteen gathers in flight per lane
    // -- was measured and bought nothing: shallow walks end in LDS, deep ones (a dictionary over prose: five trie rows per
    // position from a 38 MB table) are bound by the chip's gather rate, not by their latency)
    bool has[kRows];
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) {
        const uint32_t p = 64 * i + uint32_t(lane);
        uint32_t cand = 0, len = 0;
        const uint64_t v = base + p;
        if (blk * kB + p < a.win_n && v < a.span_end) {
            const uint64_t room = a.span_end - v;
            const uint32_t limit = room < a.L ? uint32_t(room) : a.L;
            uint32_t s = a.root;
            for (uint32_t k = 0; k < limit; k++) {
                const uint32_t c = s_acls[s_hay[p + k]];
                const uint32_t e = k == 0 ? s_root[c] : (small ? s_trie[(s << a.ashift) | c] : a.atab[(s << a.ashift) | c]);
                if (e == 0) break;
                s = e & 0x7FFFFFFFu;
                if (e >> 31) {
                    const uint32_t id1 = (small ? s_trie[a.trie_lds_words + s] : a.own_pid[s]) + 1;
                    if (a.longest || cand == 0 || id1 < cand) { cand = id1; len = k + 1; }
                }
            }
        }
        if (blk * kB + p < a.win_n) a.cand[blk * kB + p] = cand;
        s_len[p] = uint16_t(len);
        has[i] = cand != 0;
    }
    ss_fence();
    uint32_t nc[kRows], t[kRows], x[kRows];
    next_candidates(has, lane, nc);
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) t[i] = jump_of(nc[i], s_len);
    row_exits(t, x);
    // from row to row: at most 16 hops, four doubling rounds through LDS (ping-pong, stops when every chain has left)
    uint16_t* s_x = s_x_all[wave][0];
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) s_x[64 * i + lane] = uint16_t(x[i]);
    for (uint32_t r = 0; r < 4; r++) {
        bool inside = false;
#pragma unroll
        for (uint32_t i = 0; i < kRows; i++) inside |= x[i] < kB;
        if (__ballot(inside) == 0) break;
        ss_fence();
        const uint16_t* src = s_x_all[wave][r & 1];
        uint16_t* dst = s_x_all[wave][(r + 1) & 1];
#pragma unroll
        for (uint32_t i = 0; i < kRows; i++) x[i] = x[i] < kB ? src[x[i]] : x[i];
#pragma unroll
        for (uint32_t i = 0; i < kRows; i++) dst[64 * i + lane] = uint16_t(x[i]);
    }
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) {
        const uint32_t o = 64 * i + uint32_t(lane);
        if (o < a.Lc) a.e1[blk * a.Lc + o] = uint16_t(x[i] - kB);
    }
}

// group maps: gm[g][o] = exit offset of the group's last block for a chain entering its first block at offset o
__global__ __launch_bounds__(256) void k_ss_group(const uint16_t* __restrict__ e1, uint64_t nblk, uint32_t Lc, uint64_t n_groups,
                                                  uint16_t* __restrict__ gm) {
    const uint64_t gid = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    const uint64_t g = gid / Lc;
    if (g >= n_groups) return;
    uint32_t x = uint32_t(gid % Lc);
    const uint64_t b1 = (g + 1) * kG < nblk ? (g + 1) * kG : nblk;
    for (uint64_t b = g * kG; b < b1; b++) x = e1[b * Lc + x];
    gm[gid] = uint16_t(x);
}

// the one serial pass: entry offset of every group; carry[0] = entry offset of the window (in), exit offset (out)
__global__ __launch_bounds__(1024) void k_ss_top(const uint16_t* __restrict__ gm, uint32_t Lc, uint64_t n_groups,
                                                 uint16_t* __restrict__ ge, uint32_t* __restrict__ carry) {
    extern __shared__ uint16_t s_gm[];
    const uint64_t n = n_groups * Lc;
    const bool staged = n * 2 <= 64 * 1024;
    if (staged) {
        for (uint64_t i = threadIdx.x; i < n; i += 1024) s_gm[i] = gm[i];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    uint32_t x = carry[0];
    for (uint64_t g = 0; g < n_groups; g++) {
        ge[g] = uint16_t(x);
        x = staged ? s_gm[g * Lc + x] : gm[g * Lc + x];
    }
    carry[0] = x;
}

__global__ __launch_bounds__(256) void k_ss_entries(const uint16_t* __restrict__ e1, const uint16_t* __restrict__ ge, uint64_t nblk,
                                                    uint32_t Lc, uint64_t n_groups, uint16_t* __restrict__ be) {
    const uint64_t g = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (g >= n_groups) return;
    uint32_t x = ge[g];
    const uint64_t b1 = (g + 1) * kG < nblk ? (g + 1) * kG : nblk;
    for (uint64_t b = g * kG; b < b1; b++) { be[b] = uint16_t(x); x = e1[b * Lc + x]; }
}

// the positions the chain visits inside each block, hence the selected candidates: mask (16 x 64 bits) + count
__global__ __launch_bounds__(kSsWaves * 64) void k_ss_mark(const uint32_t* __restrict__ cand, const uint32_t* __restrict__ plens, uint64_t win_n,
                                                           uint64_t nblk, const uint16_t* __restrict__ be,
                                                           unsigned long long* __restrict__ mask, uint32_t* __restrict__ counts) {
    __shared__ uint16_t s_len_all[kSsWaves][kB], s_x_all[kSsWaves][kB];
    __shared__ uint8_t s_sel_all[kSsWaves][kB], s_flag_all[kSsWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t blk = uint64_t(blockIdx.x) * kSsWaves + wave;
    if (blk >= nblk) return;
    uint16_t* s_len = s_len_all[wave];
    uint16_t* s_x = s_x_all[wave];
    uint8_t* s_sel = s_sel_all[wave];
    uint8_t* s_flag = s_flag_all[wave];
    bool has[kRows];
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) {
        const uint32_t p = 64 * i + uint32_t(lane);
        const uint64_t g = blk * kB + p;
        const uint32_t c = g < win_n ? cand[g] : 0u;
        has[i] = c != 0;
        s_len[p] = c ? uint16_t(plens[c - 1]) : uint16_t(0);
        s_sel[p] = 0;
    }
    ss_fence();
    uint32_t nc[kRows], t[kRows], x[kRows];
    next_candidates(has, lane, nc);
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) t[i] = jump_of(nc[i], s_len);
    row_exits(t, x);
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) s_x[64 * i + lane] = uint16_t(x[i]);
    ss_fence();
    // where the chain enters each row (one short serial walk over the row exits; 64 = it does not), kept in lane `row`
    uint32_t entry = 64;
    for (uint32_t cur = be[blk]; cur < kB; cur = s_x[cur])
        if (uint32_t(lane) == (cur >> 6)) entry = cur & 63u;
    // inside every entered row: the in-row doubling levels again (registers), then the visited lanes top-down through a
    // 64-byte flag array
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) {
        const uint32_t le = uint32_t(__shfl(int(entry), int(i), 64));   // (uniform)
        if (le >= 64) continue;
        const uint32_t row_end = 64 * (i + 1);
        uint32_t lv[6];
        lv[0] = t[i] < row_end ? (t[i] & 63u) : 64u;
#pragma unroll
        for (int k = 1; k < 6; k++) {
            const uint32_t u = uint32_t(__shfl(int(lv[k - 1]), int(lv[k - 1] & 63u), 64));
            lv[k] = lv[k - 1] < 64 ? u : 64u;
        }
        s_flag[lane] = uint32_t(lane) == le ? 1 : 0;
        bool reach = uint32_t(lane) == le;
#pragma unroll
        for (int k = 5; k >= 0; k--) {
            ss_fence();
            if (reach && lv[k] < 64) s_flag[lv[k]] = 1;
            ss_fence();
            reach = s_flag[lane] != 0;
        }
        if (reach && nc[i] < kB) s_sel[nc[i]] = 1;   // the match chosen from a visited position (it may start in a later row)
        ss_fence();
    }
    ss_fence();
    uint32_t n = 0;
#pragma unroll
    for (uint32_t i = 0; i < kRows; i++) {
        const unsigned long long m = __ballot(s_sel[64 * i + lane] != 0);
        if (lane == 0) mask[blk * kRows + i] = m;
        n += uint32_t(__popcll(m));
    }
    if (lane == 0) counts[blk] = n;
}

// records {pattern, start, start + len} at their rank.  One wavefront per block; the records of a row of 64 positions are
// laid out in LDS in rank order and written as consecutive 8-byte words (a lane storing its own 24-byte record writes a
// third of twelve cache lines per instruction; this way an instruction fills four).
__global__ __launch_bounds__(kSsWaves * 64) void k_ss_emit(const uint32_t* __restrict__ cand, const uint32_t* __restrict__ plens, uint64_t win_lo,
                                                           uint64_t nblk, const unsigned long long* __restrict__ mask,
                                                           const uint64_t* __restrict__ offsets, uint64_t out_base, uint64_t cap,
                                                           acgpu_match* __restrict__ out) {
    __shared__ unsigned long long s_rec_all[kSsWaves][64 * 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t blk = uint64_t(blockIdx.x) * kSsWaves + wave;
    if (blk >= nblk) return;
    unsigned long long* s_rec = s_rec_all[wave];
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out);
    uint64_t at = out_base + offsets[blk];   // record index of the row's first record (wave-uniform)
    for (uint32_t i = 0; i < kRows; i++) {
        const unsigned long long m = mask[blk * kRows + i];
        if (m == 0) continue;
        const uint32_t n = uint32_t(__popcll(m));
        if ((m >> lane) & 1ull) {
            const uint32_t r = uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
            const uint64_t p = blk * kB + 64 * i + uint32_t(lane);
            const uint32_t pid = cand[p] - 1;
            const uint64_t start = win_lo + p;
            s_rec[3 * r] = pid;   // {u32 pattern, u32 pad}
            s_rec[3 * r + 1] = start;
            s_rec[3 * r + 2] = start + plens[pid];
        }
        ss_fence();
        const uint64_t room = at < cap ? cap - at : 0;
        const uint32_t words = 3 * uint32_t(room < n ? room : n);
        for (uint32_t w = uint32_t(lane); w < words; w += 64) dst[3 * at + w] = s_rec[w];
        ss_fence();
        at += n;
    }
}

}  // namespace

namespace {
struct SsLayout {
    uint64_t nblk, ng;
    uint32_t Lc;
    uint32_t* carry; uint32_t* cand; unsigned long long* mask; uint16_t* e1; uint16_t* gm; uint16_t* ge; uint16_t* be;
};
SsLayout ss_layout(void* work, uint64_t win_n, uint32_t L) {
    SsLayout y{};
    y.nblk = (win_n + kB - 1) / kB; y.ng = (y.nblk + kG - 1) / kG;
    y.Lc = std::min<uint32_t>(L, kB);
    uint8_t* w = static_cast<uint8_t*>(work);
    auto take = [&](size_t bytes) { uint8_t* p = w; w += (bytes + 255) & ~size_t(255); return p; };
    y.carry = reinterpret_cast<uint32_t*>(take(256));
    y.cand = reinterpret_cast<uint32_t*>(take(y.nblk * kB * 4));
    y.mask = reinterpret_cast<unsigned long long*>(take(y.nblk * (kB / 8)));
    y.e1 = reinterpret_cast<uint16_t*>(take(y.nblk * y.Lc * 2));
    y.gm = reinterpret_cast<uint16_t*>(take(y.ng * y.Lc * 2));
    y.ge = reinterpret_cast<uint16_t*>(take(y.ng * 2));
    y.be = reinterpret_cast<uint16_t*>(take(y.nblk * 2));
    return y;
}
}  // namespace

size_t start_select_work_bytes(uint64_t win_n, uint32_t L) {
    const uint64_t nblk = (win_n + kB - 1) / kB, ng = (nblk + kG - 1) / kG;
    const uint64_t Lc = std::min<uint32_t>(std::max<uint32_t>(L, 1), kB);
    // carry | cand | mask | e1 | gm | ge | be, each rounded up to 256 bytes (ss_layout)
    return size_t(nblk * kB * 4 + nblk * (kB / 8) + nblk * Lc * 2 + ng * Lc * 2 + ng * 2 + nblk * 2 + 8 * 256);
}

hipError_t launch_start_select(const SsTables& t, const uint8_t* hay, uint64_t span_end, uint64_t win_lo, uint64_t win_n,
                               int longest, void* work, bool first_window, const ScanScratch& sc, hipStream_t s) {
    if (t.L == 0 || t.L > kB || win_n == 0) return hipErrorInvalidValue;
    const SsLayout y = ss_layout(work, win_n, t.L);
    if (y.nblk > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipError_t e;
    if (first_window && (e = hipMemsetAsync(y.carry, 0, 4, s)) != hipSuccess) return e;   // the chain enters the span at its first position
    SsArgs a{};
    a.hay = hay; a.span_end = span_end; a.win_lo = win_lo; a.win_n = win_n;
    a.atab = t.atab; a.acls = t.acls; a.own_pid = t.own_pid; a.plens = t.plens;
    a.ashift = t.ashift; a.root = t.root; a.L = t.L; a.Lc = y.Lc; a.longest = longest;
    a.cand = y.cand; a.e1 = y.e1; a.nblk = y.nblk;
    // the whole trie in LDS while it takes at most 16 KiB (three workgroups per CU keep fitting)
    const uint64_t trie_words = uint64_t(t.n_states) << t.ashift;
    a.n_states = t.n_states;
    a.trie_lds_words = t.n_states && (trie_words + t.n_states) * 4 <= 16 * 1024 ? uint32_t(trie_words) : 0u;
    const size_t trie_lds = a.trie_lds_words ? size_t(a.trie_lds_words + t.n_states) * 4 : 0;
    k_ss_block<<<dim3(uint32_t((y.nblk + kSsWaves - 1) / kSsWaves)), dim3(kSsWaves * 64), trie_lds, s>>>(a);
    k_ss_group<<<dim3(uint32_t((y.ng * y.Lc + 255) / 256)), dim3(256), 0, s>>>(y.e1, y.nblk, y.Lc, y.ng, y.gm);
    const size_t top_lds = y.ng * y.Lc * 2 <= 64 * 1024 ? size_t(y.ng * y.Lc * 2) : 0;
    k_ss_top<<<dim3(1), dim3(1024), top_lds, s>>>(y.gm, y.Lc, y.ng, y.ge, y.carry);
    k_ss_entries<<<dim3(uint32_t((y.ng + 255) / 256)), dim3(256), 0, s>>>(y.e1, y.ge, y.nblk, y.Lc, y.ng, y.be);
    k_ss_mark<<<dim3(uint32_t((y.nblk + kSsWaves - 1) / kSsWaves)), dim3(kSsWaves * 64), 0, s>>>(y.cand, t.plens, win_n, y.nblk, y.be, y.mask, sc.counts);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return launch_scan(sc, y.nblk, s);
}

hipError_t launch_start_select_emit(const SsTables& t, uint64_t win_lo, uint64_t win_n, void* work, const ScanScratch& sc,
                                    uint64_t out_base, uint64_t cap, acgpu_match* out, hipStream_t s) {
    const SsLayout y = ss_layout(work, win_n, t.L);
    if (!out || cap <= out_base) return hipSuccess;
    k_ss_emit<<<dim3(uint32_t((y.nblk + kSsWaves - 1) / kSsWaves)), dim3(kSsWaves * 64), 0, s>>>(y.cand, t.plens, win_lo, y.nblk, y.mask, sc.offsets, out_base, cap, out);
    return hipGetLastError();
}

}  // namespace acgpu
