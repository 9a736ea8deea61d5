// Parallel non-overlapping selection over the ordered occurrence stream (rule: select.hpp).
//
// The selected matches form the orbit of the first selection under  succ(i) = "the match selected right after
// occurrence i" , which only looks ahead (succ(i) > i).  succ is computed for all occurrences in parallel; the
// orbit is then found block-wise: inside blocks of 1024 occurrences the chains are collapsed by pointer doubling in
// LDS to "first index past the block", one lane hops from block to block recording where the orbit enters each
// block, and every entered block replays its part of the chain, after which the per-block selections are compacted
// in order with the same count/scan machinery as the match records.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "select.hpp"

namespace acgpu {

namespace {

constexpr uint32_t kSelBlock = 1024;  // occurrences per block
constexpr uint32_t kNone = 0xFFFFFFFFu;

// Length of the occurrence stream as the selection kernels see it (SelectGate, kernels.hpp).
__device__ __forceinline__ uint32_t sel_len(const uint64_t* __restrict__ n_in, const SelectGate& g) {
    if (g.totals && (g.totals[1] > g.max_events || g.totals[0] > g.max_records)) return 0;
    return uint32_t(*n_in);
}

// best occurrence with start >= pos among indices >= i0 (same rule as select_nonoverlapping's inner loop)
__device__ __forceinline__ uint32_t sel_best(const acgpu_match* __restrict__ S, uint32_t M, uint32_t i0, uint64_t pos,
                                             int match_kind, uint64_t L) {
    if (match_kind == ACGPU_MATCH_STANDARD) {
        for (uint32_t j = i0; j < M; j++) if (S[j].start >= pos) return j;
        return kNone;
    }
    uint32_t best = kNone;
    uint64_t bs = 0, be = 0;
    uint32_t bp = 0;
    for (uint32_t j = i0; j < M; j++) {
        const uint64_t ms = S[j].start, me = S[j].end;
        if (best != kNone && me > bs + L) break;
        if (ms < pos) continue;
        bool better = best == kNone || ms < bs;
        if (best != kNone && ms == bs) {
            const uint32_t mp = S[j].pattern;
            if (match_kind == ACGPU_MATCH_LEFTMOST_FIRST) better = mp < bp;
            else { const uint64_t lm = me - ms, lb = be - bs; better = lm > lb || (lm == lb && mp < bp); }
        }
        if (better) { best = j; bs = ms; be = me; bp = S[j].pattern; }
    }
    return best;
}

// succ[i] = the occurrence selected right after occurrence i; exit[i] = first index >= block end reached from i by following
// succ (kNone if the chain ends inside the stream): ten doubling rounds in LDS, one occurrence per thread.  Also the root of
// the orbit and the reset of entry[] (k_sel_hop fills in the blocks the orbit enters).
__global__ __launch_bounds__(kSelBlock) void k_sel_succ_exits(const acgpu_match* __restrict__ S, const uint64_t* __restrict__ n_in,
                                                              int match_kind, uint64_t span_start, uint64_t L,
                                                              uint32_t* __restrict__ succ, uint32_t* __restrict__ root,
                                                              uint32_t* __restrict__ entry, uint32_t* __restrict__ exitp,
                                                              uint32_t* __restrict__ unresolved, SelectGate gate) {
    __shared__ uint32_t jump[kSelBlock];
    const uint32_t M = sel_len(n_in, gate);
    const uint32_t b0 = blockIdx.x * kSelBlock, b1 = b0 + kSelBlock, k = threadIdx.x, i = b0 + k;
    if (k == 0) entry[blockIdx.x] = kNone;
    if (i == 0) { *root = sel_best(S, M, 0, span_start, match_kind, L); *unresolved = 0; }
    uint32_t j = kNone;
    if (i < M) { j = sel_best(S, M, i + 1, S[i].end, match_kind, L); succ[i] = j; }
    jump[k] = j;
    __syncthreads();
    for (int round = 0; round < 10; round++) {  // chains strictly increase: 2^10 hops cover a block
        const uint32_t nj = (j != kNone && j < b1) ? jump[j - b0] : j;
        __syncthreads();
        jump[k] = j = nj;
        __syncthreads();
    }
    if (i < M) exitp[i] = j;
}

// Where does the orbit enter each block?  In parallel, from the BREAKS of the stream: an index i such that no occurrence
// crosses P = S[i-1].end (every earlier one ends at or before P -- the stream is ordered by end -- and every later one
// starts at or behind it).  Whatever was selected before, the iteration (FindIter, src/automaton.rs:857-936) is in its
// initial condition there: the first match taken at or behind a break is sel_best(i, P), a member of the orbit known without
// the orbit's history.  One thread per block finds the block's first break, that member r, and where the chain from r
// leaves the block (exit[r]) -- the entry of a later block.  Natural text has a break every few occurrences; a block
// without one (a periodic text under a self-overlapping pattern) sets `unresolved`, and the serial hop below runs instead.
// Round 5's hop alone: one dependent load per block, 0.24 us each -- 0.5 ms for 2 M occurrences (profiles/r06_call_timelines_before.txt).
__global__ __launch_bounds__(256) void k_sel_entries(const acgpu_match* __restrict__ S, const uint64_t* __restrict__ n_in, int match_kind,
                                                     uint64_t L, const uint32_t* __restrict__ exitp, const uint32_t* __restrict__ root,
                                                     uint32_t* __restrict__ entry, uint32_t* __restrict__ unresolved, uint32_t nblocks,
                                                     SelectGate gate) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const uint32_t M = sel_len(n_in, gate);
    if (b >= nblocks || uint64_t(b) * kSelBlock >= M) return;
    const uint32_t b0 = b * kSelBlock, b1 = b0 + kSelBlock < M ? b0 + kSelBlock : M;
    if (b == 0) { const uint32_t r = *root; if (r != kNone && r < M) entry[r / kSelBlock] = r; }
    if (b1 == M) return;   // (the last block hands the orbit to nobody)
    uint32_t brk = kNone;
    uint32_t budget = 4 * kSelBlock;   // records looked at before the block is given up as break-free
    for (uint32_t i = b0 > 0 ? b0 : 1; i < b1 && brk == kNone && budget; i++) {
        const uint64_t P = S[i - 1].end;
        bool ok = true;
        for (uint32_t j = i; j < M && S[j].end < P + L && budget; j++, budget--)
            if (S[j].start < P) { ok = false; break; }
        if (budget) budget--;
        if (ok && budget) brk = i;
    }
    if (brk == kNone) { *unresolved = 1; return; }
    const uint32_t r = sel_best(S, M, brk, S[brk - 1].end, match_kind, L);
    if (r == kNone) return;                                         // the orbit ends before the next block
    if (r >= b0 + kSelBlock) { entry[r / kSelBlock] = r; return; }  // nothing is taken between the break and r: r enters its block
    const uint32_t e = exitp[r];
    if (e != kNone && e < M) entry[e / kSelBlock] = e;
}

// one lane: where does the orbit enter each block?  (Only when a block had no break: k_sel_entries.)
__global__ void k_sel_hop(const uint32_t* __restrict__ exitp, const uint32_t* __restrict__ root,
                          const uint64_t* __restrict__ n_in, uint32_t* __restrict__ entry, uint32_t nblocks,
                          const uint32_t* __restrict__ unresolved, SelectGate gate) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || *unresolved == 0) return;
    const uint32_t M = sel_len(n_in, gate);
    uint32_t cur = *root;
    while (cur != kNone && cur < M) {
        entry[cur / kSelBlock] = cur;
        cur = exitp[cur];
    }
}

// every entered block marks the part of the chain that runs through it; the selected indices are stored compactly per
// block, the counts feed the scan.  Not a serial replay (config 5 selects nearly every occurrence: a thousand dependent
// steps per block, 58-68 us, the longest kernel of the selection): ten doubling levels of the block's succ pointers in LDS,
// the visited entries marked top-down from the entry, compacted with ballots.
__global__ __launch_bounds__(kSelBlock) void k_sel_mark(const uint32_t* __restrict__ succ, const uint32_t* __restrict__ entry,
                                                        const uint64_t* __restrict__ n_in, uint32_t* __restrict__ sel_idx,
                                                        uint32_t* __restrict__ counts, SelectGate gate) {
    __shared__ uint16_t s_lv[10][kSelBlock];   // level k: 2^k hops from the entry's index, kSelBlock = left the block
    __shared__ uint8_t s_reach[kSelBlock];
    __shared__ uint32_t s_wave[kSelBlock / 64];
    const uint32_t b = blockIdx.x, b0 = b * kSelBlock, b1 = b0 + kSelBlock, p = threadIdx.x;
    const uint32_t start = entry[b];
    if (start == kNone) { if (p == 0) counts[b] = 0; return; }   // (block-uniform: the chain does not enter this block)
    const uint32_t M = sel_len(n_in, gate);
    const uint32_t nx = b0 + p < M ? succ[b0 + p] : kNone;
    s_lv[0][p] = uint16_t(nx != kNone && nx < b1 ? nx - b0 : kSelBlock);
    s_reach[p] = p == start - b0 ? 1 : 0;
    __syncthreads();
    for (int k = 1; k < 10; k++) {
        const uint32_t j = s_lv[k - 1][p];
        s_lv[k][p] = j < kSelBlock ? s_lv[k - 1][j] : uint16_t(kSelBlock);
        __syncthreads();
    }
    for (int k = 9; k >= 0; k--) {
        if (s_reach[p]) { const uint32_t j = s_lv[k][p]; if (j < kSelBlock) s_reach[j] = 1; }
        __syncthreads();
    }
    const bool sel = s_reach[p] != 0;
    const unsigned long long m = __ballot(sel);
    const uint32_t lane = p & 63, wave = p >> 6;
    if (lane == 0) s_wave[wave] = uint32_t(__popcll(m));
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < kSelBlock / 64; w++) { const uint32_t c = s_wave[w]; if (w < wave) before += c; total += c; }
    if (sel) sel_idx[b0 + before + uint32_t(__popcll(m & ((1ull << lane) - 1ull)))] = b0 + p;
    if (p == 0) counts[b] = total;
}

__global__ __launch_bounds__(256) void k_sel_scatter(const acgpu_match* __restrict__ S, const uint32_t* __restrict__ sel_idx,
                                                     const uint32_t* __restrict__ counts, const uint64_t* __restrict__ offsets,
                                                     acgpu_match* __restrict__ out, uint64_t cap) {
    const uint32_t b = blockIdx.x, n = counts[b];
    const uint64_t o = offsets[b];
    for (uint32_t k = threadIdx.x; k < n; k += 256)
        if (o + k < cap) out[o + k] = S[sel_idx[b * kSelBlock + k]];
}

// Streams of up to kSelFewBlocks blocks (config 5: 45 blocks): the block's offset is the sum of the counts before it -- no
// scan launches; the last block reports the number of selected records.  (Nothing in this kernel reads the stream length,
// so `total` may be the word it was read from.)
constexpr uint32_t kSelFewBlocks = uint32_t(kSelectFewLimit / kSelBlock);
__global__ __launch_bounds__(256) void k_sel_scatter_few(const acgpu_match* __restrict__ S, const uint32_t* __restrict__ sel_idx,
                                                         const uint32_t* __restrict__ counts, acgpu_match* __restrict__ out,
                                                         uint64_t cap, uint64_t* __restrict__ total, SelectGate gate) {
    __shared__ uint32_t s_part[4];
    const uint32_t b = blockIdx.x, n = counts[b];
    uint32_t before = 0;
    for (uint32_t k = threadIdx.x; k < b; k += 256) before += counts[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += uint32_t(__shfl_xor(int(before), o, 64));
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = before;
    __syncthreads();
    const uint64_t o = uint64_t(s_part[0]) + s_part[1] + s_part[2] + s_part[3];
    for (uint32_t k = threadIdx.x; k < n; k += 256)
        if (o + k < cap) out[o + k] = S[sel_idx[b * kSelBlock + k]];
    if (b + 1 == gridDim.x && threadIdx.x == 0) {
        *total = o + n;
        if (gate.host) { gate.host[0] = gate.totals[0]; gate.host[1] = gate.totals[1]; gate.host[2] = o + n; }
    }
}

}  // namespace

size_t select_scratch_bytes(uint64_t m) {
    const uint64_t nb = (m + kSelBlock - 1) / kSelBlock;
    return size_t(3 * m * 4 + nb * 4 + 64);   // succ, exit, sel_idx | entry | root, unresolved
}

// S: ordered occurrence stream (device), m: its length (host copy; the kernels read *n_in on the device, equal to m).
// `work` = select_scratch_bytes(m) bytes of device scratch.  The selected records go to out[0..cap), their number
// to sc.totals[0] (read it after the stream has drained).
hipError_t launch_select_parallel(const acgpu_match* S, uint64_t m, const uint64_t* n_in, int match_kind,
                                  uint64_t span_start, uint64_t L, void* work, const ScanScratch& sc, acgpu_match* out,
                                  uint64_t cap, hipStream_t s, SelectGate gate) {
    if (m == 0 || m >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const uint32_t nb = uint32_t((m + kSelBlock - 1) / kSelBlock);
    uint32_t* succ = static_cast<uint32_t*>(work);
    uint32_t* exitp = succ + m;
    uint32_t* sel_idx = exitp + m;
    uint32_t* entry = sel_idx + m;
    uint32_t* root = entry + nb;
    uint32_t* unresolved = root + 1;
    hipError_t e = hipSuccess;
    k_sel_succ_exits<<<dim3(nb), dim3(kSelBlock), 0, s>>>(S, n_in, match_kind, span_start, L, succ, root, entry, exitp, unresolved, gate);
    k_sel_entries<<<dim3((nb + 255) / 256), dim3(256), 0, s>>>(S, n_in, match_kind, L, exitp, root, entry, unresolved, nb, gate);
    k_sel_hop<<<dim3(1), dim3(64), 0, s>>>(exitp, root, n_in, entry, nb, unresolved, gate);
    k_sel_mark<<<dim3(nb), dim3(kSelBlock), 0, s>>>(succ, entry, n_in, sel_idx, sc.counts, gate);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (nb <= kSelFewBlocks) {
        k_sel_scatter_few<<<dim3(nb), dim3(256), 0, s>>>(S, sel_idx, sc.counts, out, cap, sc.totals, gate);
        return hipGetLastError();
    }
    if (gate.totals) return hipErrorInvalidValue;   // (guessed streams are small: the form above)
    if ((e = launch_scan(sc, nb, s)) != hipSuccess) return e;
    k_sel_scatter<<<dim3(nb), dim3(256), 0, s>>>(S, sel_idx, sc.counts, sc.offsets, out, cap);
    return hipGetLastError();
}

}  // namespace acgpu
