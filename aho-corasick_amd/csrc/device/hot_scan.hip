// Fast-path tables of the Standard/unanchored full DFA (hid renumbering, 256-wide u16 table, prefix-filter tables)
// and the LDS-row FILL kernel of the count -> scan -> fill pipeline.  The count kernels live in lds_walk.hip (LDS
// transition walk) and pf_scan.hip (prefix filter).
// Match states are recognised by one compare (hid >= first_match) exactly like the reference's
// `sid <= max_special_id` trick (src/dfa.rs:229-241), just with the order reversed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <utility>

#include "../host/lw_tables.hpp"
#include "hot.hpp"
#include "launch_util.hpp"
#include "tile_walk.hpp"

namespace acgpu {

namespace {

constexpr uint32_t kMaxHotRows = 192;  // 192 x 512 B = 96 KiB of LDS

// ------------------------------------------------------------------------------------- fill
// Same job as k_walk_fill (kernels.hip) for automata that have hot tables: one wavefront per non-empty chunk, the
// chunk and its warm-up staged in LDS, lane l walks sub-range l once and remembers its first match events.  The walk
// itself never leaves LDS on the common path: byte -> class -> class-compressed u16 row of the start state / the
// distance-1 states (rebuilt per workgroup from the 256-wide table); deeper states read the global table.  A step
// is three dependent LDS reads (~0.3 us) instead of an L2 round trip (~1.25 us measured), and the match records are
// produced from the reference's own match lists (hid -> DFA state id), so the order inside one `end` is the state's
// match-list order exactly as in k_walk_fill.
constexpr int kHfWaves = 16;
constexpr uint32_t kHfStage = 2048 + 512;   // staged bytes per wavefront (chunk + warm-up + alignment slack)
constexpr int kHfEvents = 2;

__global__ __launch_bounds__(kHfWaves * 64) void k_hot_fill(DfaEng eng, const uint16_t* __restrict__ tab,
                                                            const uint32_t* __restrict__ hid2sid, uint32_t n_hot,
                                                            uint32_t first_match, uint32_t start, ScanGeom g,
                                                            const uint64_t* __restrict__ active,
                                                            const uint64_t* __restrict__ totals, uint64_t cap,
                                                            const uint64_t* __restrict__ aoff,
                                                            acgpu_match* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint8_t s_cls[256];
    __shared__ uint8_t s_rep[256];
    const uint32_t ncls = 1u << eng.d.stride2;                                   // row stride of the compressed table
    uint16_t* s_ctab = reinterpret_cast<uint16_t*>(smem);                        // [n_hot][ncls]
    uint8_t* s_hay_all = smem + ((size_t(n_hot) * ncls * 2 + 15) & ~size_t(15));  // kHfWaves * kHfStage
    const uint64_t n_active = totals[1];
    if (totals[0] > cap || uint64_t(blockIdx.x) * kHfWaves >= n_active) return;
    if (threadIdx.x < 256) {
        const uint8_t c = eng.cls[threadIdx.x];
        s_cls[threadIdx.x] = c;
        s_rep[c] = uint8_t(threadIdx.x);   // any byte of the class: they share every transition
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_hot * ncls; i += kHfWaves * 64) {
        const uint32_t h = i >> eng.d.stride2, c = i & (ncls - 1);
        s_ctab[i] = tab[(h << 8) | s_rep[c]];   // classes beyond alphabet_len are never looked up
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* s_hay = s_hay_all + size_t(wave) * kHfStage;
    auto write = [&](uint32_t sid, uint64_t end, acgpu_match* dst) -> uint32_t {
        const uint32_t n = eng.match_len(sid);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t pid = eng.match_pattern(sid, i);
            acgpu_match m; m.pattern = pid; m._pad = 0; m.end = end; m.start = end - eng.pattern_len(pid);
            dst[i] = m;
        }
        return n;
    };
    for (uint64_t a = uint64_t(blockIdx.x) * kHfWaves + wave; a < n_active; a += uint64_t(gridDim.x) * kHfWaves) {
        const uint64_t ci = active[a];
        const ChunkRange r = chunk_range(g, ci);
        const uint64_t w16 = r.w & ~uint64_t(15);
        for (uint64_t o = uint64_t(lane) * 16; w16 + o < r.hi; o += 64 * 16) {   // host guarantees r.hi - w16 <= kHfStage
            ACGPU_HAY_CHECK(g, w16 + o, 16);
            *reinterpret_cast<uint4*>(s_hay + o) = *reinterpret_cast<const uint4*>(g.hay16 + w16 + o);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // split [r.lo, r.hi) into 64 sub-ranges of `sub` bytes (the last ones may be empty)
        const uint64_t len = r.hi - r.lo;
        const uint64_t sub = (len + 63) / 64;
        uint64_t lo = r.lo + uint64_t(lane) * sub, hi = lo + sub;
        if (lo > r.hi) lo = r.hi;
        if (hi > r.hi) hi = r.hi;
        uint64_t w = lo >= g.halo ? lo - g.halo : 0;
        if (w < g.cold_floor) w = g.cold_floor;
        const bool sm = ci == 0 && lane == 0 && g.emit_start_matches && start >= first_match;
        uint64_t ev_end[kHfEvents] = {};
        uint32_t ev_hid[kHfEvents] = {}, nev = 0, c = 0;
        auto walk = [&](bool emit_now, acgpu_match* dst) {
            uint32_t n_out = 0, k_ev = 0;
            auto event = [&](uint32_t h, uint64_t end) {
                if (emit_now) { n_out += write(hid2sid[h], end, dst + n_out); return; }
#pragma unroll
                for (int k = 0; k < kHfEvents; k++) if (k_ev == uint32_t(k)) { ev_end[k] = end; ev_hid[k] = h; }
                k_ev++;
                n_out += eng.match_len(hid2sid[h]);
            };
            if (sm) event(start, g.cold_floor - g.base_mis);
            uint32_t h = start;
            if (hi > lo)
                for (uint64_t v = w; v < hi; v++) {
                    const uint32_t byte = s_hay[v - w16];
                    h = h < n_hot ? s_ctab[(h << eng.d.stride2) | s_cls[byte]] : tab[(h << 8) | byte];
                    if (h >= first_match && v >= lo) event(h, v + 1 - g.base_mis);
                }
            nev = k_ev;
            return n_out;
        };
        c = walk(false, nullptr);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (c) {
            acgpu_match* dst = out + aoff[a] + (incl - c);
            if (nev <= uint32_t(kHfEvents)) {
#pragma unroll
                for (int k = 0; k < kHfEvents; k++)
                    if (uint32_t(k) < nev) dst += write(hid2sid[ev_hid[k]], ev_end[k], dst);
            } else {
                (void)walk(true, dst);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // s_hay is reused by this wave's next chunk
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

hipError_t build_hot_tables(const NNfa& n, const Dfa& d, HotTables& out) {
    out.ready = false;
    std::vector<uint32_t> order, sid2hid;   // hid -> nnfa sid and back (host/lw_tables.cpp)
    uint32_t first_match = 0;
    hid_order(n, order, sid2hid, first_match);
    const uint32_t su = n.special.start_unanchored_id, sa = n.special.start_anchored_id;
    const size_t nh = order.size();
    const bool small = nh <= 65535;   // the 256-wide u16 table of the LDS-row engines needs 16-bit state ids
    if (nh > kPfMaxStates) return hipSuccess;
    // hot rows: start state + non-match states at distance 1 (stored depth 0), capped by the LDS budget
    uint32_t n_hot = 1;  // DEAD row is row 0 (all zeros); it is never looked up, but keeps indices simple
    for (size_t h = 1; h < first_match; h++) {
        const uint32_t s = order[h];
        const bool shallow = (s == su) || n.depth[s] == 0;
        if (!shallow || n_hot >= kMaxHotRows) break;
        n_hot++;
    }
    std::vector<uint16_t> tab(small ? nh * 256 : 0, 0);
    std::vector<uint32_t> hid2sid(nh, 0);
    for (size_t h = 1; h < nh; h++) {
        const uint32_t s = order[h];
        hid2sid[h] = s << d.stride2;
        if (!small) continue;
        const uint32_t* row = &d.trans[size_t(s) << d.stride2];
        for (int b = 0; b < 256; b++) tab[h * 256 + b] = uint16_t(sid2hid[row[d.byte_classes[b]] >> d.stride2]);
    }
    hipError_t e;
    if (small) {
        if ((e = hipMalloc(reinterpret_cast<void**>(&out.tab), tab.size() * sizeof(uint16_t))) != hipSuccess) return e;
        if ((e = hipMemcpy(out.tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.hid2sid), hid2sid.size() * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipMemcpy(out.hid2sid, hid2sid.data(), hid2sid.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.n_states = uint32_t(nh);
    out.first_match = first_match;
    out.n_hot = n_hot;
    out.start = sid2hid[su];
    out.ready = small;
    if ((e = build_lw_tables(n, d, order, sid2hid, first_match, out)) != hipSuccess) return e;

    // ---- prefix-filter tables (pf_scan.hip): only without empty patterns, and while the 64 KiB Bloom table stays
    // selective (two entries per pattern in 512 Ki bits: <= 6 % fill)
    out.pf_ready = false;
    if (n.min_pattern_len == 0 || n.pattern_lens.empty() || n.pattern_lens.size() > kPfMaxPatterns) return hipSuccess;
    auto is_trie_child = [&](uint32_t parent, uint32_t k) {  // transition k of `parent` is a trie edge
        const uint32_t t = n.tnext[k];
        return t != kFail && t != kDead && t != su && t != sa && (parent != su || t != su);
    };
    std::vector<uint32_t> own(nh, 0);
    for (size_t h = 1; h < nh; h++) {
        const uint32_t s = order[h];
        if (s == su) continue;
        const uint32_t dist = n.depth[s] + 1;
        for (uint32_t k = n.moff[s]; k < n.moff[s + 1]; k++)
            if (n.pattern_lens[n.mpid[k]] == dist) own[h]++;
    }
    // trie-only transition table, class-compressed with the table's own class map: class 0 = bytes on no trie edge, every
    // byte that labels an edge gets a class of its own.  Rows of 2^ashift entries instead of 256: 128 B per state for
    // lower-case dictionaries, 512 B for printable ASCII -- level 3 of the filters walks it with dependent gathers, and
    // whether those hit L2 / MALL or go to HBM is most of their cost on inputs full of true prefix matches
    std::vector<uint8_t> acls(256, 0);
    uint32_t n_acls = 1;
    {
        bool used[256] = {false};
        for (size_t h = 1; h < nh; h++) {
            const uint32_t s = order[h];
            for (uint32_t k = n.toff[s]; k < n.toff[s + 1]; k++) if (is_trie_child(s, k)) used[n.tbyte[k]] = true;
        }
        for (int b = 0; b < 256; b++) if (used[b]) acls[b] = uint8_t(n_acls++ & 0xFF);
    }
    uint32_t ashift = 0;
    while ((1u << ashift) < n_acls) ashift++;
    if (n_acls > 255) { ashift = 8; for (int b = 0; b < 256; b++) acls[b] = uint8_t(b); }   // (every byte labels an edge: identity map)
    std::vector<uint32_t> atab(nh << ashift, 0);
    for (size_t h = 1; h < nh; h++) {
        const uint32_t s = order[h];
        for (uint32_t k = n.toff[s]; k < n.toff[s + 1]; k++) {
            if (!is_trie_child(s, k)) continue;
            const uint32_t ch = sid2hid[n.tnext[k]];
            atab[(h << ashift) + acls[n.tbyte[k]]] = ch | (own[ch] ? 0x80000000u : 0u);
        }
    }
    // first-level Bloom table (64 KiB of 32-bit words), probed at every other haystack position q only, with the
    // word addressed by a hash of b[q+1..q+3].  Every pattern occurrence starts either at a probed q ("type 0":
    // its bytes 1..3 are the key, its byte 0 selects the bit, tested with b[q]) or at q+1 ("type 1": its bytes
    // 0..2 are the key, its byte 3 selects the bit, tested with b[q+4]).  Patterns shorter than four bytes fill in
    // every value of the bytes they do not have.
    // A second table of the same construction under an unrelated hash (pf_hash2, kPfBits2Bytes) is probed only for the
    // survivors of the first one: a false positive of one table passes the other with its fill probability.
    const uint32_t bits_bytes = 64 * 1024;
    // Large sets (HotTables::pf_exact2): the second table holds one entry per pattern keyed by its true start instead
    // (filled after this loop), so here only the first table is written.
    const bool exact2 = n.pattern_lens.size() > kPfExact2Patterns;
    out.pf_exact2 = exact2;
    std::vector<uint32_t> bits(bits_bytes / 4, 0), bits2(kPfBits2Bytes / 4, 0);
    uint32_t sink = 0;
    struct TwoWords {   // the word of the key in both tables
        uint32_t &w1, &w2;
        void operator=(uint32_t v) { w1 = v; w2 = v; }
        void operator|=(uint32_t v) { w1 |= v; w2 |= v; }
    };
    auto word_of = [&](uint32_t b0, uint32_t b1, uint32_t b2) -> TwoWords {
        const uint32_t key = b0 | (b1 << 8) | (b2 << 16);
        return TwoWords{bits[(pf_hash(key) & (bits_bytes - 1)) >> 2],
                        exact2 ? sink : bits2[(pf_hash2(key) & (kPfBits2Bytes - 1)) >> 2]};
    };
    auto bit_of = [](uint32_t b) { return 1u << (31 - (b & 31)); };
    // third table (HBM / L2): exact first four bytes of every pattern, ~64 bits per pattern
    const bool use_x = n.min_pattern_len >= 4 && n.pattern_lens.size() >= 256;   // pfx_scan.hip tables
    const bool use3 = (n.pattern_lens.size() >= kPfBits3Patterns && n.min_pattern_len >= 3) || use_x;
    std::vector<uint32_t> xbits(use_x ? kPfxBitsBytes / 4 : 0, 0);
    std::vector<std::pair<uint32_t, uint32_t>> xkeys;   // (first four bytes, depth-4 node | own flag)
    uint32_t log3 = 20;
    while (use3 && log3 < 28 && (uint64_t(1) << log3) < uint64_t(n.pattern_lens.size()) * 64) log3++;
    std::vector<uint32_t> bits3(use3 ? (size_t(1) << log3) / 32 : 0, 0);
    auto set3 = [&](uint32_t key4) { const uint32_t h = pf_hash3(key4, log3); bits3[h >> 5] |= 1u << (h & 31); };
    for (uint32_t k = n.toff[su]; k < n.toff[su + 1]; k++) {
        if (!is_trie_child(su, k)) continue;
        const uint32_t b0 = n.tbyte[k], n1 = n.tnext[k];
        if (own[sid2hid[n1]]) {  // 1-byte pattern
            for (uint32_t yz = 0; yz < 65536; yz++) word_of(b0, yz & 0xFF, yz >> 8) = 0xFFFFFFFFu;  // type 1: key (b0,*,*)
            for (auto& w : bits) w |= bit_of(b0);                                                  // type 0: any key
            if (!exact2) for (auto& w : bits2) w |= bit_of(b0);
        }
        for (uint32_t k2 = n.toff[n1]; k2 < n.toff[n1 + 1]; k2++) {
            const uint32_t b1 = n.tbyte[k2], n2 = n.tnext[k2];
            if (own[sid2hid[n2]]) {  // 2-byte pattern
                for (uint32_t z = 0; z < 256; z++) word_of(b0, b1, z) = 0xFFFFFFFFu;                 // type 1: key (b0,b1,*)
                for (uint32_t yz = 0; yz < 65536; yz++) word_of(b1, yz & 0xFF, yz >> 8) |= bit_of(b0);  // type 0: key (b1,*,*)
            }
            for (uint32_t k3 = n.toff[n2]; k3 < n.toff[n2 + 1]; k3++) {
                const uint32_t b2 = n.tbyte[k3], n3 = n.tnext[k3];
                if (own[sid2hid[n3]]) {  // 3-byte pattern
                    word_of(b0, b1, b2) = 0xFFFFFFFFu;                                         // type 1: any 4th byte
                    for (uint32_t z = 0; z < 256; z++) word_of(b1, b2, z) |= bit_of(b0);         // type 0: key (b1,b2,*)
                    if (use3) for (uint32_t z = 0; z < 256; z++) set3(b0 | (b1 << 8) | (b2 << 16) | (z << 24));
                }
                for (uint32_t k4 = n.toff[n3]; k4 < n.toff[n3 + 1]; k4++) {
                    const uint32_t b3 = n.tbyte[k4];
                    word_of(b0, b1, b2) |= bit_of(b3);  // type 1
                    word_of(b1, b2, b3) |= bit_of(b0);  // type 0
                    if (use3) set3(b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
                    if (use_x) {
                        const uint32_t key4 = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24), hx = pfx_hash(key4);
                        xbits[pfx_word(hx)] |= pfx_mask(hx);
                        const uint32_t h4 = sid2hid[n.tnext[k4]];
                        xkeys.emplace_back(key4, h4 | (own[h4] ? 0x80000000u : 0u));
                    }
                }
            }
        }
    }
    if (exact2) {   // one entry per trie path of depth <= 4 from the start state, keyed by the true start
        auto word2_of = [&](uint32_t b0, uint32_t b1, uint32_t b2) -> uint32_t& {
            return bits2[(pf_hash2(b0 | (b1 << 8) | (b2 << 16)) & (kPfBits2Bytes - 1)) >> 2];
        };
        for (uint32_t k = n.toff[su]; k < n.toff[su + 1]; k++) {
            if (!is_trie_child(su, k)) continue;
            const uint32_t b0 = n.tbyte[k], n1 = n.tnext[k];
            if (own[sid2hid[n1]]) for (uint32_t yz = 0; yz < 65536; yz++) word2_of(b0, yz & 0xFF, yz >> 8) = 0xFFFFFFFFu;
            for (uint32_t k2 = n.toff[n1]; k2 < n.toff[n1 + 1]; k2++) {
                const uint32_t b1 = n.tbyte[k2], n2 = n.tnext[k2];
                if (own[sid2hid[n2]]) for (uint32_t z = 0; z < 256; z++) word2_of(b0, b1, z) = 0xFFFFFFFFu;
                for (uint32_t k3 = n.toff[n2]; k3 < n.toff[n2 + 1]; k3++) {
                    const uint32_t b2 = n.tbyte[k3], n3 = n.tnext[k3];
                    if (own[sid2hid[n3]]) word2_of(b0, b1, b2) = 0xFFFFFFFFu;
                    for (uint32_t k4 = n.toff[n3]; k4 < n.toff[n3 + 1]; k4++) word2_of(b0, b1, b2) |= bit_of(n.tbyte[k4]);
                }
            }
        }
    }
    if (use3) {
        if ((e = hipMalloc(reinterpret_cast<void**>(&out.pf_bits3), bits3.size() * 4)) != hipSuccess) return e;
        if ((e = hipMemcpy(out.pf_bits3, bits3.data(), bits3.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
        out.pf_bits3_log2 = log3;
    }
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.pf_bits2), kPfBits2Bytes)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.pf_bits2, bits2.data(), kPfBits2Bytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.pf_bits), bits_bytes)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.pf_bits, bits.data(), bits_bytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.pf_bits_bytes = bits_bytes;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.atab), atab.size() * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.own_cnt), own.size() * 4)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.atab, atab.data(), atab.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&out.acls), 256)) != hipSuccess) return e;
    if ((e = hipMemcpy(out.acls, acls.data(), 256, hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.ashift = ashift;
    if ((e = hipMemcpy(out.own_cnt, own.data(), own.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
    out.n_patterns = uint32_t(n.pattern_lens.size());
    if (use_x) {
        uint32_t lg = 10;   // buckets of two slots, load <= 1/8
        while ((size_t(2) << lg) < xkeys.size() * 8) lg++;
        const uint32_t nb = 1u << lg;
        std::vector<uint32_t> map(size_t(nb) * 4, 0);   // per bucket: key0, val0, key1, val1
        for (const auto& kv : xkeys) {
            for (uint32_t b = pfx_map_bucket(kv.first, lg);; b = (b + 1) & (nb - 1)) {
                uint32_t* q = &map[size_t(b) * 4];
                if (q[1] == 0) { q[0] = kv.first; q[1] = kv.second; break; }
                if (q[3] == 0) { q[2] = kv.first; q[3] = kv.second; break; }
                q[1] |= kPfxMapOverflow;   // a key that belongs here lives further on: lookups that miss here go on
            }
        }
        if ((e = hipMalloc(reinterpret_cast<void**>(&out.pfx_map), map.size() * 4)) != hipSuccess) return e;
        if ((e = hipMemcpy(out.pfx_map, map.data(), map.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
        out.pfx_map_log2 = lg;
        // the long-prefix map: every trie path of length `depth` from the start state (depth <= shortest pattern, so every
        // pattern passes through exactly one of them)
        const uint32_t depth = uint32_t(std::min<size_t>(8, n.min_pattern_len));
        static const bool no_long = std::getenv("ACGPU_PFX_NO_LONG_KEY") != nullptr;   // A/B knob
        if (depth > 4 && !no_long) {
            struct Path { uint32_t lo, hi, node; };
            std::vector<Path> paths;
            struct Frame { uint32_t sid, d; uint64_t key; };
            std::vector<Frame> stack{{su, 0, 0}};
            while (!stack.empty()) {
                const Frame f = stack.back(); stack.pop_back();
                if (f.d == depth) {
                    const uint32_t hd = sid2hid[f.sid];
                    paths.push_back({uint32_t(f.key), uint32_t(f.key >> 32), hd | (own[hd] ? 0x80000000u : 0u)});
                    continue;
                }
                for (uint32_t k = n.toff[f.sid]; k < n.toff[f.sid + 1]; k++)
                    if (is_trie_child(f.sid, k)) stack.push_back({n.tnext[k], f.d + 1, f.key | (uint64_t(n.tbyte[k]) << (8 * f.d))});
            }
            // one entry per bucket, load <= 1/32 (1/8 beyond 2^17 prefixes: 64 MiB at most).  A lookup that misses in a bucket
            // carrying the overflow mark must look further, and a verifier round waits for the slowest of its 256 lookups:
            // at load 1/8 about 1 % of the buckets are marked and nine rounds in ten paid a second dependent gather
            uint32_t lg8 = 10;
            while ((size_t(1) << lg8) < paths.size() * (paths.size() <= (size_t(1) << 17) ? 32 : 8)) lg8++;
            const uint32_t nb8 = 1u << lg8;
            std::vector<uint32_t> map8(size_t(nb8) * 4, 0);   // per bucket: bytes 0..3, bytes 4..7, value, 0
            for (const Path& pt : paths) {
                for (uint32_t b = pfx_map8_bucket(pt.lo, pt.hi, lg8);; b = (b + 1) & (nb8 - 1)) {
                    uint32_t* q = &map8[size_t(b) * 4];
                    if ((q[2] & ~kPfxMapOverflow) == 0) { q[0] = pt.lo; q[1] = pt.hi; q[2] |= pt.node; break; }
                    q[2] |= kPfxMapOverflow;
                }
            }
            if ((e = hipMalloc(reinterpret_cast<void**>(&out.pfx_map8), map8.size() * 4)) != hipSuccess) return e;
            if ((e = hipMemcpy(out.pfx_map8, map8.data(), map8.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
            out.pfx_map8_log2 = lg8;
            out.pfx_depth = depth;
        }
        if ((e = hipMalloc(reinterpret_cast<void**>(&out.pfx_bits), kPfxBitsBytes)) != hipSuccess) return e;
        if ((e = hipMemcpy(out.pfx_bits, xbits.data(), kPfxBitsBytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
        out.pfx_ready = true;
    }
    out.pf_ready = true;
    return hipSuccess;
}

bool hot_fill_supported(const HotTables& h, const ScanGeom& g) {
    if (!h.ready) return false;
    return uint64_t(g.chunk) + g.halo + 32 <= kHfStage;   // chunk + warm-up + 16-byte alignment of both ends
}

// max_waves: upper bound of the number of non-empty chunks the grid should cover in one pass (grid-stride beyond)
hipError_t launch_hot_fill(const HotTables& h, const DevAutomaton& a, const ScanGeom& g, const uint64_t* active,
                           const uint64_t* totals, uint64_t cap, uint64_t max_waves, const uint64_t* aoff,
                           acgpu_match* out, hipStream_t s) {
    uint64_t waves = max_waves < g.n_chunks ? max_waves : g.n_chunks;
    uint64_t blocks = (waves + kHfWaves - 1) / kHfWaves;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    DfaEng eng; eng.d = a.dfa; eng.cls = a.dfa.classes;
    const size_t smem = ((size_t(h.n_hot) * (size_t(1) << a.dfa.stride2) * 2 + 15) & ~size_t(15)) + size_t(kHfWaves) * kHfStage;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_hot_fill), 160 * 1024 - 1024); e != hipSuccess) return e;
    k_hot_fill<<<dim3(uint32_t(blocks)), dim3(kHfWaves * 64), smem, s>>>(eng, h.tab, h.hid2sid, h.n_hot, h.first_match,
                                                                         h.start, g, active, totals, cap, aoff, out);
    return hipGetLastError();
}

}  // namespace acgpu
