// Host-side tables of the two prefix-filter kernels (device/pf_scan.hip, device/pfx_scan.hip): Bloom tables, the exact
// level-2 structures and the trie-only transition table of level 3.  Pure host code, so that what matters most about
// a filter -- that its tables let EVERY occurrence through -- is testable without a GPU: pf_emulate_count() below runs
// both kernels' decision logic over a haystack on the CPU (tests/test_pf_tables.py compares it with the oracle through
// acgpu_test_pf_host).
#include "pf_tables.hpp"

#include <algorithm>
#include <cstdlib>
#include <utility>

namespace acgpu {

// `order` = hid -> nnfa sid, `sid2hid` its inverse (hid_order, host/lw_tables.cpp).  false: the automaton is not served by
// the prefix filters (an empty pattern, too many patterns).
bool build_pf_host(const NNfa& n, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid, PfHostTables& t,
                   int want_tails, bool want_key8_x2, bool want_short) {
    t = PfHostTables();
    const uint32_t su = n.special.start_unanchored_id, sa = n.special.start_anchored_id;
    const size_t nh = order.size();
    // ---- prefix-filter tables (pf_scan.hip): only without empty patterns, and while the 64 KiB Bloom table stays
    // selective (two entries per pattern in 512 Ki bits: <= 6 % fill)
    if (n.min_pattern_len == 0 || n.pattern_lens.empty() || n.n_patterns > kPfMaxPatterns) return false;
    auto is_trie_child = [&](uint32_t parent, uint32_t k) {  // transition k of `parent` is a trie edge
        const uint32_t t = n.tnext[k];
        return t != kFail && t != kDead && t != su && t != sa && (parent != su || t != su);
    };
    std::vector<uint32_t> own(nh, 0), own_pid(nh, 0);   // own_pid: the lowest id among them (start_select.hip)
    for (size_t h = 1; h < nh; h++) {
        const uint32_t s = order[h];
        if (s == su) continue;
        const uint32_t dist = n.depth[s] + 1;
        for (uint32_t k = n.moff[s]; k < n.moff[s + 1]; k++)
            if (n.pattern_lens[n.mpid[k]] == dist) {
                if (own[h] == 0 || n.mpid[k] < own_pid[h]) own_pid[h] = n.mpid[k];
                own[h]++;
            }
    }
    // ---- short mode (pf_tables.hpp): one or two distinct stragglers of 3..8 bytes beside patterns of >= 9 bytes
    struct ShortPat { uint32_t hid, len; uint8_t b[8]; };
    std::vector<ShortPat> shorts;
    size_t min_long = SIZE_MAX;
    bool short_ok = want_short && want_key8_x2 && n.n_patterns >= 256 && n.min_pattern_len < 9 && n.min_pattern_len >= 3;
    if (short_ok) {
        struct FrameS { uint32_t sid, d; uint8_t b[8]; };
        std::vector<FrameS> st{{su, 0, {0, 0, 0, 0, 0, 0, 0, 0}}};
        while (!st.empty() && short_ok) {
            const FrameS f = st.back(); st.pop_back();
            const uint32_t h = sid2hid[f.sid];
            if (f.d > 0 && own[h]) {
                ShortPat sp{h, f.d, {0, 0, 0, 0, 0, 0, 0, 0}};
                std::copy(f.b, f.b + 8, sp.b);
                shorts.push_back(sp);
                if (shorts.size() > kPfxShortMax) short_ok = false;
            }
            if (f.d == 8) continue;
            for (uint32_t k = n.toff[f.sid]; k < n.toff[f.sid + 1]; k++)
                if (is_trie_child(f.sid, k)) { FrameS c = f; c.sid = n.tnext[k]; c.b[f.d] = n.tbyte[k]; c.d = f.d + 1; st.push_back(c); }
        }
        for (size_t h = 1; h < nh && short_ok; h++)
            if (own[h] && n.depth[order[h]] + 1 >= 9) min_long = std::min<size_t>(min_long, n.depth[order[h]] + 1);
        short_ok = short_ok && !shorts.empty() && min_long != SIZE_MAX;
    }
    const size_t eff_min_len = short_ok ? min_long : n.min_pattern_len;
    // patterns ending in `hd` that the long-key tables report: in short mode none at the prefix depth (eight bytes = a straggler)
    auto own_long8 = [&](uint32_t hd) { return short_ok ? 0u : own[hd]; };
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
    // Case folding of the KEYS (HotTables::pf_fold): under ascii_case_insensitive every letter edge exists twice and leads
    // to one child, so a 3-byte key has up to 8 spellings and the tables fill up accordingly (config 5: the same kernel
    // takes 2.2 ms where the case-sensitive twin takes 1.57).  When the start state's edges look like that -- at least 1.2
    // edges per distinct child (printable-ASCII sets: 95 edges, 69 children) -- keys are inserted, and looked up by the kernel, with 0x20 or-ed into every byte: one
    // spelling per key.  The bit selectors use the low five bits of a byte, which the fold leaves alone; level 3 reads the
    // haystack itself.  Folding is exact for any automaton (the same function on both sides of every comparison); it only
    // pays when spellings collapse.
    bool fold = false;
    {
        uint32_t edges = 0;
        std::vector<uint32_t> kids;
        for (uint32_t k = n.toff[su]; k < n.toff[su + 1]; k++) if (is_trie_child(su, k)) { edges++; kids.push_back(n.tnext[k]); }
        std::sort(kids.begin(), kids.end());
        kids.erase(std::unique(kids.begin(), kids.end()), kids.end());
        fold = !kids.empty() && 5 * edges >= 6 * kids.size();   // (case-sensitive tries: exactly one edge per child)
    }
    const uint32_t fm = fold ? 0x202020u : 0u;
    const uint32_t bits_bytes = 64 * 1024;
    // Large sets (HotTables::pf_exact2): the second table holds one entry per pattern keyed by its true start instead
    // (filled after this loop), so here only the first table is written.
    const bool exact2 = n.n_patterns > kPfExact2Patterns;
    t.exact2 = exact2;
    std::vector<uint32_t> bits(bits_bytes / 4, 0), bits2(kPfBits2Bytes / 4, 0);
    uint32_t sink = 0;
    struct TwoWords {   // the word of the key in both tables
        uint32_t &w1, &w2;
        void operator=(uint32_t v) { w1 = v; w2 = v; }
        void operator|=(uint32_t v) { w1 |= v; w2 |= v; }
    };
    auto word_of = [&](uint32_t b0, uint32_t b1, uint32_t b2) -> TwoWords {
        const uint32_t key = (b0 | (b1 << 8) | (b2 << 16)) | fm;
        return TwoWords{bits[(pf_hash(key) & (bits_bytes - 1)) >> 2],
                        exact2 ? sink : bits2[(pf_hash2(key) & (kPfBits2Bytes - 1)) >> 2]};
    };
    auto bit_of = [](uint32_t b) { return 1u << (31 - (b & 31)); };
    // third table (HBM / L2): exact first four bytes of every pattern, ~64 bits per pattern
    const bool use_x = eff_min_len >= 4 && n.n_patterns >= 256;   // pfx_scan.hip tables (short mode: the 4-byte ones miss stragglers of three bytes)
    const bool use3 = (n.n_patterns >= kPfBits3Patterns && n.min_pattern_len >= 3) || use_x;
    std::vector<uint32_t> xbits(use_x ? kPfxBitsBytes / 4 : 0, 0);
    std::vector<std::pair<uint32_t, uint32_t>> xkeys;   // (first four bytes, depth-4 node | own flag)
    uint32_t log3 = 20;
    while (use3 && log3 < 28 && (uint64_t(1) << log3) < uint64_t(n.n_patterns) * 64) log3++;
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
            return bits2[(pf_hash2((b0 | (b1 << 8) | (b2 << 16)) | fm) & (kPfBits2Bytes - 1)) >> 2];
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
    t.use3 = use3;
    t.fold = fold;
    t.bits3_log2 = use3 ? log3 : 0;
    t.bits_bytes = bits_bytes;
    t.ashift = ashift;
    t.n_patterns = uint32_t(n.n_patterns);
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
        t.pfx_map.swap(map);
        t.pfx_map_log2 = lg;
        t.pfx_prefixes = uint32_t(xkeys.size());
        // the long-prefix map: every trie path of length `depth` from the start state (depth <= shortest pattern, so every
        // pattern passes through exactly one of them)
        const uint32_t depth = uint32_t(std::min<size_t>(8, eff_min_len));
        if (depth > 4) {
            struct Path { uint32_t lo, hi, node; };
            std::vector<Path> paths;
            struct Frame { uint32_t sid, d; uint64_t key; };
            std::vector<Frame> stack{{su, 0, 0}};
            while (!stack.empty()) {
                const Frame f = stack.back(); stack.pop_back();
                if (f.d == depth) {
                    const uint32_t hd = sid2hid[f.sid];
                    if (short_ok && n.toff[f.sid] == n.toff[f.sid + 1]) continue;   // (an 8-byte straggler nothing longer begins with)
                    paths.push_back({uint32_t(f.key), uint32_t(f.key >> 32), hd | (own_long8(hd) ? 0x80000000u : 0u)});
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
            // tail records (hot.hpp): every pattern end at or below the node lies within kPfxTailMaxLen bytes of it and there are
            // at most kPfxTailMaxRecs of them (want_tails == 1: exactly one, not at the node itself -- the chain tails of round 4)
            const bool no_tails = want_tails == 0;
            std::vector<uint32_t> tails;
            auto tail_of = [&](uint32_t hd) -> uint32_t {   // index + 1 of the node's first record (| kPfxTailMulti), 0 = none
                if (no_tails || tails.size() / kPfxTailWords + kPfxTailMaxRecs >= (size_t(1) << 20)) return 0;   // (20-bit index in the hit entries)
                struct Rec { uint32_t bytes[4]; uint32_t node, len, cnt; };
                struct Frame2 { uint32_t sid, len; uint32_t bytes[4]; };
                std::vector<Rec> recs;
                std::vector<Frame2> st{{order[hd], 0, {0, 0, 0, 0}}};
                while (!st.empty()) {
                    const Frame2 f = st.back(); st.pop_back();
                    const uint32_t h = sid2hid[f.sid];
                    if (own[h] && !(short_ok && f.len == 0)) {   // (short mode: a pattern ending AT the prefix node is a straggler's)
                        if (own[h] > 0xFFFFFFu || recs.size() == kPfxTailMaxRecs) return 0;
                        recs.push_back({{f.bytes[0], f.bytes[1], f.bytes[2], f.bytes[3]}, h, f.len, own[h]});
                    }
                    for (uint32_t k = n.toff[f.sid]; k < n.toff[f.sid + 1]; k++) {
                        if (!is_trie_child(f.sid, k)) continue;
                        if (f.len == kPfxTailMaxLen) return 0;   // a pattern ends further down than a record holds
                        Frame2 c = f;
                        c.sid = n.tnext[k]; c.bytes[f.len >> 2] |= uint32_t(n.tbyte[k]) << (8 * (f.len & 3)); c.len = f.len + 1;
                        st.push_back(c);
                    }
                }
                if (recs.empty()) return 0;
                if (want_tails == 1 && (recs.size() != 1 || recs[0].len == 0)) return 0;
                std::stable_sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) { return x.len < y.len; });
                const uint32_t first = uint32_t(tails.size() / kPfxTailWords) + 1;
                for (size_t i = 0; i < recs.size(); i++) {
                    const Rec& r = recs[i];
                    // (word 6: where the walk would start, for the last bytes of a span -- the node with its own-pattern flag)
                    const uint32_t rec[kPfxTailWords] = {r.bytes[0], r.bytes[1], r.bytes[2], r.bytes[3], r.node, r.len | (r.cnt << 8),
                                                         hd | (own_long8(hd) ? 0x80000000u : 0u), uint32_t(recs.size() - 1 - i)};
                    tails.insert(tails.end(), rec, rec + kPfxTailWords);
                }
                return first | (recs.size() > 1 ? kPfxTailMulti : 0u);
            };
            for (const Path& pt : paths) {
                for (uint32_t b = pfx_map8_bucket(pt.lo, pt.hi, lg8);; b = (b + 1) & (nb8 - 1)) {
                    uint32_t* q = &map8[size_t(b) * 4];
                    if ((q[2] & ~kPfxMapOverflow) == 0) {
                        q[0] = pt.lo; q[1] = pt.hi; q[2] |= pt.node;
                        q[3] = tail_of(pt.node & 0x3FFFFFFFu);
                        break;
                    }
                    q[2] |= kPfxMapOverflow;
                }
            }
            t.pfx_tail_nodes = uint32_t(tails.size() / kPfxTailWords);
            t.pfx_tails.swap(tails);
            t.pfx_map8.swap(map8);
            t.pfx_map8_log2 = lg8;
            t.pfx_depth = depth;
            // level 1 on the whole long prefix (5..8 bytes; a prefix shorter than eight enters with its bytes 4.. zero-padded and
            // the kernel masks its window the same way: one VALU operation per position).  Round 4 built this table for 8-byte
            // prefixes only: ONE 7-byte word in a dictionary sent natural text back to the 4-byte key (7 % survivors, 2.4x the time)
            if (depth >= 5 && paths.size() <= kPfxKey8MaxPrefixes) {
                std::vector<uint32_t> xbits8(kPfxBitsBytes / 4, 0);
                for (const Path& pt : paths) { const uint32_t h8 = pfx_hash8(pt.lo, pt.hi); xbits8[pfx_word(h8)] |= pfx_mask(h8); }
                t.xbits8.swap(xbits8);
                const bool no_x2 = !want_key8_x2;
                if (eff_min_len >= 9 && !no_x2) {   // every 9-byte trie path, as type 0 and as type 1 (hot.hpp)
                    std::vector<uint32_t> x2(kPfxBitsBytes / 4, 0);
                    struct F9 { uint32_t sid, d; uint8_t b[9]; };
                    std::vector<F9> st9{{su, 0, {0, 0, 0, 0, 0, 0, 0, 0, 0}}};
                    auto le32 = [](const uint8_t* q) { return uint32_t(q[0]) | (uint32_t(q[1]) << 8) | (uint32_t(q[2]) << 16) | (uint32_t(q[3]) << 24); };
                    while (!st9.empty()) {
                        const F9 f = st9.back(); st9.pop_back();
                        if (f.d == 9) {
                            const uint32_t h0 = pfx_hash8(le32(f.b + 1), le32(f.b + 5)), h1 = pfx_hash8(le32(f.b), le32(f.b + 4));
                            x2[pfx_word(h0)] |= pfx_x2_mask(h0, f.b[0], 0);
                            x2[pfx_word(h1)] |= pfx_x2_mask(h1, f.b[8], 1);
                            continue;
                        }
                        for (uint32_t k = n.toff[f.sid]; k < n.toff[f.sid + 1]; k++)
                            if (is_trie_child(f.sid, k)) { F9 c = f; c.sid = n.tnext[k]; c.b[f.d] = n.tbyte[k]; c.d = f.d + 1; st9.push_back(c); }
                    }
                    t.xbits8x2.swap(x2);
                }
            }
        }
        t.pfx_ok = true;
    }
    if (short_ok) {
        // (without the every-other-position table -- too many prefixes -- no kernel serves the short mode: the set as a whole, then)
        if (t.xbits8x2.empty()) return build_pf_host(n, order, sid2hid, t, want_tails, want_key8_x2, false);
        t.short_n = uint32_t(shorts.size());
        for (size_t i = 0; i < shorts.size(); i++) {
            const ShortPat& sp = shorts[i];
            t.short_lo[i] = uint32_t(sp.b[0]) | (uint32_t(sp.b[1]) << 8) | (uint32_t(sp.b[2]) << 16) | (uint32_t(sp.b[3]) << 24);
            t.short_hi[i] = uint32_t(sp.b[4]) | (uint32_t(sp.b[5]) << 8) | (uint32_t(sp.b[6]) << 16) | (uint32_t(sp.b[7]) << 24);
            t.short_len[i] = sp.len; t.short_node[i] = sp.hid;
        }
        t.pfx4_complete = n.min_pattern_len >= 4;
    }
    t.own.swap(own); t.own_pid.swap(own_pid); t.atab.swap(atab); t.acls.swap(acls);
    t.bits.swap(bits); t.bits2.swap(bits2); t.bits3.swap(bits3); t.xbits.swap(xbits);
    t.ok = true;
    return true;
}


// ---- CPU model of the filters' decisions (test hook).  It follows the kernels level by level -- the same hashes, tables and
// the same notion of which start positions a probe stands for -- but not their scheduling: what it establishes is that the
// TABLES admit every occurrence (no false negatives) and that level 3 counts each occurrence once with the right
// multiplicity.  Bytes past the end of the haystack read as zero (the kernels see the bytes of the 16-byte hull there;
// no pattern that ends inside the span depends on them).
namespace {

struct PfModel {
    const PfHostTables& t;
    const uint8_t* hay;
    size_t len;
    uint32_t byte(size_t i) const { return i < len ? hay[i] : 0u; }
    // level 3 from trie node `s` (hid) reached with the bytes before `at`; returns the occurrences it finds
    uint64_t walk(uint32_t s, size_t at) const {
        uint64_t found = 0;
        for (; at < len; at++) {
            const uint32_t e = t.atab[(size_t(s) << t.ashift) | t.acls[hay[at]]];
            if (e == 0) break;
            s = e & 0x7FFFFFFFu;
            if (e >> 31) found += t.own[s];
        }
        return found;
    }
};

}  // namespace

uint64_t pf_emulate_count(const PfHostTables& t, uint32_t start_hid, const uint8_t* hay, size_t len, int kernel, uint64_t* info) {
    const PfModel m{t, hay, len};
    uint64_t total = 0, survivors1 = 0, survivors2 = 0, survivors_gate = 0, tail_hits = 0;
    if (kernel == 0) {
        // two-type filter: probes at the odd offsets q of 16-byte rows, i.e. at every odd q relative to the row origin; the
        // kernel's rows start at a 16-byte boundary of the virtual origin, so relative to the haystack start the probed
        // positions are the odd ones when the haystack is 16-byte aligned -- which this model assumes (offset 0 = row
        // start).  Start 0 has no probe to its left: the kernel hands it to level 3 directly.
        auto level3 = [&](size_t v) {   // the kernel's level 3 for start v: bits3 gate, then the trie walk from the start state
            if (v >= len) return;
            if (t.use3 && v + 4 <= len) {
                const uint32_t k4 = m.byte(v) | (m.byte(v + 1) << 8) | (m.byte(v + 2) << 16) | (m.byte(v + 3) << 24);
                const uint32_t h = pf_hash3(k4, t.bits3_log2);
                if (!((t.bits3[h >> 5] >> (h & 31)) & 1u)) return;
            }
            total += m.walk(start_hid, v);
        };
        level3(0);
        for (size_t q = 1; q < len + 1; q += 2) {
            const uint32_t fm = t.fold ? 0x202020u : 0u;   // (the kernel folds its row registers: keys only, selectors keep their low 5 bits)
            const uint32_t key = (m.byte(q + 1) | (m.byte(q + 2) << 8) | (m.byte(q + 3) << 16)) | fm;
            const uint32_t w = t.bits[(pf_hash(key) & (t.bits_bytes - 1)) >> 2];
            const bool l1 = ((w << (m.byte(q) & 31)) | (w << (m.byte(q + 4) & 31))) >> 31;
            if (!l1) continue;
            survivors1++;
            bool ok_a, ok_b;
            if (t.exact2) {   // one entry per trie path keyed by the true start: q -> key b[q..q+2], bit b[q+3]; q+1 likewise
                const uint32_t ka = (m.byte(q) | (m.byte(q + 1) << 8) | (m.byte(q + 2) << 16)) | fm;
                const uint32_t wa = t.bits2[(pf_hash2(ka) & (kPfBits2Bytes - 1)) >> 2];
                const uint32_t wb = t.bits2[(pf_hash2(key) & (kPfBits2Bytes - 1)) >> 2];
                ok_a = (wa << (m.byte(q + 3) & 31)) >> 31;
                ok_b = (wb << (m.byte(q + 4) & 31)) >> 31;
            } else {
                const uint32_t w2 = t.bits2[(pf_hash2(key) & (kPfBits2Bytes - 1)) >> 2];
                ok_a = ok_b = ((w2 << (m.byte(q) & 31)) | (w2 << (m.byte(q + 4) & 31))) >> 31;
            }
            if (ok_a || ok_b) survivors2++;
            if (ok_a) level3(q);
            if (ok_b) level3(q + 1);
        }
    } else {
        if (!t.pfx_ok) return ~uint64_t(0);
        if (kernel == 3 && t.xbits8.empty()) return ~uint64_t(0);
        if (kernel == 4 && t.xbits8x2.empty()) return ~uint64_t(0);
        // short mode: the long-key tables hold the long patterns only -- the every-other-position kernel compares the
        // stragglers itself, the other long-key kernels do not run; the 4-byte kernel while no straggler is shorter than that
        if (t.short_n && (kernel == 2 || kernel == 3 || (kernel == 1 && !t.pfx4_complete))) return ~uint64_t(0);
        if (t.short_n && kernel == 4)
            for (size_t q = 0; q < len; q++)
                for (uint32_t i = 0; i < t.short_n; i++) {
                    const uint32_t sl = t.short_len[i];
                    if (q + sl > len) continue;
                    bool same = true;
                    for (uint32_t k = 0; k < sl && same; k++) same = m.byte(q + k) == (((k < 4 ? t.short_lo[i] : t.short_hi[i]) >> (8 * (k & 3))) & 0xFFu);
                    if (same) total += t.own[t.short_node[i]];
                }
        const bool long_key = !t.pfx_map8.empty() && kernel >= 2;
        const bool key8 = kernel == 3, x2 = kernel == 4;
        const uint32_t depth = long_key ? t.pfx_depth : 4;
        for (size_t q = 0; q + depth <= len; q++) {
            const uint32_t key4 = m.byte(q) | (m.byte(q + 1) << 8) | (m.byte(q + 2) << 16) | (m.byte(q + 3) << 24);
            const uint32_t hi4 = m.byte(q + 4) | (m.byte(q + 5) << 8) | (m.byte(q + 6) << 16) | (m.byte(q + 7) << 24);
            if (x2) {
                // the kernel probes the odd offsets p: a start at an odd position is tested as type 0 (key = its bytes 1..8,
                // selector = its byte 0), a start at an even position q as type 1 of the probe at q - 1 (key = its bytes
                // 0..7, selector = its byte 8); position 0 has no probe in front of it and is handed to level 2 as it is
                if (q != 0) {
                    uint32_t h, mask;
                    if (q & 1) {
                        const uint32_t lo = m.byte(q + 1) | (m.byte(q + 2) << 8) | (m.byte(q + 3) << 16) | (m.byte(q + 4) << 24);
                        const uint32_t hi = m.byte(q + 5) | (m.byte(q + 6) << 8) | (m.byte(q + 7) << 16) | (m.byte(q + 8) << 24);
                        h = pfx_hash8(lo, hi); mask = pfx_x2_mask(h, m.byte(q), 0);
                    } else {
                        h = pfx_hash8(key4, hi4); mask = pfx_x2_mask(h, m.byte(q + 8), 1);
                    }
                    if ((t.xbits8x2[pfx_word(h)] & mask) != mask) continue;
                }
            } else {
                const uint32_t himask = depth >= 8 ? 0xFFFFFFFFu : (1u << (8 * (depth - 4))) - 1u;
                const uint32_t h = key8 ? pfx_hash8(key4, hi4 & himask) : pfx_hash(key4);
                const uint32_t mask = pfx_mask(h);
                if (((key8 ? t.xbits8 : t.xbits)[pfx_word(h)] & mask) != mask) continue;
            }
            survivors1++;
            uint32_t node = 0, tail = 0;
            if (long_key) {
                uint32_t khi = 0;
                for (uint32_t i = 4; i < depth; i++) khi |= m.byte(q + i) << (8 * (i - 4));
                const uint32_t nb = 1u << t.pfx_map8_log2;
                for (uint32_t b = pfx_map8_bucket(key4, khi, t.pfx_map8_log2);; b = (b + 1) & (nb - 1)) {
                    const uint32_t* e = &t.pfx_map8[size_t(b) * 4];
                    const uint32_t val = e[2] & ~kPfxMapOverflow;
                    if (val && e[0] == key4 && e[1] == khi) { node = val; tail = e[3]; break; }
                    if (!(e[2] & kPfxMapOverflow)) break;
                }
            } else {
                if (t.use3) {   // the exact-prefix bit table in front of the map (k_pfx_count's gate)
                    const uint32_t h3 = pf_hash3(key4, t.bits3_log2);
                    if (!((t.bits3[h3 >> 5] >> (h3 & 31)) & 1u)) continue;
                    survivors_gate++;
                }
                const uint32_t nb = 1u << t.pfx_map_log2;
                for (uint32_t b = pfx_map_bucket(key4, t.pfx_map_log2);; b = (b + 1) & (nb - 1)) {
                    const uint32_t* e = &t.pfx_map[size_t(b) * 4];
                    if (e[1] && e[0] == key4) { node = e[1] & ~kPfxMapOverflow; break; }
                    if (e[3] && e[2] == key4) { node = e[3]; break; }
                    if (!(e[1] & kPfxMapOverflow)) break;
                }
            }
            if (!node) continue;
            survivors2++;
            if (tail && q + depth + kPfxTailMaxLen <= len) {   // the kernel's tail path: masked compares of the 16 bytes behind the prefix
                const uint32_t* rec = &t.pfx_tails[size_t((tail & ~kPfxTailMulti) - 1) * kPfxTailWords];
                for (;; rec += kPfxTailWords) {
                    const uint32_t tl = rec[5] & 0xFFu;
                    bool same = true;
                    for (uint32_t i = 0; i < tl && same; i++) same = m.byte(q + depth + i) == ((rec[i >> 2] >> (8 * (i & 3))) & 0xFFu);
                    if (same) total += rec[5] >> 8;
                    if (rec[7] == 0) break;
                }
                tail_hits++;
                continue;
            }
            const uint32_t s = node & 0x7FFFFFFFu;
            if (node >> 31) total += t.own[s];
            total += m.walk(s, q + depth);
        }
    }
    if (info) { info[0] = survivors1; info[1] = survivors2; info[2] = survivors_gate; info[3] = tail_hits; }
    return total;
}

}  // namespace acgpu
