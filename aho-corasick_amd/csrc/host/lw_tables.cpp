// Host-side tables of the LDS walk engine (device/lds_walk.hip): either one row per state (kLwFull), or dense rows +
// single-exception handles + exception chains (kLwNarrow / kLwWide), laid out as the LDS image the kernel copies in.
// Pure host code (no HIP), so the tables and the step rules are testable without a GPU: lw_emulate_count() below walks
// a haystack with exactly the kernel's fast-step / flag / inline-count / exact-redo logic (tests/test_lw_tables.py
// compares it with the oracle through acgpu_test_lw_host).
#include "lw_tables.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace acgpu {

// hid order: DEAD, non-match states breadth first, match states breadth first (n.bfs = [START_U, START_A, queue...];
// START_A and FAIL are unreachable from an unanchored walk).
void hid_order(const NNfa& n, std::vector<uint32_t>& order, std::vector<uint32_t>& sid2hid, uint32_t& first_match) {
    const uint32_t sa = n.special.start_anchored_id;
    order.clear();
    order.reserve(n.states());
    order.push_back(kDead);
    for (uint32_t s : n.bfs) if (s != sa && !n.is_match(s)) order.push_back(s);
    first_match = uint32_t(order.size());
    for (uint32_t s : n.bfs) if (s != sa && n.is_match(s)) order.push_back(s);
    sid2hid.assign(n.states(), 0);
    for (size_t h = 0; h < order.size(); h++) sid2hid[order[h]] = uint32_t(h);
}

namespace {

// The engine's class map.  LDS form: reference classes whose columns are identical over every state are merged (coarser
// than the reference's ByteClasses, src/util/alphabet.rs:224-250, e.g. both cases of a letter under
// ascii_case_insensitive), which keeps one trie edge = one exception.  Computed form: the reference's classes are
// monotone step functions of the byte (alphabet.rs:235-250), and when every class but the "other" one holds a single
// byte the map is a clamp onto [lo, hi]: 0 = below, b - lo + 1 inside, n + 1 = above.
struct ClassMap {
    uint32_t ncls = 0;
    uint8_t of_byte[256] = {0};
    std::vector<uint8_t> rep;   // rep[c] = a byte of class c
    bool computed = false;
    int lo = 0, n = 0;          // computed: class = clamp(b - lo + 1, 0, n + 1)
};

bool merged_classes(const Dfa& d, const std::vector<uint32_t>& order, ClassMap& out) {
    const size_t nh = order.size(), alen = d.alphabet_len;
    auto drow = [&](uint32_t sid) { return &d.trans[size_t(sid) << d.stride2]; };
    std::vector<uint32_t> cls_of_dclass(alen, 0), rep;   // rep[c] = a reference class of engine class c
    std::vector<uint64_t> sig(alen, 0xCBF29CE484222325ull);
    for (size_t h = 1; h < nh; h++) {
        const uint32_t* row = drow(order[h]);
        for (size_t c = 0; c < alen; c++) sig[c] = (sig[c] ^ row[c]) * 0x100000001B3ull + (sig[c] >> 29);
    }
    std::unordered_map<uint64_t, std::vector<uint32_t>> by_sig;   // signature -> engine classes with it
    for (size_t c = 0; c < alen; c++) {
        uint32_t found = UINT32_MAX;
        for (uint32_t cand : by_sig[sig[c]]) {
            bool same = true;
            for (size_t h = 1; h < nh && same; h++) { const uint32_t* row = drow(order[h]); same = row[c] == row[rep[cand]]; }
            if (same) { found = cand; break; }
        }
        if (found == UINT32_MAX) { found = uint32_t(rep.size()); rep.push_back(uint32_t(c)); by_sig[sig[c]].push_back(found); }
        cls_of_dclass[c] = found;
    }
    if (rep.size() > 256) return false;
    out = ClassMap();
    out.ncls = uint32_t(rep.size());
    out.rep.assign(out.ncls, 0);
    std::vector<uint8_t> seen(out.ncls, 0);
    for (int b = 0; b < 256; b++) {
        const uint32_t c = cls_of_dclass[d.byte_classes[b]];
        out.of_byte[b] = uint8_t(c);
        if (!seen[c]) { seen[c] = 1; out.rep[c] = uint8_t(b); }
    }
    return true;
}

// The clamp form of `m`, if it has one: "other" = the class with the most bytes; every other class one byte.
bool computed_classes(const ClassMap& m, ClassMap& out) {
    uint32_t bytes_of[256] = {0};
    for (int b = 0; b < 256; b++) bytes_of[m.of_byte[b]]++;
    uint32_t other = 0;
    for (uint32_t c = 1; c < m.ncls; c++) if (bytes_of[c] > bytes_of[other]) other = c;
    int lo = -1, hi = -1;
    for (int b = 0; b < 256; b++) {
        if (m.of_byte[b] == other) continue;
        if (bytes_of[m.of_byte[b]] != 1) return false;   // a merged class (two cases of a letter): one edge would be two exceptions
        if (lo < 0) lo = b;
        hi = b;
    }
    if (lo < 0) return false;
    const int n = hi - lo + 1;
    if (n + 2 > 256) return false;
    out = ClassMap();
    out.computed = true;
    out.lo = lo; out.n = n;
    out.ncls = uint32_t(n + 2);
    out.rep.assign(out.ncls, m.rep[other]);
    for (int b = 0; b < 256; b++) {
        const int c = b < lo ? 0 : b > hi ? n + 1 : b - lo + 1;
        out.of_byte[b] = uint8_t(c);
        if (c >= 1 && c <= n) out.rep[size_t(c)] = uint8_t(b);
    }
    return true;
}

struct Build {
    const NNfa& n;
    const Dfa& d;
    const std::vector<uint32_t>& order;
    const std::vector<uint32_t>& sid2hid;
    uint32_t first_match;
    size_t nh;
    std::vector<uint16_t> mlen;   // [nh - first_match]

    bool match_lens() {
        mlen.assign(nh - first_match, 0);
        for (size_t h = first_match; h < nh; h++) {
            const uint32_t o = order[h] - 2;   // DFA match-state index: (sid >> stride2) - 2 with sid = nnfa id << stride2 (dfa.rs:553-555)
            const uint32_t len = d.moff[o + 1] - d.moff[o];
            if (len > 0xFFFFu) return false;
            mlen[h - first_match] = uint16_t(len);
        }
        return true;
    }
    // class-compressed transition table over hids under class map m; false if a transition dies
    bool deltas(const ClassMap& m, std::vector<uint32_t>& dl) const {
        dl.assign(nh * m.ncls, 0);
        for (size_t h = 1; h < nh; h++) {
            const uint32_t* row = &d.trans[size_t(order[h]) << d.stride2];
            for (uint32_t c = 0; c < m.ncls; c++) {
                const uint32_t t = sid2hid[row[d.byte_classes[m.rep[c]]] >> d.stride2];
                if (t == 0) return false;   // an unanchored Standard DFA never dies; the engine relies on it
                dl[h * m.ncls + c] = t;
            }
        }
        return true;
    }
    void class_fields(const ClassMap& m, uint32_t rows_k, LwHostTables& out) const {
        out.classes = m.ncls;
        out.rows_k = rows_k;
        out.computed_cls = m.computed;
        if (m.computed) { out.cc_add = int32_t(rows_k) + 1 - m.lo; out.cc_lo = int32_t(rows_k); out.cc_hi = int32_t(rows_k) + m.n + 1; }
        uint16_t* cls = reinterpret_cast<uint16_t*>(out.image.data());
        for (int b = 0; b < 256; b++) cls[b] = uint16_t(rows_k + m.of_byte[b]);
    }

    // ---- kLwFull: one row per state (+ one column for the offset of its match list), all rows within 64 KiB
    bool full(const ClassMap& m, LwHostTables& out) const {
        const uint32_t row_dw = (m.ncls + 1) | 1u;
        const uint64_t row_bytes_total = uint64_t(nh - 1) * row_dw * 4;
        if (row_bytes_total > 0x10000u) return false;   // the row's byte address is the high half of the handle
        for (uint16_t len : mlen) if (len > 4095) return false;   // four handles are summed before the count is taken out
        const uint64_t bytes = row_bytes_total + kLwClsBytes;
        std::vector<uint32_t> dl;
        if (!deltas(m, dl)) return false;
        // match lists: {pattern id, pattern length} per entry
        uint64_t n_list = 0;
        for (size_t h = first_match; h < nh; h++) n_list += mlen[h - first_match];
        const bool with_lists = bytes + 8 * n_list + 16 <= kLwLdsBudget;
        out = LwHostTables();
        out.flavour = kLwFull;
        out.image.assign(size_t((bytes + (with_lists ? 8 * n_list : 0) + 15) & ~uint64_t(15)) / 4, 0);
        // sync states (kLwFullSync) and the monotone property, from the trie: breadth-first, so a state's failure target is done
        std::vector<uint8_t> sync(nh, 0);
        bool monotone = true;
        {
            const uint32_t su = n.special.start_unanchored_id;
            for (uint32_t s : n.bfs) {
                if (s == n.special.start_anchored_id) continue;
                if (s == su) { sync[sid2hid[s]] = 1; continue; }
                const bool leaf = n.toff[s + 1] == n.toff[s];
                sync[sid2hid[s]] = leaf && sync[sid2hid[n.fail[s]]];
                if (!leaf && n.is_match(s)) monotone = false;
            }
            if (n.is_match(su)) monotone = false;   // (an empty pattern)
        }
        out.monotone = monotone;
        // no two occurrences can ever overlap, nor end at one place: every match state is a sync state with one pattern
        out.disjoint = n.min_pattern_len >= 1;
        for (size_t h = first_match; h < nh; h++) if (!sync[h] || mlen[h - first_match] != 1) out.disjoint = false;
        auto H = [&](uint32_t h) {
            return (((h - 1) * row_dw * 4) << 16) | (h >= first_match ? uint32_t(mlen[h - first_match]) : 0u) | (sync[h] ? kLwFullSync : 0u);
        };
        uint32_t* rows = out.image.data() + kLwClsBytes / 4;
        for (size_t h = 1; h < nh; h++)
            for (uint32_t c = 0; c < m.ncls; c++) rows[(h - 1) * row_dw + c] = H(dl[h * m.ncls + c]);
        if (with_lists) {
            uint32_t at = uint32_t(row_bytes_total);   // table-relative byte offset
            out.mlist_off = at;
            for (size_t h = first_match; h < nh; h++) {
                const uint32_t o = order[h] - 2;   // DFA match-state index (dfa.rs:553-555)
                rows[(h - 1) * row_dw + m.ncls] = at;
                for (uint32_t i = d.moff[o]; i < d.moff[o + 1]; i++) {
                    const uint32_t pid = d.mpid[i];
                    out.image[(kLwClsBytes + at) / 4] = pid;
                    out.image[(kLwClsBytes + at) / 4 + 1] = n.pattern_lens[pid];
                    at += 8;
                }
            }
        }
        // class values premultiplied by four (byte offsets into a row): the u16 map, or the clamp in that domain
        out.classes = m.ncls;
        out.rows_k = 0;
        out.computed_cls = m.computed;
        if (m.computed) { out.cc_add = 4 * (1 - m.lo); out.cc_lo = 0; out.cc_hi = 4 * (m.n + 1); }
        uint16_t* cls = reinterpret_cast<uint16_t*>(out.image.data());
        for (int b = 0; b < 256; b++) cls[b] = uint16_t(4 * m.of_byte[b]);
        out.row_bytes = 4 * row_dw;
        out.start = H(sid2hid[n.special.start_unanchored_id]);
        out.n_dense = uint32_t(nh - 1);
        out.first_match = first_match;
        out.n_states = uint32_t(nh);
        out.n_idx = uint32_t(nh);
        out.ok = true;
        return true;
    }

    // ---- kLwNarrow / kLwWide: dense rows + exceptions
    bool sparse(const ClassMap& m, int force_flavour, LwHostTables& out) const {
        const uint32_t ncls = m.ncls;
        std::vector<uint32_t> dl;
        if (!deltas(m, dl)) return false;
        auto delta = [&](uint32_t h, uint32_t c) -> uint32_t { return dl[size_t(h) * ncls + c]; };
        const uint32_t su = n.special.start_unanchored_id, sa = n.special.start_anchored_id;
        // Row stride: ncls entries rounded up to an ODD number of dwords (a-z: 27, printable ASCII: 97), not to a power of two:
        // with a stride of 128 dwords the bank of a lookup is (class mod 32) whatever the row, and the gathers of the walk pay
        // for the hot banks (profiles/r03_hot_pmc.json); with an odd stride the row index spreads the same classes over all banks.
        const uint32_t row_dw = ncls | 1u, row_bytes = 4u * row_dw;
        if (nh + 2 > 16384) return false;   // da = 4 * idx must stay below 64 KiB

        // ---- dense-state selection, in breadth-first (fail-closed) order; retried with fewer rows if LDS overflows
        struct St { uint32_t D = 0; int32_t row = -1; std::vector<uint32_t> diff; };
        std::vector<St> st(nh);
        std::vector<uint32_t> bfs_h;   // hids in breadth-first order, start state first
        for (uint32_t s : n.bfs) if (s != sa) bfs_h.push_back(sid2hid[s]);
        const uint32_t h_start = sid2hid[su];
        const uint32_t deep_est = (4 * uint32_t(nh + 1) + 1023) & ~1023u;
        if (deep_est + 2 * row_bytes + kLwClsBytes + 64 > kLwLdsBudget) return false;
        const uint32_t lds_rows = (kLwLdsBudget - kLwClsBytes - 64 - deep_est) / row_bytes - 1;
        uint32_t want_rows = 0;   // states that would get a row if rows were free
        {
            std::vector<uint8_t> has_row(nh, 0);
            std::vector<uint32_t> Dn(nh, 0);
            for (uint32_t h : bfs_h) {
                if (h == h_start) { has_row[h] = 1; Dn[h] = h; want_rows++; continue; }
                const uint32_t f = sid2hid[n.fail[order[h]]];
                Dn[h] = has_row[f] ? f : Dn[f];
                uint32_t diffs = 0;
                for (uint32_t c = 0; c < ncls && diffs < 2; c++) diffs += delta(h, c) != delta(Dn[h], c);
                if (diffs >= 2) { has_row[h] = 1; Dn[h] = h; want_rows++; }
            }
        }
        // Two layouts: base 8 | e 8 | da 16 bits, and -- for alphabets of at most 64 classes, whose rows are short enough that
        // LDS holds more than 254 of them (a-z sets) -- 10 | 6 | 16.  The wide layout costs the fast step two more VALU
        // operations (no SDWA byte selects), so it is chosen only when the narrow one would turn row states into chains.
        bool wide = ncls <= 64 && want_rows > 254 && lds_rows > 254;
        if (force_flavour == kLwWide) { if (ncls > 64) return false; wide = true; }
        if (force_flavour == kLwNarrow) wide = false;
        const uint32_t base_shift = wide ? 22 : 24, e_mask = wide ? 0x3Fu : 0xFFu;
        auto mk = [&](uint32_t base, uint32_t e, uint32_t idx) { return (base << base_shift) | (e << 16) | (idx << 2); };
        uint32_t max_rows = std::min<uint32_t>(wide ? 1022 : 254, lds_rows);
        uint32_t n_dense = 0, n_virtual = 0, rows_off = 0;
        for (int attempt = 0; attempt < 16; attempt++) {
            n_dense = 0; n_virtual = 0;
            for (auto& x : st) { x.row = -1; x.diff.clear(); x.D = 0; }
            for (uint32_t h : bfs_h) {
                St& x = st[h];
                if (h == h_start) { x.row = int32_t(n_dense++); x.D = h; continue; }
                const uint32_t f = sid2hid[n.fail[order[h]]];
                x.D = st[f].row >= 0 ? f : st[f].D;
                for (uint32_t c = 0; c < ncls; c++) if (delta(h, c) != delta(x.D, c)) x.diff.push_back(c);
                if (x.diff.size() >= 2) {
                    if (n_dense < max_rows) { x.row = int32_t(n_dense++); x.D = h; x.diff.clear(); }
                    else n_virtual += uint32_t(x.diff.size());
                }
            }
            // deep (padded to 1 KiB) | rows + poison row | nxt (u32) + vhid (u16) per virtual slot | mlen (u16) per match state
            rows_off = (4u * uint32_t(nh + n_virtual + 1) + 1023) & ~1023u;
            const uint64_t need = uint64_t(rows_off) + uint64_t(n_dense + 1) * row_bytes + 6ull * n_virtual + 2ull * (nh - first_match) + kLwClsBytes + 64;
            if (need <= kLwLdsBudget && nh + n_virtual + 1 <= 16384) break;
            if (max_rows <= 1) return false;
            const uint64_t over = need > kLwLdsBudget ? need - kLwLdsBudget : row_bytes;
            const uint32_t drop = uint32_t(std::max<uint64_t>(1, (over + row_bytes - 1) / row_bytes));
            max_rows = max_rows > drop ? max_rows - drop : 1;
            if (attempt == 15) return false;
        }
        // every state at distance <= 1 must be dense or single-exception: they serve ~90 % of the bytes
        for (uint32_t h : bfs_h) {
            const uint32_t s = order[h];
            if ((s == su || n.depth[s] == 0) && st[h].row < 0 && st[h].diff.size() >= 2) return false;
        }
        const uint32_t n_idx = uint32_t(nh) + n_virtual + 1, poison_idx = n_idx - 1, poison_row = n_dense;
        const uint32_t poison = mk(poison_row, 0, poison_idx);

        // ---- handles.  Real states: idx = hid; a multi state's idx is the first of its virtual slots.
        std::vector<uint32_t> H(nh, poison), vslot(nh, 0);
        {
            uint32_t next_virtual = uint32_t(nh);
            for (uint32_t h : bfs_h) {
                const St& x = st[h];
                if (x.row >= 0) H[h] = mk(uint32_t(x.row), 0, h);
                else if (x.diff.size() <= 1) H[h] = mk(uint32_t(st[x.D].row), x.diff.empty() ? 0u : x.diff[0], h);
                else { vslot[h] = next_virtual; H[h] = mk(poison_row, x.diff[0], next_virtual); next_virtual += uint32_t(x.diff.size()); }
            }
        }
        const uint32_t nxt_off = rows_off + (n_dense + 1) * row_bytes;
        const uint32_t vhid_off = nxt_off + 4 * n_virtual, mlen_off = (vhid_off + 2 * n_virtual + 3) & ~3u;
        const uint32_t image_bytes = (kLwClsBytes + mlen_off + 2 * uint32_t(mlen.size()) + 15) & ~15u;
        if (image_bytes > kLwLdsBudget) return false;
        out = LwHostTables();
        out.flavour = wide ? kLwWide : kLwNarrow;
        out.image.assign(image_bytes / 4, poison);
        uint8_t* img = reinterpret_cast<uint8_t*>(out.image.data()) + kLwClsBytes;
        uint32_t* deep = reinterpret_cast<uint32_t*>(img);
        uint32_t* rows = reinterpret_cast<uint32_t*>(img + rows_off);
        uint32_t* nxt = reinterpret_cast<uint32_t*>(img + nxt_off);
        uint16_t* vhid = reinterpret_cast<uint16_t*>(img + vhid_off);
        for (uint32_t h : bfs_h) {
            const St& x = st[h];
            if (x.row >= 0)
                for (uint32_t c = 0; c < ncls; c++) rows[size_t(x.row) * row_dw + c] = H[delta(h, c)];
            if (x.row < 0 && x.diff.size() >= 2) {   // exception chain over consecutive virtual slots, the last one on D's row
                const uint32_t k = uint32_t(x.diff.size()), v0 = vslot[h];
                for (uint32_t j = 0; j < k; j++) {
                    deep[v0 + j] = H[delta(h, x.diff[j])];
                    vhid[v0 + j - nh] = uint16_t(h);
                    const bool last = j + 2 == k;
                    nxt[v0 + j - nh] = j + 1 < k ? mk(last ? uint32_t(st[x.D].row) : poison_row, x.diff[j + 1], v0 + j + 1) : poison;
                }
                deep[h] = poison;   // never addressed: no handle carries a multi state's own hid
            } else {
                deep[h] = H[delta(h, (H[h] >> 16) & e_mask)];
            }
        }
        deep[0] = poison;
        deep[poison_idx] = poison;
        std::memcpy(img + mlen_off, mlen.data(), mlen.size() * 2);
        class_fields(m, rows_off / 4, out);
        out.row_bytes = row_bytes;
        out.rows_off = rows_off;
        out.nxt_off = nxt_off; out.vhid_off = vhid_off; out.mlen_off = mlen_off;
        out.fm_addr = 4 * first_match;
        out.virt_addr = 4 * uint32_t(nh);
        out.poison_row = poison_row;
        out.start = H[h_start];
        out.n_dense = n_dense;
        out.n_multi = 0;
        for (uint32_t h : bfs_h) if (st[h].row < 0 && st[h].diff.size() >= 2) out.n_multi++;
        out.first_match = first_match;
        out.n_states = uint32_t(nh);
        out.n_idx = n_idx;
        out.ok = true;
        return true;
    }
};

}  // namespace

// `order` = hid -> nnfa sid, `sid2hid` its inverse (build_hot_tables); first_match = first match hid.
bool build_lw_host(const NNfa& n, const Dfa& d, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid,
                   uint32_t first_match, LwHostTables& out, int force_flavour, int force_cls) {
    out = LwHostTables();
    const size_t nh = order.size();
    if (nh < 2 || nh > 60000) return false;
    Build b{n, d, order, sid2hid, first_match, nh, {}};
    if (!b.match_lens()) return false;
    ClassMap lc, cc;
    if (!merged_classes(d, order, lc)) return false;
    const bool have_cc = force_cls != 0 && computed_classes(lc, cc);
    if (force_cls == 1 && !have_cc) return false;
    const bool lc_allowed = force_cls != 1;
    // one row per state when that fits 64 KiB.  Here the LDS class map comes first: this walk is three VALU operations per
    // byte with it (class address, row address, sum) and six with the clamp, its gathers meet few conflicts (small automata:
    // many lanes in the same row), and the merged classes make the rows shorter -- 4.0-4.2 TB/s against 3.6-3.8 on the
    // reference's small-set definitions (profiles/r05_full_cls_ab.jsonl)
    if (force_flavour < 0 || force_flavour == kLwFull) {
        if (lc_allowed && b.full(lc, out)) return true;
        if (have_cc && b.full(cc, out)) return true;
        if (force_flavour == kLwFull) return false;
    }
    // rows + exceptions: the clamp costs a few columns per row (the bytes of the range no pattern uses); it is taken when
    // that is a small share of the row
    const bool cc_cheap = have_cc && (force_cls == 1 || cc.ncls <= lc.ncls + 8 + lc.ncls / 8);
    // (not under the wide layout: its step is two VALU operations longer already and the VALU is what bounds the walk --
    // 1 000 a-z patterns: 1 962 GB/s with the LDS map, 1 739 computed, profiles/r05_hot_ab.jsonl)
    if (cc_cheap && b.sparse(cc, force_flavour, out) && (out.flavour != kLwWide || force_cls == 1)) return true;
    if (lc_allowed && b.sparse(lc, force_flavour, out)) return true;
    return out.ok;
}

// ---- CPU emulation of the kernel's walk over one cold-started range (test hook): 4 fast steps per dword with the
// deep-address flag, matches of exact handles counted from the LDS match-length table, exact redo of dwords that met a
// multi state; kLwFull: the literal walk, counts taken from the handles.
namespace {
struct Emu {
    const LwHostTables& t;
    const uint8_t* img;   // image + kLwClsBytes
    uint32_t rd32(uint32_t a) const { uint32_t v; std::memcpy(&v, img + a, 4); return v; }
    uint32_t rd16(uint32_t a) const { uint16_t v; std::memcpy(&v, img + a, 2); return v; }
    uint32_t cls(uint8_t b) const {   // class VALUE: class + rows_k; kLwFull: 4 * class
        if (t.computed_cls) {
            const int32_t x = (t.flavour == kLwFull ? 4 * int32_t(b) : int32_t(b)) + t.cc_add;
            return uint32_t(std::min(std::max(x, t.cc_lo), t.cc_hi));
        }
        return reinterpret_cast<const uint16_t*>(t.image.data())[b];
    }
    uint32_t base_of(uint32_t h) const { return h >> (t.wide() ? 22 : 24); }
    uint32_t e_of(uint32_t h) const { return (h >> 16) & (t.wide() ? 0x3Fu : 0xFFu); }
    uint32_t fast(uint32_t h, uint8_t byte) const {
        const uint32_t cv = cls(byte);
        if (t.flavour == kLwFull) return rd32((h >> 16) + cv);
        const uint32_t ra = base_of(h) * t.row_bytes + 4 * cv;
        return rd32(e_of(h) == (cv & 0xFFu) ? (h & 0xFFFFu) : ra);
    }
    uint32_t careful(uint32_t h, uint8_t byte) const {
        if (t.flavour == kLwFull) return fast(h, byte);
        const uint32_t cv = cls(byte), c = cv & 0xFFu;
        for (int hop = 0; hop < 4096; hop++) {
            const uint32_t da = h & 0xFFFFu;
            if (e_of(h) == c) return rd32(da);
            const uint32_t b = base_of(h);
            if (b != t.poison_row) return rd32(b * t.row_bytes + cv * 4);
            h = rd32(t.nxt_off + (da - t.virt_addr));   // multi state / chain link: idx is a virtual slot
        }
        return h;
    }
    uint32_t match_len(uint32_t h) const {
        if (t.flavour == kLwFull) return h & kLwFullLenMask;
        uint32_t da = h & 0xFFFFu;
        if (da >= t.virt_addr) da = 4 * rd16(t.vhid_off + (da - t.virt_addr) / 2);   // first slot of a multi state
        return da >= t.fm_addr ? rd16(t.mlen_off + (da - t.fm_addr) / 2) : 0u;
    }
};
}  // namespace

// Share of the dwords that take the exact path on a haystack that "looks like the patterns": 32 KiB of bytes drawn
// uniformly from the bytes that begin some pattern (the inputs the prefix filter hands over are of that kind).  The
// routing rule prices the LDS walk with it (device/hot.hpp: lw_route_cb).
double lw_estimate_redo(const LwHostTables& t) {
    if (t.flavour == kLwFull) return 0.0;
    Emu e{t, reinterpret_cast<const uint8_t*>(t.image.data()) + kLwClsBytes};
    std::vector<uint8_t> first;
    for (int b = 0; b < 256; b++) if (e.careful(t.start, uint8_t(b)) != t.start) first.push_back(uint8_t(b));
    if (first.empty()) return 0.0;
    std::vector<uint8_t> hay(32 * 1024);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto& b : hay) { x = x * 6364136223846793005ull + 1442695040888963407ull; b = first[size_t((x >> 33) % first.size())]; }
    uint64_t redo = 0;
    (void)lw_emulate_count(t, hay.data(), hay.size(), &redo);
    return double(redo) / double(hay.size() / 4);
}

// kLwFull with match lists: the records of the overlapping search over hay[0..len) (cold start at 0), produced the way
// k_lw_fill produces them -- end position by end position, each state's list in its stored order.
bool lw_emulate_records(const LwHostTables& t, const uint8_t* hay, size_t len, std::vector<acgpu_match>& out) {
    out.clear();
    if (t.flavour != kLwFull || !t.mlist_off) return false;
    Emu e{t, reinterpret_cast<const uint8_t*>(t.image.data()) + kLwClsBytes};
    auto emit = [&](uint32_t h, uint64_t end) {
        const uint32_t list = e.rd32((h >> 16) + 4 * t.classes);
        for (uint32_t i = 0; i < (h & kLwFullLenMask); i++) {
            acgpu_match m;
            m.pattern = e.rd32(list + 8 * i); m._pad = 0; m.end = end; m.start = end - e.rd32(list + 8 * i + 4);
            out.push_back(m);
        }
    };
    uint32_t h = t.start;
    emit(h, 0);
    for (size_t at = 0; at < len; at++) { h = e.fast(h, hay[at]); emit(h, at + 1); }
    return true;
}

// The event form of device/lds_emit.hip on the CPU: per lane-chunk the count walk's events -- a dword that gained a record
// becomes {dword index, records of the lane-chunk so far, row address of the state before it | owned mask << 4 | walked mask,
// the four bytes}; interior lane-chunks warm up on whole 16-byte pieces and own whole dwords, the first lane-chunks of the
// span (and the last, ragged one) walk byte by byte with masks -- then the scan of the lane-chunk counts and the emit's
// re-walk of every event, in REVERSE order of arrival (the place of a record must not depend on the order of the events).
bool lw_emulate_event_records(const LwHostTables& t, const uint8_t* hay, size_t len, uint32_t chunk, uint32_t halo, std::vector<acgpu_match>& out) {
    out.clear();
    if (t.flavour != kLwFull || !t.mlist_off || chunk < 16 || (chunk & (chunk - 1)) != 0) return false;
    Emu e{t, reinterpret_cast<const uint8_t*>(t.image.data()) + kLwClsBytes};
    struct Ev { uint32_t gd, before, state, w; };
    std::vector<Ev> events;
    const size_t n_chunks = (len + chunk - 1) / chunk;
    std::vector<uint64_t> counts(n_chunks, 0);
    const uint32_t warm = (halo + 15) & ~15u;
    auto dword_at = [&](size_t p) { uint32_t w = 0; for (int k = 0; k < 4; k++) if (p + k < len) w |= uint32_t(hay[p + k]) << (8 * k); return w; };
    for (size_t j = 0; j < n_chunks; j++) {
        const size_t lo = j * chunk, hi = std::min<size_t>(lo + chunk, len);
        const bool interior = lo >= warm && hi == lo + chunk;
        size_t w0 = interior ? lo - warm : (lo >= halo ? lo - halo : 0);
        uint32_t h = t.start, cnt = 0;
        for (size_t p = w0 & ~size_t(3); p < hi; p += 4) {
            const uint32_t h0 = h, c0 = cnt, w = dword_at(p);
            uint32_t walked = 0, owned = 0;
            for (int k = 0; k < 4; k++) {
                const size_t v = p + k;
                if (v < w0 || v >= hi) continue;
                walked |= 1u << k;
                h = e.fast(h, uint8_t(w >> (8 * k)));
                if (v >= lo) { owned |= 1u << k; cnt += h & kLwFullLenMask; }
            }
            if (cnt != c0) events.push_back({uint32_t(p >> 2), c0, (h0 & 0xFFFF0000u) | (owned << 4) | walked, w});
        }
        counts[j] = cnt;
    }
    std::vector<uint64_t> offsets(n_chunks + 1, 0);
    for (size_t j = 0; j < n_chunks; j++) offsets[j + 1] = offsets[j] + counts[j];
    out.resize(offsets[n_chunks]);
    uint32_t shift = 0;
    while ((1u << shift) < chunk / 4) shift++;
    for (size_t i = events.size(); i-- > 0;) {
        const Ev& v = events[i];
        size_t at = offsets[v.gd >> shift] + v.before;
        uint64_t end = uint64_t(v.gd) << 2;
        uint32_t h = v.state;
        for (int k = 0; k < 4; k++) {
            end++;
            if (!((v.state >> k) & 1u)) continue;
            h = e.fast(h, uint8_t(v.w >> (8 * k)));
            const uint32_t n = h & kLwFullLenMask;
            if (!n || !((v.state >> (4 + k)) & 1u)) continue;
            const uint32_t list = e.rd32((h >> 16) + 4 * t.classes);
            for (uint32_t r = 0; r < n; r++) {
                acgpu_match m;
                m.pattern = e.rd32(list + 8 * r); m._pad = 0; m.end = end; m.start = end - e.rd32(list + 8 * r + 4);
                out[at++] = m;
            }
        }
    }
    return true;
}

uint64_t lw_emulate_count(const LwHostTables& t, const uint8_t* hay, size_t len, uint64_t* redo_dwords) {
    Emu e{t, reinterpret_cast<const uint8_t*>(t.image.data()) + kLwClsBytes};
    uint64_t cnt = e.match_len(t.start), redo = 0;   // start-state matches (empty patterns) at the span start
    uint32_t h = t.start;
    size_t at = 0;
    if (t.flavour == kLwFull) {
        for (; at < len; at++) { h = e.fast(h, hay[at]); cnt += h & kLwFullLenMask; }
        if (redo_dwords) *redo_dwords = 0;
        return cnt;
    }
    for (; at + 4 <= len; at += 4) {
        const uint32_t h0 = h;
        uint32_t hk[4], worst = 0;
        for (int k = 0; k < 4; k++) { h = hk[k] = e.fast(h, hay[at + k]); worst = std::max(worst, h & 0xFFFFu); }
        if (worst >= t.virt_addr) {          // a multi state or poison: the handles are not exact
            redo++;
            h = h0;
            for (int k = 0; k < 4; k++) { h = e.careful(h, hay[at + k]); cnt += e.match_len(h); }
        } else if (worst >= t.fm_addr) {     // match states only: exact handles, one table lookup per matching byte
            for (int k = 0; k < 4; k++) if ((hk[k] & 0xFFFFu) >= t.fm_addr) cnt += e.rd16(t.mlen_off + ((hk[k] & 0xFFFFu) - t.fm_addr) / 2);
        }
    }
    for (; at < len; at++) { h = e.careful(h, hay[at]); cnt += e.match_len(h); }
    if (redo_dwords) *redo_dwords = redo;
    return cnt;
}

}  // namespace acgpu
