// Host-side tables of the LDS walk engine (device/lds_walk.hip): dense rows + single-exception handles + exception
// chains, laid out as the LDS image the kernel copies in.  Pure host code (no HIP), so the tables and the step rules
// are testable without a GPU: lw_emulate_count() below walks a haystack with exactly the kernel's fast-step /
// flag / exact-redo logic (tests/test_lw_tables.py compares it with the oracle through acgpu_test_lw_host).
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

// `order` = hid -> nnfa sid, `sid2hid` its inverse (build_hot_tables); first_match = first match hid.
bool build_lw_host(const NNfa& n, const Dfa& d, const std::vector<uint32_t>& order, const std::vector<uint32_t>& sid2hid,
                   uint32_t first_match, LwHostTables& out) {
    out = LwHostTables();
    const size_t nh = order.size();
    if (nh < 2 || nh > 60000) return false;
    const uint32_t su = n.special.start_unanchored_id, sa = n.special.start_anchored_id;
    const size_t alen = d.alphabet_len;
    auto drow = [&](uint32_t sid) { return &d.trans[size_t(sid) << d.stride2]; };

    // ---- the engine's own class map: reference classes with identical columns over every state are merged
    std::vector<uint32_t> cls_of_dclass(alen, 0), rep;   // rep[c] = a reference class of engine class c
    {
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
    }
    const uint32_t ncls = uint32_t(rep.size());
    if (ncls > 256) return false;
    // Row stride: ncls entries rounded up to an ODD number of dwords (a-z: 27, printable ASCII: 97), not to a power of two.
    // LDS has 64 banks of one dword: with a stride of 128 dwords the bank of a lookup is (class mod 64) whatever the row --
    // classes 0-31 and 64-95 share 32 of the banks, the other 32 serve one class each -- and the gathers of the walk pay
    // for the hot banks (SQ_LDS_BANK_CONFLICT 55 % of the LDS cycles, profiles/r03_hot_pmc.json); with an odd stride the
    // row index spreads the same classes over all banks.  The rows are also a quarter shorter, so more states get one.
    // (ACGPU_LW_POW2_ROWS=1: the power-of-two stride of rounds 1-3, for the A/B.)
    static const bool pow2_rows = std::getenv("ACGPU_LW_POW2_ROWS") != nullptr;
    uint32_t row_dw = ncls | 1u;
    if (pow2_rows) { row_dw = 1; while (row_dw < ncls) row_dw <<= 1; }
    const uint32_t row_bytes = 4u * row_dw;
    if (uint64_t(nh) * 4 + 2ull * row_bytes + kLwClsBytes + 64 > kLwLdsBudget) return false;   // deep[] alone would not fit
    std::vector<uint32_t> dl(nh * ncls, 0);   // class-compressed transition table over hids
    for (size_t h = 1; h < nh; h++) {
        const uint32_t* row = drow(order[h]);
        for (uint32_t c = 0; c < ncls; c++) {
            const uint32_t t = sid2hid[row[rep[c]] >> d.stride2];
            if (t == 0) return false;   // an unanchored Standard DFA never dies; the engine relies on it
            dl[h * ncls + c] = t;
        }
    }
    auto delta = [&](uint32_t h, uint32_t c) -> uint32_t { return dl[size_t(h) * ncls + c]; };

    // ---- dense-state selection, in breadth-first (fail-closed) order; retried with fewer rows if LDS overflows
    struct St { uint32_t D = 0; int32_t row = -1; std::vector<uint32_t> diff; };
    std::vector<St> st(nh);
    std::vector<uint32_t> bfs_h;   // hids in breadth-first order, start state first
    for (uint32_t s : n.bfs) if (s != sa) bfs_h.push_back(sid2hid[s]);
    const uint32_t h_start = sid2hid[su];
    // handle = {base: row of D, e: exception class, idx}.  Two layouts: 8 | 8 | 16 bits, and -- for alphabets of at most 64
    // engine classes, whose rows are short enough that LDS holds more than 254 of them (a-z sets: 128-byte rows, LDS has
    // room for ~700) -- 10 | 6 | 16.  The wide-base layout costs the fast step two more VALU operations (no SDWA byte
    // selects), so it is chosen only when the narrow one would have to turn states with rows into exception chains.
    const uint32_t lds_rows = (kLwLdsBudget - kLwClsBytes - 64 - uint32_t(4 * (nh + 1))) / row_bytes - 1;
    uint32_t want_rows = 0;   // states that would get a row if rows were free
    {
        std::vector<uint8_t> has_row(nh, 0);
        std::vector<uint32_t> Dn(nh, 0);
        const uint32_t hs = sid2hid[su];
        for (uint32_t s : n.bfs) {
            if (s == sa) continue;
            const uint32_t h = sid2hid[s];
            if (h == hs) { has_row[h] = 1; Dn[h] = h; want_rows++; continue; }
            const uint32_t f = sid2hid[n.fail[order[h]]];
            Dn[h] = has_row[f] ? f : Dn[f];
            uint32_t diffs = 0;
            for (uint32_t c = 0; c < ncls && diffs < 2; c++) diffs += delta(h, c) != delta(Dn[h], c);
            if (diffs >= 2) { has_row[h] = 1; Dn[h] = h; want_rows++; }
        }
    }
    const bool wide = ncls <= 64 && want_rows > 254 && lds_rows > 254;
    const uint32_t base_shift = wide ? 22 : 24, e_mask = wide ? 0x3Fu : 0xFFu;
    auto mk = [&](uint32_t base, uint32_t e, uint32_t idx) { return (base << base_shift) | (e << 16) | idx; };
    uint32_t max_rows = std::min<uint32_t>(wide ? 1022 : 254, lds_rows);
    uint32_t n_dense = 0, n_virtual = 0;
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
        // rows + poison row | deep | nxt (u32) + vhid (u16) per virtual slot | mlen (u16) per match state
        const uint64_t need = uint64_t(n_dense + 1) * row_bytes + 4ull * (nh + n_virtual + 1) + 6ull * n_virtual + 2ull * (nh - first_match) + kLwClsBytes + 64;
        if (need <= kLwLdsBudget && nh + n_virtual + 1 <= 65536) break;
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

    // match-list lengths as u16
    std::vector<uint16_t> mlen(nh - first_match, 0);
    for (size_t h = first_match; h < nh; h++) {
        const uint32_t o = order[h] - 2;   // DFA match-state index: (sid >> stride2) - 2 with sid = nnfa id << stride2 (dfa.rs:553-555)
        const uint32_t len = d.moff[o + 1] - d.moff[o];
        if (len > 0xFFFFu) return false;
        mlen[h - first_match] = uint16_t(len);
    }

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
    const uint32_t deep_off = (n_dense + 1) * row_bytes;   // offsets relative to kLwClsBytes
    const uint32_t nxt_off = deep_off + 4 * n_idx, vhid_off = nxt_off + 4 * n_virtual, mlen_off = (vhid_off + 2 * n_virtual + 3) & ~3u;
    const uint32_t image_bytes = (kLwClsBytes + mlen_off + 2 * uint32_t(mlen.size()) + 15) & ~15u;
    if (image_bytes > kLwLdsBudget) return false;
    std::vector<uint32_t> image(image_bytes / 4, poison);
    uint8_t* img = reinterpret_cast<uint8_t*>(image.data()) + kLwClsBytes;
    uint32_t* rows = reinterpret_cast<uint32_t*>(img);
    uint32_t* deep = reinterpret_cast<uint32_t*>(img + deep_off);
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
    deep[poison_idx] = poison;
    std::memcpy(img + mlen_off, mlen.data(), mlen.size() * 2);
    uint8_t* cls = reinterpret_cast<uint8_t*>(image.data());
    for (int b = 0; b < 256; b++) cls[b] = uint8_t(cls_of_dclass[d.byte_classes[b]]);

    out.image.swap(image);
    out.row_bytes = row_bytes;
    out.wide = wide;
    out.deep_off = deep_off;
    out.nxt_off = nxt_off; out.vhid_off = vhid_off; out.mlen_off = mlen_off;
    out.fm_addr = deep_off + 4 * first_match;
    out.poison_row = poison_row;
    out.start = H[h_start];
    out.n_dense = n_dense;
    out.n_multi = 0;
    for (uint32_t h : bfs_h) if (st[h].row < 0 && st[h].diff.size() >= 2) out.n_multi++;
    out.classes = ncls;
    out.first_match = first_match;
    out.n_states = uint32_t(nh);
    out.n_idx = n_idx;
    out.ok = true;
    return true;
}

// ---- CPU emulation of the kernel's walk over one cold-started range (test hook): 4 fast steps per dword with the
// deep-address flag, exact redo of flagged dwords, matches counted from the LDS match-length table.
namespace {
struct Emu {
    const LwHostTables& t;
    const uint8_t* img;   // image + kLwClsBytes
    uint32_t rd32(uint32_t a) const { uint32_t v; std::memcpy(&v, img + a, 4); return v; }
    uint32_t rd16(uint32_t a) const { uint16_t v; std::memcpy(&v, img + a, 2); return v; }
    uint32_t cls(uint8_t b) const { return reinterpret_cast<const uint8_t*>(t.image.data())[b]; }
    uint32_t deep_addr(uint32_t h) const { return t.deep_off + 4 * (h & 0xFFFFu); }
    uint32_t base_of(uint32_t h) const { return h >> (t.wide ? 22 : 24); }
    uint32_t e_of(uint32_t h) const { return (h >> 16) & (t.wide ? 0x3Fu : 0xFFu); }
    uint32_t fast(uint32_t h, uint8_t byte) const {
        const uint32_t c = cls(byte);
        const uint32_t ra = base_of(h) * t.row_bytes + 4 * c;
        return rd32(e_of(h) == c ? deep_addr(h) : ra);
    }
    uint32_t careful(uint32_t h, uint8_t byte) const {
        const uint32_t c = cls(byte);
        for (int hop = 0; hop < 4096; hop++) {
            const uint32_t idx = h & 0xFFFFu;
            if (e_of(h) == c) return rd32(t.deep_off + idx * 4);
            const uint32_t b = base_of(h);
            if (b != t.poison_row) return rd32(b * t.row_bytes + c * 4);
            h = rd32(t.nxt_off + (idx - t.n_states) * 4);
        }
        return h;
    }
    uint32_t match_len(uint32_t h) const {
        uint32_t idx = h & 0xFFFFu;
        if (idx >= t.n_states) idx = rd16(t.vhid_off + (idx - t.n_states) * 2);
        return idx >= t.first_match ? rd16(t.mlen_off + (idx - t.first_match) * 2) : 0u;
    }
};
}  // namespace

// Share of the dwords that take the exact path on a haystack that "looks like the patterns": 32 KiB of bytes drawn
// uniformly from the bytes that begin some pattern (the inputs the prefix filter hands over are of that kind).  The
// routing rule prices the LDS walk with it (device/hot.hpp: lw_route_cb).
double lw_estimate_redo(const LwHostTables& t) {
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

uint64_t lw_emulate_count(const LwHostTables& t, const uint8_t* hay, size_t len, uint64_t* redo_dwords) {
    Emu e{t, reinterpret_cast<const uint8_t*>(t.image.data()) + kLwClsBytes};
    uint64_t cnt = e.match_len(t.start), redo = 0;   // start-state matches (empty patterns) at the span start
    uint32_t h = t.start;
    size_t at = 0;
    for (; at + 4 <= len; at += 4) {
        const uint32_t h0 = h;
        uint32_t worst = 0;
        for (int k = 0; k < 4; k++) { h = e.fast(h, hay[at + k]); worst = std::max(worst, e.deep_addr(h)); }
        if (worst >= t.fm_addr) {
            redo++;
            h = h0;
            for (int k = 0; k < 4; k++) { h = e.careful(h, hay[at + k]); cnt += e.match_len(h); }
        }
    }
    for (; at < len; at++) { h = e.careful(h, hay[at]); cnt += e.match_len(h); }
    if (redo_dwords) *redo_dwords = redo;
    return cnt;
}

}  // namespace acgpu
