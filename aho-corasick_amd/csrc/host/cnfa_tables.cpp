// Host-side tables of the contiguous-NFA walk kernel (device/cnfa_walk.hip): which states live in LDS, which dense states
// get a row table of their own, and the patched copy of `repr` that names both by index.  Pure host code: cnfa_emulate_count() walks a haystack with the kernel's own
// step (tests/test_cnfa_tables.py compares it with the oracle through acgpu_test_cnfa_host).
#include "cnfa_tables.hpp"

#include <algorithm>
#include <unordered_map>

namespace acgpu {

// Host: which states go to LDS -- the start state (slot 0) and its children while they are dense and LDS lasts -- and the
// patched copy of `repr` that names them by slot.
bool build_cnfa_hot_host(const CNfa& c, CnfaHotHost& t) {
    t = CnfaHotHost();
    const uint32_t alen = uint32_t(c.alphabet_len);
    const uint32_t row_words = alen + 1;
    const std::vector<uint32_t>& r = c.repr;
    const uint32_t start = c.special.start_unanchored_id;
    if (r.empty() || start == 0 || (r[start] & 0xFFu) != 0xFFu) return false;   // no unanchored start / not dense
    // state words carry the tags kCnfaSlotTag / kCnfaMidTag in their top bits: a `repr` that reaches them cannot be named
    // (the reference allows ids up to 2^31 - 2; such an automaton keeps the literal walk)
    if (r.size() + kCnfaReprPad >= kCnfaMidTag) return false;
    const size_t lds_budget = 80 * 1024 - 256 - 1024;   // two workgroups per CU
    const uint32_t max_slots = uint32_t(std::min<size_t>(254, lds_budget / (size_t(row_words) * 4 + 4)));
    if (max_slots < 1) return false;
    std::vector<uint32_t> ids{start};
    for (uint32_t k = 0; k < alen && ids.size() < max_slots; k++) {
        const uint32_t t = r[start + 2 + k];
        if (t == 1 /*FAIL*/ || t == 0 || t == start) continue;
        if ((r[t] & 0xFFu) != 0xFFu) continue;                       // only dense records have the row layout
        if (std::find(ids.begin(), ids.end(), t) == ids.end()) ids.push_back(t);
    }
    std::unordered_map<uint32_t, int> slot_map;
    for (size_t q = 0; q < ids.size(); q++) slot_map.emplace(ids[q], int(q));
    auto slot_of = [&](uint32_t id) -> int { const auto it = slot_map.find(id); return it == slot_map.end() ? -1 : it->second; };
    // second tier ("mid" states): the dense states that do not fit LDS -- the grandchildren of the start state -- in
    // breadth-first order while their fail words and match counts fit LDS (6 bytes each).  Their rows go to a table of
    // their own in global memory, so a step from one of them is ONE gather (the transition) instead of two (state header
    // + speculative transition): the state word says what the header would.
    std::unordered_map<uint32_t, uint32_t> mid_map;
    std::vector<uint32_t> mids;
    if (r.size() < (size_t(1) << 30)) {
        std::vector<uint32_t> queue{start};
        std::vector<bool> seen(r.size(), false);
        seen[start] = true;
        for (size_t qi = 0; qi < queue.size() && mids.size() < kCnfaMaxMid; qi++) {
            const uint32_t o = queue[qi];
            if ((r[o] & 0xFFu) != 0xFFu) continue;   // (dense states form the top of the trie: nothing dense below a sparse state)
            if (slot_of(o) < 0) { mid_map.emplace(o, uint32_t(mids.size())); mids.push_back(o); }
            for (uint32_t k = 0; k < alen; k++) {
                const uint32_t x = r[o + 2 + k];
                if (x > 1 && x < r.size() && !seen[x]) { seen[x] = true; queue.push_back(x); }
            }
        }
    }
    auto tagged = [&](uint32_t id) -> uint32_t {
        const int q = slot_of(id);
        if (q >= 0) return kCnfaSlotTag | uint32_t(q);
        const auto it = mid_map.find(id);
        return it == mid_map.end() ? id : (kCnfaMidTag | it->second);
    };
    std::vector<uint32_t> rt(r);   // the patched copy: fail words and transition targets that name an LDS-resident state
    rt.resize(rt.size() + kCnfaReprPad, 0);
    // what the states outside LDS look like: a traversal of the trie edges from the start state
    bool dense_outside = false, sorted_sparse = true;
    {
        std::vector<uint32_t> todo{start};
        std::vector<bool> seen(r.size(), false);
        seen[start] = true;
        auto visit = [&](uint32_t t) { if (t > 1 && t < r.size() && !seen[t]) { seen[t] = true; todo.push_back(t); } };
        while (!todo.empty()) {
            const uint32_t o = todo.back(); todo.pop_back();
            const uint32_t kind = r[o] & 0xFFu;
            rt[o + 1] = tagged(r[o + 1]);
            if (kind == 0xFFu) {
                if (slot_of(o) < 0 && !mid_map.count(o)) dense_outside = true;
                for (uint32_t k = 0; k < alen; k++) { rt[o + 2 + k] = tagged(r[o + 2 + k]); visit(r[o + 2 + k]); }
            } else if (kind == 0xFEu) {
                rt[o + 2] = tagged(r[o + 2]);
                visit(r[o + 2]);
            } else {
                const uint32_t tl = kind, cl = (tl + 3) >> 2;
                uint32_t prev = 0;
                for (uint32_t i = 0; i < tl; i++) {
                    const uint32_t c8 = (r[o + 2 + (i >> 2)] >> (8 * (i & 3))) & 0xFFu;
                    if (i && c8 <= prev) sorted_sparse = false;
                    prev = c8;
                    rt[o + 2 + cl + i] = tagged(r[o + 2 + cl + i]);
                    visit(r[o + 2 + cl + i]);
                }
            }
        }
    }
    t.dense_outside = dense_outside;
    t.sorted_sparse = sorted_sparse;
    t.rows.assign(ids.size() * row_words, 0);
    t.mcnt.assign(ids.size(), 0);
    for (size_t q = 0; q < ids.size(); q++) {
        t.rows[q * row_words] = tagged(r[ids[q] + 1]);
        for (uint32_t k = 0; k < alen; k++) t.rows[q * row_words + 1 + k] = tagged(r[ids[q] + 2 + k]);
        if (ids[q] != 0 && ids[q] <= c.special.max_match_id) {   // a match state: its list length (contiguous.rs:581-598)
            const uint32_t packed = r[ids[q] + 2 + alen];
            t.mcnt[q] = (packed & (1u << 31)) ? 1u : packed;
            t.slot_matches = true;
        }
    }
    while ((1u << t.mid_shift) < alen) t.mid_shift++;   // rows of 2^mid_shift words: the row address is a shift
    t.mid_rows.assign(mids.size() << t.mid_shift, 0);
    t.mid_fail.assign(mids.size(), 0);
    t.mid_mcnt.assign(mids.size(), 0);
    for (size_t i = 0; i < mids.size(); i++) {
        t.mid_fail[i] = tagged(r[mids[i] + 1]);
        for (uint32_t k = 0; k < alen; k++) t.mid_rows[(i << t.mid_shift) + k] = tagged(r[mids[i] + 2 + k]);
        if (mids[i] <= c.special.max_match_id) {
            const uint32_t packed = r[mids[i] + 2 + alen];
            const uint32_t ml = (packed & (1u << 31)) ? 1u : packed;
            if (ml > 0xFFFFu) return false;   // (u16 counts in LDS)
            t.mid_mcnt[i] = uint16_t(ml);
            t.mid_matches = true;
        }
    }
    t.repr_t.swap(rt);
    t.row_words = row_words;
    t.n_slots = uint32_t(ids.size());
    t.n_mid = uint32_t(mids.size());
    t.ok = true;
    return true;
}

// The kernel's step (CnfaFastStep::step, device/cnfa_walk.hip) over haystack[0..len), cold start at 0; returns the
// overlapping search's match count (start-state matches of an empty pattern included).
uint64_t cnfa_emulate_count(const CnfaHotHost& t, const CNfa& c, const uint8_t* hay, size_t len) {
    const uint32_t* repr = t.repr_t.data();
    const uint32_t rw = t.row_words;
    auto is_match = [&](uint32_t sid) { return sid != 0 && sid <= c.special.max_match_id; };
    auto match_len = [&](uint32_t sid) -> uint32_t {   // contiguous.rs:581-598 on the patched copy (match words are not patched)
        const uint32_t kind = repr[sid] & 0xFFu;
        const uint32_t base = kind == 0xFFu ? sid + 2 + uint32_t(c.alphabet_len) : sid + 2 + ((kind + 3) >> 2) + kind;
        const uint32_t packed = repr[base];
        return (packed & (1u << 31)) ? 1u : packed;
    };
    uint64_t cnt = t.mcnt[0];   // slot 0 = the unanchored start state
    uint32_t sid = kCnfaSlotTag;
    for (size_t at = 0; at < len; at++) {
        const uint32_t k = c.byte_classes[hay[at]];
        uint32_t o = sid;
        for (;;) {
            if (o & kCnfaSlotTag) {
                const uint32_t row = (o & 0xFFFFu) * rw;
                const uint32_t nx = t.rows[row + 1 + k];
                if (nx != 1u /*FAIL*/) { o = nx; break; }
                o = t.rows[row];
                continue;
            }
            if (o & kCnfaMidTag) {
                const uint32_t idx = o & (kCnfaMidTag - 1);
                const uint32_t nx = t.mid_rows[(size_t(idx) << t.mid_shift) + k];
                if (nx != 1u) { o = nx; break; }
                o = t.mid_fail[idx];
                continue;
            }
            const uint32_t head = repr[o], fail = repr[o + 1];
            const uint32_t kind = head & 0xFFu;
            bool found = false;
            if (kind == 0xFFu) {
                const uint32_t nx = repr[o + 2 + k];
                if (nx != 1u) { o = nx; found = true; }
            } else if (kind == 0xFEu) {
                if (k == ((head >> 8) & 0xFFu)) { o = repr[o + 2]; found = true; }
            } else {
                const uint32_t tl = kind, cl = (tl + 3) >> 2;
                for (uint32_t i = 0; i < tl; i++) {
                    const uint32_t c8 = (repr[o + 2 + (i >> 2)] >> (8 * (i & 3))) & 0xFFu;
                    if (c8 == k) { o = repr[o + 2 + cl + i]; found = true; break; }
                    if (t.sorted_sparse && c8 > k) break;   // the kernel stops at the first larger class (word-wise)
                }
            }
            if (found) break;
            o = fail;
        }
        sid = o;
        if (o & kCnfaSlotTag) cnt += t.mcnt[o & 0xFFFFu];
        else if (o & kCnfaMidTag) cnt += t.mid_mcnt[o & (kCnfaMidTag - 1)];
        else if (o == 0) break;                       // DEAD (anchored automata only)
        else if (is_match(o)) cnt += match_len(o);
    }
    return cnt;
}

}  // namespace acgpu
