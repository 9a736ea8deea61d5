// Host-side tables of the contiguous-NFA "shallow-skip" walk (device/cnfa_tri.hip).
//
// The failure-link walk of src/nfa/contiguous.rs:186-247 spends almost all of its steps near the root: with 100 000
// patterns 89 % of the steps start in a state of depth 2 and fail straight back to depth 1 and on to another state of
// depth 2.  Those steps need no state at all: while the automaton's state has depth <= 2 it IS the longest suffix of the
// text that is a trie node of depth <= 2 -- a function of the last two bytes -- and the next byte c leaves that regime
// exactly when the trigram (a, b, c) of the last two bytes and c is a trie node of depth 3 (a longer suffix cannot be
// a node: its prefix of length >= 3 would have made the state deeper).  So the kernel keeps, in LDS, one bit per
// (pair of classes, class): "the trie has this node of depth 3"; a clear bit is a whole step (state -> fail -> ... ->
// the depth <= 2 state of the new pair) with no table access beyond that bit, a set bit is the transition into depth 3,
// where the literal walk over `repr` (state records, failure links) takes over until a failure link leads back to
// depth <= 2.  Pure host code; cnfa_tri_emulate_count() is the kernel's walk on the CPU (tests/test_cnfa_tables.py
// compares it with the oracle through acgpu_test_cnfa_host).
#include "cnfa_tri_tables.hpp"

#include <algorithm>
#include <utility>

namespace acgpu {

namespace {

using Edge = std::pair<uint32_t, uint32_t>;   // (class, target state)

// the trie edges out of state o (failure transitions and the start state's self loops are not edges)
void trie_children(const std::vector<uint32_t>& r, uint32_t alen, uint32_t start, uint32_t o, std::vector<Edge>& out) {
    out.clear();
    const uint32_t kind = r[o] & 0xFFu;
    if (kind == 0xFFu) {
        for (uint32_t k = 0; k < alen; k++) {
            const uint32_t t = r[o + 2 + k];
            if (t <= 1 || (o == start && t == start)) continue;
            out.emplace_back(k, t);
        }
    } else if (kind == 0xFEu) {
        out.emplace_back((r[o] >> 8) & 0xFFu, r[o + 2]);
    } else {
        const uint32_t tl = kind, cl = (tl + 3) >> 2;
        for (uint32_t i = 0; i < tl; i++)
            out.emplace_back((r[o + 2 + (i >> 2)] >> (8 * (i & 3))) & 0xFFu, r[o + 2 + cl + i]);
    }
}

uint32_t match_len_of(const CNfa& c, uint32_t sid) {   // contiguous.rs:581-598
    if (sid == 0 || sid > c.special.max_match_id) return 0;
    const uint32_t kind = c.repr[sid] & 0xFFu;
    const uint32_t base = kind == 0xFFu ? sid + 2 + uint32_t(c.alphabet_len) : sid + 2 + ((kind + 3) >> 2) + kind;
    const uint32_t packed = c.repr[base];
    return (packed & (1u << 31)) ? 1u : packed;
}

}  // namespace

bool build_cnfa_tri_host(const CNfa& c, CnfaTriHost& t) {
    t = CnfaTriHost();
    const std::vector<uint32_t>& r = c.repr;
    const uint32_t alen = uint32_t(c.alphabet_len);
    const uint32_t start = c.special.start_unanchored_id;
    if (r.empty() || start == 0 || alen == 0 || alen > 256 || r.size() >= (size_t(1) << 31)) return false;

    // the trie down to depth 3
    struct Node { uint32_t o, k1, k2, k3; };
    std::vector<Edge> e1, e2, e3;
    std::vector<Node> d1, d2, d3;
    std::vector<bool> used(alen, false);
    trie_children(r, alen, start, start, e1);
    for (const Edge& a : e1) {
        d1.push_back({a.second, a.first, 0, 0});
        used[a.first] = true;
        trie_children(r, alen, start, a.second, e2);
        for (const Edge& b : e2) {
            d2.push_back({b.second, a.first, b.first, 0});
            used[b.first] = true;
            trie_children(r, alen, start, b.second, e3);
            for (const Edge& d : e3) {
                d3.push_back({d.second, a.first, b.first, d.first});
                used[d.first] = true;
            }
        }
    }
    // every state, through the trie edges: the classes in use, and (below) the fail words to tag
    std::vector<uint32_t> all_states;
    {
        // (under ascii_case_insensitive a node is reached through two classes -- 'a' and 'A' are different byte classes
        // leading to ONE child -- so every state is visited once, not once per path)
        std::vector<uint32_t> todo{start};
        std::vector<bool> seen(r.size(), false);
        seen[start] = true;
        std::vector<Edge> ch;
        while (!todo.empty()) {
            const uint32_t o = todo.back(); todo.pop_back();
            all_states.push_back(o);
            trie_children(r, alen, start, o, ch);
            for (const Edge& e : ch) {
                used[e.first] = true;
                if (e.second < r.size() && !seen[e.second]) { seen[e.second] = true; todo.push_back(e.second); }
            }
        }
    }
    // compact classes: the classes that label a trie edge in ascending order, everything else = U
    std::vector<uint32_t> compact(alen, 0);
    uint32_t U = 0;
    for (uint32_t k = 0; k < alen; k++) if (used[k]) compact[k] = U++;
    for (uint32_t k = 0; k < alen; k++) if (!used[k]) compact[k] = U;
    if (U == 0 || U >= 255) return false;
    const uint32_t A = U + 1;
    const uint32_t bw = U / 32 + 1;   // bit U (the "no edge" class) exists and is never set
    const size_t pairs = size_t(A) * A;

    // which states have depth <= 2, and are any of them match states
    std::vector<uint32_t> shallow{start};
    for (const Node& n : d1) shallow.push_back(n.o);
    for (const Node& n : d2) shallow.push_back(n.o);
    std::sort(shallow.begin(), shallow.end());
    auto is_shallow = [&](uint32_t id) { return std::binary_search(shallow.begin(), shallow.end(), id); };
    for (uint32_t s : shallow) if (match_len_of(c, s)) t.shallow_matches = true;
    t.start_mlen = match_len_of(c, start);

    t.lds_bytes = pairs * (size_t(bw) * 4 + 2 + (t.shallow_matches ? 1 : 0)) + 512 + kTriLaneBuf;
    if (t.lds_bytes > kTriLdsBudget) return false;

    t.uc.assign(256, 0);
    t.inv.assign(256, 0);
    for (int b = 0; b < 256; b++) t.uc[b] = uint8_t(compact[c.byte_classes[b]]);
    for (uint32_t k = 0; k < alen; k++) if (used[k]) t.inv[compact[k]] = uint8_t(k);
    for (uint32_t k = 0; k < alen; k++) if (!used[k]) t.inv[U] = uint8_t(k);   // (if every class is in use no byte maps to U)

    // bitmap + children in (pair, compact class) order
    std::sort(d3.begin(), d3.end(), [&](const Node& x, const Node& y) {
        const uint64_t kx = (uint64_t(compact[x.k1]) * A + compact[x.k2]) * A + compact[x.k3];
        const uint64_t ky = (uint64_t(compact[y.k1]) * A + compact[y.k2]) * A + compact[y.k3];
        return kx < ky;
    });
    t.bits.assign(pairs * bw, 0);
    std::vector<uint32_t> n_of(pairs, 0);
    for (const Node& n : d3) {
        const size_t pr = size_t(compact[n.k1]) * A + compact[n.k2];
        const uint32_t uc = compact[n.k3];
        t.bits[pr * bw + (uc >> 5)] |= 1u << (uc & 31);
        n_of[pr]++;
    }
    uint32_t G = 1;
    for (;; G <<= 1) {
        if (G > 64) return false;
        uint64_t cur = 0;
        for (size_t p = 0; p < pairs; p++) cur += (uint64_t(n_of[p]) + G - 1) / G;
        if (cur <= 65535) break;
    }
    t.granule = G;
    t.base.assign(pairs, 0);
    {
        uint64_t cur = 0;
        for (size_t p = 0; p < pairs; p++) { t.base[p] = uint16_t(cur); cur += (uint64_t(n_of[p]) + G - 1) / G; }
        t.child.assign(size_t(cur) * G + 1, TriChild{0, 0, 0, 0});
    }

    // repr3: fail words that name a state of depth <= 2 are tagged
    std::vector<uint32_t> r3(r);
    r3.resize(r3.size() + kTriReprPad, 0);
    for (uint32_t o : all_states) if (is_shallow(r[o + 1])) r3[o + 1] = kTriShallow;
    {
        std::vector<uint32_t> fill(pairs, 0);
        for (const Node& n : d3) {
            const size_t pr = size_t(compact[n.k1]) * A + compact[n.k2];
            TriChild& e = t.child[size_t(t.base[pr]) * G + fill[pr]++];
            e.o = n.o; e.head = r3[n.o]; e.fail = r3[n.o + 1]; e.d0 = r3[n.o + 2];
        }
    }
    if (t.shallow_matches) {
        // the state of depth <= 2 a pair of classes stands for, and its match-list length
        std::vector<uint32_t> s1(A, start);
        for (const Node& n : d1) s1[compact[n.k1]] = n.o;
        t.st2.assign(pairs, start);
        for (uint32_t ua = 0; ua < A; ua++) for (uint32_t ub = 0; ub < A; ub++) t.st2[size_t(ua) * A + ub] = s1[ub];
        for (const Node& n : d2) t.st2[size_t(compact[n.k1]) * A + compact[n.k2]] = n.o;
        t.mc2.assign(pairs, 0);
        for (size_t p = 0; p < pairs; p++) {
            const uint32_t ml = match_len_of(c, t.st2[p]);
            if (ml > 0xFFu) return false;   // (one byte per pair in LDS)
            t.mc2[p] = uint8_t(ml);
        }
    }
    t.repr3.swap(r3);
    t.n_used = U; t.apair = A; t.bw = bw;
    t.ok = true;
    return true;
}

uint64_t cnfa_tri_emulate_count(const CnfaTriHost& t, const CNfa& c, const uint8_t* hay, size_t len, uint64_t* steps) {
    const uint32_t* r3 = t.repr3.data();
    const uint32_t A = t.apair, bw = t.bw, alen = uint32_t(c.alphabet_len);
    uint64_t g_child = 0, g_rec = 0, g_other = 0;
    uint64_t cnt = t.start_mlen;
    enum { SHALLOW, NOREC, REC } mode = SHALLOW;
    uint32_t o = 0, head = 0, fail = 0, d0 = 0, d1 = 0;
    bool have_d1 = false, pend = false;
    uint32_t pr = t.n_used * A + t.n_used, ub = t.n_used;
    auto word = [&](uint32_t i) -> uint32_t {   // word i of the current state's record
        if (i == 0) return head;
        if (i == 1) return fail;
        if (i == 2) return d0;
        if (i == 3 && have_d1) return d1;
        g_other++;
        return r3[o + i];
    };
    auto account = [&]() {
        if (o == 0 || o > c.special.max_match_id) return;
        const uint32_t kind = head & 0xFFu;
        const uint32_t base = kind == 0xFFu ? 2 + alen : (kind == 0xFEu ? 3u : 2 + ((kind + 3) >> 2) + kind);
        const uint32_t packed = word(base);
        cnt += (packed & (1u << 31)) ? 1u : packed;
    };
    for (size_t at = 0; at < len; at++) {
        const uint32_t k = c.byte_classes[hay[at]], uc = t.uc[hay[at]];
        const bool bit = (t.bits[size_t(pr) * bw + (uc >> 5)] >> (uc & 31)) & 1u;
        const uint32_t pr_new = ub * A + uc;
        bool consumed = false;
        while (!consumed || mode == NOREC) {
            if (mode == REC && !consumed) {   // one step of contiguous.rs:186-247 from the record in hand
                const uint32_t kind = head & 0xFFu;
                bool found = false;
                uint32_t target = 0;
                if (kind == 0xFEu) {
                    if (k == ((head >> 8) & 0xFFu)) { found = true; target = d0; }
                } else if (kind == 0xFFu) {
                    g_other++;
                    const uint32_t nx = r3[o + 2 + k];
                    if (nx != 1u) { found = true; target = nx; }
                } else {
                    const uint32_t tl = kind, cl = (tl + 3) >> 2;
                    for (uint32_t i = 0; i < tl && !found; i++) {
                        const uint32_t w = word(2 + (i >> 2));
                        if (((w >> (8 * (i & 3))) & 0xFFu) == k) { found = true; target = word(2 + cl + i); }
                    }
                }
                if (found) { o = target; consumed = true; mode = NOREC; pend = true; }
                else if (fail & kTriShallow) mode = SHALLOW;
                else { o = fail; mode = NOREC; }
            }
            if (mode == SHALLOW && !consumed) {
                consumed = true;
                if (bit) {   // the trie node (a, b, c): one gather brings the state and the head of its record
                    const uint32_t* wv = &t.bits[size_t(pr) * bw];
                    uint32_t rank = 0;
                    for (uint32_t i = 0; i < (uc >> 5); i++) rank += uint32_t(__builtin_popcount(wv[i]));
                    rank += uint32_t(__builtin_popcount(wv[uc >> 5] & ((1u << (uc & 31)) - 1)));
                    const TriChild& e = t.child[size_t(t.base[pr]) * t.granule + rank];
                    g_child++;
                    o = e.o; head = e.head; fail = e.fail; d0 = e.d0; have_d1 = false;
                    mode = REC;
                    account();
                } else if (t.shallow_matches) {
                    cnt += t.mc2[pr_new];
                }
            }
            if (mode == NOREC) {
                g_rec++;
                head = r3[o]; fail = r3[o + 1]; d0 = r3[o + 2]; d1 = r3[o + 3]; have_d1 = true;
                mode = REC;
                if (pend) { account(); pend = false; }
            }
        }
        pr = pr_new; ub = uc;
    }
    if (steps) { steps[0] = g_child; steps[1] = g_rec; steps[2] = g_other; }
    return cnt;
}

}  // namespace acgpu
