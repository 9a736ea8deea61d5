// Host-side tables of the DFA "shallow-skip" transition walk (device/dfa_tri.hip): the same exact skip of the depth <= 2
// regime as the contiguous-NFA walk's (cnfa_tri_tables.cpp explains it) in front of the per-byte primitive
// `sid = trans[sid + classes[byte]]` (src/dfa.rs:218-226).  While the state has depth <= 2 it is a function of the last
// two bytes and the next byte leaves that regime exactly when the trigram is a trie node: one LDS bit.  Below depth 2
// the walk is the reference's: one table load per byte -- until a transition lands on a state of depth <= 2 again, which
// the device's copy of the table says by a tag bit on the target.
// The trie comes from the noncontiguous NFA the DFA was built from (DFA state index == nNFA state id for the
// single-start layouts, src/dfa.rs:553-560).  Pure host code.
#include "dfa_tri_tables.hpp"

#include <algorithm>
#include <utility>

namespace acgpu {

bool build_dfa_tri_host(const NNfa& n, const Dfa& d, DfaTriHost& t) {
    t = DfaTriHost();
    const uint32_t start_n = n.special.start_unanchored_id;
    const uint32_t s2 = uint32_t(d.stride2), alen = uint32_t(d.alphabet_len);
    if (d.trans.empty() || d.special.start_unanchored_id == 0 || d.special.start_anchored_id != 0) return false;   // unanchored single-start layout only
    if (d.state_len != n.states() || d.special.start_unanchored_id != (start_n << s2)) return false;
    if (d.trans.size() >= (size_t(1) << 31) || d.trans.size() * 4 > (size_t(512) << 20)) return false;   // (a tagged copy of the table is made)
    using Edge = std::pair<uint32_t, uint32_t>;   // (class, nNFA child)
    auto children = [&](uint32_t s, std::vector<Edge>& out) {
        out.clear();
        for (uint32_t i = n.toff[s]; i < n.toff[s + 1]; i++) {
            const uint32_t c = n.tnext[i];
            if (c <= 1 || c == start_n) continue;   // (the start state's self loops are not trie edges)
            const uint32_t k = d.byte_classes[n.tbyte[i]];
            bool seen = false;
            for (const Edge& e : out) if (e.first == k) { seen = true; break; }   // (bytes of one class lead to one child)
            if (!seen) out.emplace_back(k, c);
        }
    };
    struct Node { uint32_t s, k1, k2, k3; };
    std::vector<Edge> e1, e2, e3, ch;
    std::vector<Node> d1, d2, d3;
    std::vector<bool> used(alen, false);
    children(start_n, e1);
    for (const Edge& a : e1) {
        d1.push_back({a.second, a.first, 0, 0});
        children(a.second, e2);
        for (const Edge& b : e2) {
            d2.push_back({b.second, a.first, b.first, 0});
            children(b.second, e3);
            for (const Edge& c : e3) d3.push_back({c.second, a.first, b.first, c.first});
        }
    }
    std::vector<uint8_t> depth_le2(n.states(), 0);
    depth_le2[start_n] = 1;
    for (const Node& x : d1) depth_le2[x.s] = 1;
    for (const Node& x : d2) depth_le2[x.s] = 1;
    {   // the classes in use: every trie edge.  (Under ascii_case_insensitive a node is reached through two classes --
        // 'a' and 'A' are different byte classes leading to ONE child -- so every state is visited once, not once per path:
        // a 160-byte pattern has 2^160 of those.)
        std::vector<uint32_t> todo{start_n};
        std::vector<bool> seen(n.states(), false);
        seen[start_n] = true;
        while (!todo.empty()) {
            const uint32_t s = todo.back(); todo.pop_back();
            children(s, ch);
            for (const Edge& e : ch) {
                used[e.first] = true;
                if (!seen[e.second]) { seen[e.second] = true; todo.push_back(e.second); }
            }
        }
    }
    std::vector<uint32_t> compact(alen, 0);
    uint32_t U = 0;
    for (uint32_t k = 0; k < alen; k++) if (used[k]) compact[k] = U++;
    for (uint32_t k = 0; k < alen; k++) if (!used[k]) compact[k] = U;
    if (U == 0 || U >= 255) return false;
    const uint32_t A = U + 1, bw = U / 32 + 1;
    const size_t pairs = size_t(A) * A;
    auto mlen = [&](uint32_t s) -> uint32_t {   // match-list length of nNFA/DFA state s (dfa.rs:275-286)
        return (s >= 2 && s <= n.special.max_match_id) ? d.moff[s - 2 + 1] - d.moff[s - 2] : 0u;
    };
    for (uint32_t s = 0; s < n.states(); s++) if (depth_le2[s] && mlen(s)) t.shallow_matches = true;
    t.start_mlen = mlen(start_n);
    t.lds_bytes = pairs * (size_t(bw) * 4 + 2 + (t.shallow_matches ? 1 : 0)) + 512 + kTriLaneBuf;
    if (t.lds_bytes > kTriLdsBudget) return false;

    t.uc.assign(256, 0);
    t.inv.assign(256, 0);
    for (int b = 0; b < 256; b++) t.uc[b] = uint8_t(compact[d.byte_classes[b]]);
    for (uint32_t k = 0; k < alen; k++) if (used[k]) t.inv[compact[k]] = uint8_t(k);
    for (uint32_t k = 0; k < alen; k++) if (!used[k]) t.inv[U] = uint8_t(k);

    std::sort(d3.begin(), d3.end(), [&](const Node& x, const Node& y) {
        const uint64_t kx = (uint64_t(compact[x.k1]) * A + compact[x.k2]) * A + compact[x.k3];
        const uint64_t ky = (uint64_t(compact[y.k1]) * A + compact[y.k2]) * A + compact[y.k3];
        return kx < ky;
    });
    t.bits.assign(pairs * bw, 0);
    std::vector<uint32_t> n_of(pairs, 0);
    for (const Node& x : d3) {
        const size_t pr = size_t(compact[x.k1]) * A + compact[x.k2];
        const uint32_t uc = compact[x.k3];
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
        t.child.assign(size_t(cur) * G + 4, 0);
        std::vector<uint32_t> fill(pairs, 0);
        for (const Node& x : d3) {
            const size_t pr = size_t(compact[x.k1]) * A + compact[x.k2];
            t.child[size_t(t.base[pr]) * G + fill[pr]++] = x.s << s2;
        }
    }
    t.trans3 = d.trans;
    for (uint32_t& w : t.trans3) if (w != 0 && depth_le2[w >> s2]) w |= kTriShallow;
    if (t.shallow_matches) {
        std::vector<uint32_t> s1(A, start_n);
        for (const Node& x : d1) s1[compact[x.k1]] = x.s;
        t.st2.assign(pairs, start_n << s2);
        for (uint32_t ua = 0; ua < A; ua++) for (uint32_t ub = 0; ub < A; ub++) t.st2[size_t(ua) * A + ub] = s1[ub] << s2;
        for (const Node& x : d2) t.st2[size_t(compact[x.k1]) * A + compact[x.k2]] = x.s << s2;
        t.mc2.assign(pairs, 0);
        for (size_t p = 0; p < pairs; p++) {
            const uint32_t ml = mlen(t.st2[p] >> s2);
            if (ml > 0xFFu) return false;   // (one byte per pair in LDS)
            t.mc2[p] = uint8_t(ml);
        }
    }
    t.n_used = U; t.apair = A; t.bw = bw;
    t.ok = true;
    return true;
}

}  // namespace acgpu
