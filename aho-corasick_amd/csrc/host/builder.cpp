// CPU construction of the automata (product code; NOT the oracle).
//
// Produces tables identical to the reference's -- same state numbering, same
// match-list order -- with different machinery:
//   * trie edges live in an open-addressing hash (parent,byte)->child plus a
//     counting-sorted CSR, instead of sorted in-vector linked lists
//     (ref: src/nfa/noncontiguous.rs:381-423);
//   * the shuffle is a position permutation that replays the reference's swap
//     sequence (ref: src/nfa/noncontiguous.rs:1399-1481, src/util/remapper.rs:104-154);
//   * DFA rows are filled by breadth-first row propagation
//     (row(s) = row(fail(s)) overridden by s's own edges), O(states * alphabet),
//     instead of one failure-chain walk per cell (ref: src/dfa.rs:544-607).
// What must match the reference exactly, and where it is decided there:
//   pattern -> trie insertion, leftmost-first abandon rule   noncontiguous.rs:1057-1150
//   byte classes                                             alphabet.rs:224-250
//   failure BFS order, match-list concatenation order        noncontiguous.rs:1275-1374
//   leftmost start-loop closing                              noncontiguous.rs:1620-1638
//   special ids                                              noncontiguous.rs:1462-1479
//   DFA layout (one start / both starts)                     dfa.rs:431-724
//   contiguous NFA word format                               contiguous.rs:686-820, 937-1009
#include <algorithm>
#include <cstring>
#include <numeric>

#include "automaton.hpp"

namespace acgpu {

namespace {

inline uint8_t opposite_ascii_case(uint8_t b) {
    if (b >= 'A' && b <= 'Z') return uint8_t(b + 32);
    if (b >= 'a' && b <= 'z') return uint8_t(b - 32);
    return b;
}

// (parent, byte) -> child, open addressing, power-of-two capacity.
class EdgeMap {
  public:
    EdgeMap() { rehash(1024); }
    uint32_t find(uint32_t parent, uint8_t byte) const {
        uint64_t key = (uint64_t(parent) << 8) | byte;
        size_t i = slot(key);
        for (;;) {
            if (keys_[i] == kEmpty) return kFail;
            if (keys_[i] == key) return vals_[i];
            i = (i + 1) & mask_;
        }
    }
    void insert(uint32_t parent, uint8_t byte, uint32_t child) {
        if ((count_ + 1) * 10 > keys_.size() * 6) rehash(keys_.size() * 2);
        uint64_t key = (uint64_t(parent) << 8) | byte;
        size_t i = slot(key);
        while (keys_[i] != kEmpty) i = (i + 1) & mask_;
        keys_[i] = key;
        vals_[i] = child;
        count_++;
    }

  private:
    static constexpr uint64_t kEmpty = ~0ull;
    size_t slot(uint64_t key) const {
        key ^= key >> 33; key *= 0xff51afd7ed558ccdull; key ^= key >> 33;
        return size_t(key) & mask_;
    }
    void rehash(size_t cap) {
        std::vector<uint64_t> ok; std::vector<uint32_t> ov;
        ok.swap(keys_); ov.swap(vals_);
        keys_.assign(cap, kEmpty); vals_.assign(cap, 0); mask_ = cap - 1; count_ = 0;
        for (size_t i = 0; i < ok.size(); i++)
            if (ok[i] != kEmpty) insert(uint32_t(ok[i] >> 8), uint8_t(ok[i] & 0xFF), ov[i]);
    }
    std::vector<uint64_t> keys_;
    std::vector<uint32_t> vals_;
    size_t mask_ = 0, count_ = 0;
};

struct Edge { uint32_t parent; uint8_t byte; uint32_t child; };

size_t stride2_of(size_t alphabet_len) {
    size_t p = 1, z = 0;
    while (p < alphabet_len) { p <<= 1; z++; }
    return z;
}

}  // namespace

uint32_t NNfa::follow(uint32_t sid, uint8_t byte) const {
    uint32_t lo = toff[sid], hi = toff[sid + 1];
    if (hi - lo == 256) return tnext[lo + byte];  // start states / DEAD: all 256 explicit
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (tbyte[mid] < byte) lo = mid + 1; else hi = mid;
    }
    if (lo < toff[sid + 1] && tbyte[lo] == byte) return tnext[lo];
    return kFail;
}

uint32_t NNfa::next_state(bool anchored, uint32_t sid, uint8_t byte) const {
    for (;;) {
        uint32_t next = follow(sid, byte);
        if (next != kFail) return next;
        if (anchored) return kDead;
        sid = fail[sid];
    }
}

acgpu_status build_nnfa(const BuildOptions& o, const uint8_t* const* pats, const size_t* lens, size_t npats,
                        NNfa& out) {
    const bool leftmost = o.match_kind != ACGPU_MATCH_STANDARD;
    const bool casei = o.ascii_case_insensitive;
    constexpr uint32_t START_U = 2, START_A = 3;  // allocation order: DEAD, FAIL, START_U, START_A

    // ---- 1. trie (ids in allocation order) ----
    EdgeMap emap;
    std::vector<Edge> edges;
    std::vector<uint32_t> depth = {0, 0, 0, 0};
    std::vector<std::pair<uint32_t, uint32_t>> owns;  // (node, pid) in pid order
    std::vector<uint8_t> has_own = {0, 0, 0, 0};
    bool byteset[256] = {false};
    out = NNfa();
    out.match_kind = o.match_kind;
    out.n_patterns = npats;
    out.pattern_lens.assign(o.pattern_ids ? o.id_space : npats, 0);
    for (size_t i = 0; i < npats; i++) {
        if (i > kSmallIndexMax) return ACGPU_ERR_PATTERN_ID_OVERFLOW;
        const uint8_t* pat = pats[i];
        const size_t plen = lens[i];
        if (plen > kSmallIndexMax) return ACGPU_ERR_PATTERN_TOO_LONG;
        const uint32_t pid = o.pattern_ids ? o.pattern_ids[i] : uint32_t(i);
        if (pid >= out.pattern_lens.size()) return ACGPU_ERR_INVALID_ARGUMENT;
        out.min_pattern_len = std::min(out.min_pattern_len, plen);
        out.max_pattern_len = std::max(out.max_pattern_len, plen);
        out.pattern_lens[pid] = uint32_t(plen);
        uint32_t prev = START_U;
        bool saw_match = false, abandoned = false;
        for (size_t d = 0; d < plen; d++) {
            const uint8_t b = pat[d];
            saw_match = saw_match || has_own[prev];
            if (o.match_kind == ACGPU_MATCH_LEFTMOST_FIRST && saw_match) { abandoned = true; break; }
            byteset[b] = true;                 // set_range(b,b): boundaries at b-1 and b
            if (b > 0) byteset[b - 1] = true;
            if (casei) {
                uint8_t ob = opposite_ascii_case(b);
                byteset[ob] = true;
                if (ob > 0) byteset[ob - 1] = true;
            }
            uint32_t next = emap.find(prev, b);
            if (next == kFail) {
                if (depth.size() > kSmallIndexMax) return ACGPU_ERR_STATE_ID_OVERFLOW;
                next = uint32_t(depth.size());
                depth.push_back(uint32_t(d));  // stored depth == distance-1 (noncontiguous.rs:1100,1136)
                has_own.push_back(0);
                emap.insert(prev, b, next);
                edges.push_back({prev, b, next});
                if (casei) {
                    uint8_t ob = opposite_ascii_case(b);
                    if (ob != b) { emap.insert(prev, ob, next); edges.push_back({prev, ob, next}); }
                }
            }
            prev = next;
        }
        if (abandoned) continue;
        owns.emplace_back(prev, pid);
        has_own[prev] = 1;
    }
    const size_t N = depth.size();
    if (edges.size() + 3 * 256 + 1 > kSmallIndexMax) return ACGPU_ERR_STATE_ID_OVERFLOW;

    // ---- 2. byte classes (alphabet.rs:235-250) ----
    {
        uint8_t cls = 0;
        for (int b = 0;; b++) {
            out.byte_classes[b] = cls;
            if (b == 255) break;
            if (byteset[b]) cls++;
        }
    }

    // children CSR by parent, sorted by byte (counting sort; casei edges included)
    std::vector<uint32_t> coff(N + 1, 0);
    for (const Edge& e : edges) coff[e.parent + 1]++;
    for (size_t i = 0; i < N; i++) coff[i + 1] += coff[i];
    std::vector<uint8_t> cbyte(edges.size());
    std::vector<uint32_t> cnext(edges.size());
    {
        std::vector<uint32_t> cur(coff.begin(), coff.end() - 1);
        for (const Edge& e : edges) { uint32_t k = cur[e.parent]++; cbyte[k] = e.byte; cnext[k] = e.child; }
        for (size_t s = 0; s < N; s++) {
            uint32_t lo = coff[s], hi = coff[s + 1];
            if (hi - lo > 1) {  // small insertion sort keyed by byte
                for (uint32_t a = lo + 1; a < hi; a++) {
                    uint8_t kb = cbyte[a]; uint32_t kn = cnext[a]; uint32_t j = a;
                    while (j > lo && cbyte[j - 1] > kb) { cbyte[j] = cbyte[j - 1]; cnext[j] = cnext[j - 1]; j--; }
                    cbyte[j] = kb; cnext[j] = kn;
                }
            }
        }
    }

    // ---- 3. failure links + match lists, breadth first (noncontiguous.rs:1275-1374) ----
    // follow() during construction: START_U has the self loop, DEAD loops, trie nodes use the hash.
    auto follow0 = [&](uint32_t s, uint8_t b) -> uint32_t {
        if (s == kDead) return kDead;
        uint32_t c = emap.find(s, b);
        if (c != kFail) return c;
        return s == START_U ? START_U : kFail;
    };
    std::vector<uint32_t> fail(N, START_U);
    fail[kDead] = 0; fail[kFail] = 0; fail[START_U] = 0;  // alloc_state reads start id before it is set
    fail[START_A] = kDead;                                 // set_anchored_start_state :1584
    // match lists: lazily allocated vectors
    std::vector<int32_t> lidx(N, -1);
    std::vector<std::vector<uint32_t>> lists;
    auto list_of = [&](uint32_t s) -> std::vector<uint32_t>& {
        if (lidx[s] < 0) { lidx[s] = int32_t(lists.size()); lists.emplace_back(); }
        return lists[size_t(lidx[s])];
    };
    auto has_list = [&](uint32_t s) { return lidx[s] >= 0 && !lists[size_t(lidx[s])].empty(); };
    auto append_list = [&](uint32_t src, uint32_t dst) {
        if (!has_list(src)) return;
        const size_t si = size_t(lidx[src]);
        const size_t n0 = lists[si].size();
        std::vector<uint32_t>& d = list_of(dst);      // may reallocate `lists`
        const std::vector<uint32_t>& s = lists[si];
        d.reserve(d.size() + n0);
        for (size_t k = 0; k < n0; k++) d.push_back(s[k]);
    };
    for (auto& pr : owns) list_of(pr.first).push_back(pr.second);
    append_list(START_U, START_A);  // :1577 (before any BFS appends)

    std::vector<uint32_t> queue;
    queue.reserve(N);
    std::vector<uint8_t> seen(casei ? N : 0, 0);
    for (uint32_t k = coff[START_U]; k < coff[START_U + 1]; k++) {
        uint32_t next = cnext[k];
        if (casei) { if (seen[next]) continue; seen[next] = 1; }
        queue.push_back(next);
        if (leftmost && has_list(next)) fail[next] = kDead;
    }
    for (size_t qh = 0; qh < queue.size(); qh++) {
        const uint32_t id = queue[qh];
        for (uint32_t k = coff[id]; k < coff[id + 1]; k++) {
            const uint32_t next = cnext[k];
            const uint8_t byte = cbyte[k];
            if (casei) { if (seen[next]) continue; seen[next] = 1; }
            queue.push_back(next);
            if (leftmost && has_list(next)) { fail[next] = kDead; continue; }
            uint32_t f = fail[id];
            while (follow0(f, byte) == kFail) f = fail[f];
            f = follow0(f, byte);
            fail[next] = f;
            append_list(f, next);
        }
        if (!leftmost) append_list(START_U, id);
    }
    // every list entry is one slot of the reference's `matches` vector
    {
        uint64_t total = 1;
        for (auto& l : lists) total += l.size();
        if (total > kSmallIndexMax) return ACGPU_ERR_STATE_ID_OVERFLOW;
    }
    const bool close_loop = leftmost && has_list(START_U);  // :1620-1638

    // ---- 4. shuffle: DEAD, FAIL, MATCH.., START_U, START_A, NON-MATCH.. (:1399-1481) ----
    std::vector<uint32_t> perm(N);  // position -> original id
    std::iota(perm.begin(), perm.end(), 0u);
    uint32_t next_avail = 4;
    for (uint32_t i = 4; i < N; i++) {
        if (!has_list(perm[i])) continue;
        std::swap(perm[i], perm[next_avail]);
        next_avail++;
    }
    std::swap(perm[START_A], perm[next_avail - 1]);
    std::swap(perm[START_U], perm[next_avail - 2]);
    std::vector<uint32_t> newid(N);
    for (uint32_t p = 0; p < N; p++) newid[perm[p]] = p;
    out.special.start_unanchored_id = next_avail - 2;
    out.special.start_anchored_id = next_avail - 1;
    out.special.max_match_id = next_avail - 3;
    if (has_list(START_A)) out.special.max_match_id = out.special.start_anchored_id;
    out.special.max_special_id = out.special.max_match_id;  // no prefilter is ever built (:1043-1045)

    // ---- 5. final arrays in the new numbering ----
    out.fail.resize(N); out.depth.resize(N);
    out.toff.assign(N + 1, 0); out.moff.assign(N + 1, 0);
    size_t ttotal = 0, mtotal = 0;
    for (uint32_t p = 0; p < N; p++) {
        uint32_t oid = perm[p];
        size_t nt = (oid == kDead || oid == START_U || oid == START_A) ? 256 : (oid == kFail ? 0 : coff[oid + 1] - coff[oid]);
        ttotal += nt;
        mtotal += has_list(oid) ? lists[size_t(lidx[oid])].size() : 0;
    }
    out.tbyte.resize(ttotal); out.tnext.resize(ttotal); out.mpid.resize(mtotal);
    size_t tp = 0, mp = 0;
    for (uint32_t p = 0; p < N; p++) {
        const uint32_t oid = perm[p];
        out.fail[p] = newid[fail[oid]];
        out.depth[p] = depth[oid];
        out.toff[p] = uint32_t(tp);
        out.moff[p] = uint32_t(mp);
        if (oid == kDead) {
            for (int b = 0; b < 256; b++) { out.tbyte[tp] = uint8_t(b); out.tnext[tp++] = kDead; }
        } else if (oid == START_U || oid == START_A) {
            for (int b = 0; b < 256; b++) {
                uint32_t c = emap.find(START_U, uint8_t(b));
                uint32_t t;
                if (c != kFail) t = newid[c];
                else if (oid == START_A) t = kFail;                 // :1561-1576: copy made before the self loop exists
                else t = close_loop ? kDead : newid[START_U];       // :1597-1606, :1620-1638
                out.tbyte[tp] = uint8_t(b); out.tnext[tp++] = t;
            }
        } else if (oid != kFail) {
            for (uint32_t k = coff[oid]; k < coff[oid + 1]; k++) { out.tbyte[tp] = cbyte[k]; out.tnext[tp++] = newid[cnext[k]]; }
        }
        if (has_list(oid)) for (uint32_t pid : lists[size_t(lidx[oid])]) out.mpid[mp++] = pid;
    }
    out.toff[N] = uint32_t(tp);
    out.moff[N] = uint32_t(mp);
    out.bfs.clear();
    out.bfs.reserve(queue.size() + 2);
    out.bfs.push_back(newid[START_U]);
    out.bfs.push_back(newid[START_A]);
    for (uint32_t q : queue) out.bfs.push_back(newid[q]);
    // densify() only affects memory_usage (:1500-1526)
    out.dense_states = 0;
    for (uint32_t oid = 2; oid < N; oid++) if (size_t(depth[oid]) < o.nnfa_dense_depth) out.dense_states++;
    return ACGPU_OK;
}

// ---------------------------------------------------------------------------- DFA
namespace {
// unanchored/anchored rows over classes, ids are nNFA ids (not premultiplied)
void fill_rows(const NNfa& n, const uint8_t* classes, size_t alen, bool anchored_rows, std::vector<uint32_t>& rows) {
    const size_t N = n.states();
    rows.assign(N * alen, kDead);
    bool rep[256];
    for (int b = 0; b < 256; b++) rep[b] = (b == 0) || classes[b] != classes[b - 1];
    for (uint32_t s : n.bfs) {
        uint32_t* row = &rows[size_t(s) * alen];
        const uint32_t f = n.fail[s];
        if (!anchored_rows && f != kDead) std::memcpy(row, &rows[size_t(f) * alen], alen * sizeof(uint32_t));
        for (uint32_t k = n.toff[s]; k < n.toff[s + 1]; k++) {
            const uint8_t b = n.tbyte[k];
            if (!rep[b]) continue;                 // dfa.rs:801-835: one call per class, on its first byte
            const uint32_t t = n.tnext[k];
            if (t == kFail) { if (anchored_rows || f == kDead) row[classes[b]] = kDead; }
            else row[classes[b]] = t;
        }
    }
}
}  // namespace

acgpu_status build_dfa(const NNfa& n, int start_kind, bool byte_classes, Dfa& d, DfaRowFill fill) {
    d = Dfa();
    if (byte_classes) std::memcpy(d.byte_classes, n.byte_classes, 256);
    else for (int i = 0; i < 256; i++) d.byte_classes[i] = uint8_t(i);
    const size_t N = n.states();
    const size_t alen = size_t(d.byte_classes[255]) + 1;
    const size_t s2 = stride2_of(alen), stride = size_t(1) << s2;
    const bool both = start_kind == ACGPU_START_BOTH;
    const size_t state_len = both ? N * 2 - 4 : N;
    if (state_len > (SIZE_MAX >> s2)) return ACGPU_ERR_STATE_ID_OVERFLOW;
    const size_t trans_len = state_len << s2;
    if (trans_len - stride > kSmallIndexMax) return ACGPU_ERR_STATE_ID_OVERFLOW;  // dfa.rs:469-478
    d.state_len = state_len; d.alphabet_len = alen; d.stride2 = s2;
    d.num_match_states = (size_t(n.special.max_match_id) - 1) * (both ? 2 : 1);
    d.trans.assign(trans_len, kDead);
    d.moff.assign(d.num_match_states + 1, 0);

    std::vector<uint32_t> urows, arows;
    if (!both) {
        const bool anchored = start_kind == ACGPU_START_ANCHORED;
        if (fill) {
            if (!fill(n, d.byte_classes, alen, s2, anchored, d.trans.data())) return ACGPU_ERR_HIP;
        } else {
            fill_rows(n, d.byte_classes, alen, anchored, urows);
            for (size_t s = 0; s < N; s++)
                for (size_t k = 0; k < alen; k++) d.trans[(s << s2) + k] = urows[s * alen + k] << s2;
        }
        // matches: DFA state index == nNFA id (dfa.rs:553-560)
        for (uint32_t s = 2; s <= n.special.max_match_id; s++) d.moff[s - 2 + 1] = n.moff[s + 1] - n.moff[s];
        for (size_t i = 0; i < d.num_match_states; i++) d.moff[i + 1] += d.moff[i];
        d.mpid.resize(d.moff[d.num_match_states]);
        for (uint32_t s = 2; s <= n.special.max_match_id; s++)
            std::copy(n.mpid.begin() + n.moff[s], n.mpid.begin() + n.moff[s + 1], d.mpid.begin() + d.moff[s - 2]);
        d.special.max_special_id = n.special.max_special_id << s2;
        d.special.max_match_id = n.special.max_match_id << s2;
        d.special.start_unanchored_id = anchored ? kDead : (n.special.start_unanchored_id << s2);
        d.special.start_anchored_id = anchored ? (n.special.start_anchored_id << s2) : kDead;
        return ACGPU_OK;
    }
    // StartKind::Both (dfa.rs:617-724): ordinary states get an unanchored and an anchored copy
    if (fill) {   // rows computed on the device (premultiplied, state index == nNFA id); the interleave below stays here
        std::vector<uint32_t> tmp(N << s2);
        auto rows_from = [&](bool anchored, std::vector<uint32_t>& rows) {
            if (!fill(n, d.byte_classes, alen, s2, anchored, tmp.data())) return false;
            rows.resize(N * alen);
            for (size_t s = 0; s < N; s++)
                for (size_t k = 0; k < alen; k++) rows[s * alen + k] = tmp[(s << s2) + k] >> s2;
            return true;
        };
        if (!rows_from(false, urows) || !rows_from(true, arows)) return ACGPU_ERR_HIP;
    } else {
        fill_rows(n, d.byte_classes, alen, false, urows);
        fill_rows(n, d.byte_classes, alen, true, arows);
    }
    std::vector<uint32_t> remap_u(N, kDead), remap_a(N, kDead);
    const uint32_t su = n.special.start_unanchored_id, sa = n.special.start_anchored_id;
    uint32_t newsid = 0;
    for (uint32_t s = 0; s < N; s++) {
        if (s == kDead || s == kFail) { remap_u[s] = remap_a[s] = newsid; newsid += uint32_t(stride); }
        else if (s == su) { remap_u[s] = newsid; remap_a[s] = kDead; newsid += uint32_t(stride); }
        else if (s == sa) { remap_u[s] = kDead; remap_a[s] = newsid; newsid += uint32_t(stride); }
        else { remap_u[s] = newsid; newsid += uint32_t(stride); remap_a[s] = newsid; newsid += uint32_t(stride); }
    }
    std::vector<uint32_t> mcount(d.num_match_states, 0);
    auto set_matches = [&](uint32_t nsid, uint32_t s, bool count_only) {
        size_t idx = (nsid >> s2) - 2;
        if (count_only) mcount[idx] = n.moff[s + 1] - n.moff[s];
        else std::copy(n.mpid.begin() + n.moff[s], n.mpid.begin() + n.moff[s + 1], d.mpid.begin() + d.moff[idx]);
    };
    for (int pass = 0; pass < 2; pass++) {
        for (uint32_t s = 2; s < N; s++) {
            if (!n.is_match(s)) continue;
            if (s == su) set_matches(remap_u[s], s, pass == 0);
            else if (s == sa) set_matches(remap_a[s], s, pass == 0);
            else { set_matches(remap_u[s], s, pass == 0); set_matches(remap_a[s], s, pass == 0); }
        }
        if (pass == 0) {
            for (size_t i = 0; i < d.num_match_states; i++) d.moff[i + 1] = d.moff[i] + mcount[i];
            d.mpid.resize(d.moff[d.num_match_states]);
        }
    }
    for (uint32_t s = 2; s < N; s++) {
        if (s == su) {
            for (size_t k = 0; k < alen; k++) d.trans[remap_u[s] + k] = remap_u[urows[size_t(s) * alen + k]];
        } else if (s == sa) {
            for (size_t k = 0; k < alen; k++) d.trans[remap_a[s] + k] = remap_a[arows[size_t(s) * alen + k]];
        } else {
            for (size_t k = 0; k < alen; k++) {
                d.trans[remap_u[s] + k] = remap_u[urows[size_t(s) * alen + k]];
                d.trans[remap_a[s] + k] = remap_a[arows[size_t(s) * alen + k]];
            }
        }
    }
    d.special.max_special_id = remap_a[n.special.max_special_id];
    d.special.max_match_id = remap_a[n.special.max_match_id];
    d.special.start_unanchored_id = remap_u[su];
    d.special.start_anchored_id = remap_a[sa];
    return ACGPU_OK;
}

// ------------------------------------------------------------- contiguous NFA
namespace {
constexpr uint32_t kKindDense = 0xFF;           // contiguous.rs:455
constexpr uint32_t kKindOne = 0xFE;             // contiguous.rs:461
constexpr size_t kMaxSparseTransitions = 127;   // contiguous.rs:479
inline size_t u32_len(size_t ntrans) { return (ntrans + 3) >> 2; }
}  // namespace

acgpu_status build_cnfa(const NNfa& n, size_t dense_depth, bool byte_classes, CNfa& c) {
    c = CNfa();
    if (byte_classes) std::memcpy(c.byte_classes, n.byte_classes, 256);
    else for (int i = 0; i < 256; i++) c.byte_classes[i] = uint8_t(i);
    const size_t alen = size_t(c.byte_classes[255]) + 1;
    const size_t N = n.states();
    c.alphabet_len = alen; c.state_len = N;
    std::vector<uint32_t> old2new(N, kDead);
    std::vector<uint32_t>& r = c.repr;
    for (uint32_t s = 0; s < N; s++) {
        if (s == kFail) { old2new[s] = kFail; continue; }
        if (r.size() > kSmallIndexMax) return ACGPU_ERR_STATE_ID_OVERFLOW;
        old2new[s] = uint32_t(r.size());
        const size_t nt = n.toff[s + 1] - n.toff[s];
        const bool is_match = n.is_match(s);
        const bool force_dense = size_t(n.depth[s]) < dense_depth;
        if (force_dense || nt > kMaxSparseTransitions) {
            r.push_back(kKindDense);
            r.push_back(n.fail[s]);
            const size_t base = r.size();
            r.resize(base + alen, kFail);
            for (uint32_t k = n.toff[s]; k < n.toff[s + 1]; k++) r[base + c.byte_classes[n.tbyte[k]]] = n.tnext[k];
        } else if (nt == 1 && !is_match) {
            const uint32_t k = n.toff[s];
            r.push_back(kKindOne | (uint32_t(c.byte_classes[n.tbyte[k]]) << 8));
            r.push_back(n.fail[s]);
            r.push_back(n.tnext[k]);
        } else {
            r.push_back(uint32_t(nt));
            r.push_back(n.fail[s]);
            // class bytes packed 4 per word in memory order, last class repeated as padding (:749-784)
            const size_t words = u32_len(nt);
            for (size_t w = 0; w < words; w++) {
                uint8_t chunk[4];
                for (size_t j = 0; j < 4; j++) {
                    size_t idx = std::min(w * 4 + j, nt - 1);
                    chunk[j] = c.byte_classes[n.tbyte[n.toff[s] + idx]];
                }
                uint32_t word;
                std::memcpy(&word, chunk, 4);
                r.push_back(word);
            }
            for (uint32_t k = n.toff[s]; k < n.toff[s + 1]; k++) r.push_back(n.tnext[k]);
        }
        if (is_match) {
            const uint32_t ml = n.moff[s + 1] - n.moff[s];
            if (ml == 1) r.push_back((1u << 31) | n.mpid[n.moff[s]]);
            else {
                r.push_back(ml);
                for (uint32_t k = n.moff[s]; k < n.moff[s + 1]; k++) r.push_back(n.mpid[k]);
            }
        }
    }
    // second pass: nNFA ids -> word offsets (:980-986, :486-509)
    for (uint32_t s = 0; s < N; s++) {
        if (s == kFail) continue;
        uint32_t* st = &r[old2new[s]];
        const uint32_t kind = st[0] & 0xFF;
        st[1] = old2new[st[1]];
        if (kind == kKindDense) for (size_t k = 0; k < alen; k++) st[2 + k] = old2new[st[2 + k]];
        else if (kind == kKindOne) st[2] = old2new[st[2]];
        else { size_t tl = kind, cl = u32_len(tl); for (size_t k = 0; k < tl; k++) st[2 + cl + k] = old2new[st[2 + cl + k]]; }
    }
    c.special.max_special_id = old2new[n.special.max_special_id];
    c.special.max_match_id = old2new[n.special.max_match_id];
    c.special.start_unanchored_id = old2new[n.special.start_unanchored_id];
    c.special.start_anchored_id = old2new[n.special.start_anchored_id];
    return ACGPU_OK;
}

}  // namespace acgpu
