/*
 * ac_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see ac_oracle.h).
 *
 * Literal plain-C restatement of BurntSushi/aho-corasick 1.1.3:
 *   - noncontiguous NFA compiler      src/nfa/noncontiguous.rs:963-1646
 *   - state shuffle / remapper        src/util/remapper.rs:67-154
 *   - byte classes                    src/util/alphabet.rs:224-250
 *   - DFA builder                     src/dfa.rs:431-835
 *   - contiguous NFA builder          src/nfa/contiguous.rs:686-1009
 *   - the three next_state bodies     dfa.rs:218-226, contiguous.rs:186-247,
 *                                     noncontiguous.rs:601-626
 *   - search loops + iterators        src/automaton.rs:857-970, 1259-1549
 *   - facade guards / kind selection  src/ahocorasick.rs:2171-2261, 2778-2789
 *
 * Prefilters (src/util/prefilter.rs) are results-neutral skip-ahead and are
 * NOT restated: max_special_id == max_match_id always
 * (noncontiguous.rs:1036-1045, the "no prefilter" arm).
 *
 * Data structures deliberately mirror the reference (index-0 sentinels,
 * linked lists inside vectors) so that state numbering, match-list order and
 * every table word come out identical.
 */
#include "ac_oracle.h"

#include <stdlib.h>
#include <pthread.h>
#include <string.h>

#define DEAD 0u
#define FAIL 1u
/* StateID::MAX = i32::MAX - 1 (src/util/primitives.rs:95-111) */
#define SMALLINDEX_MAX 0x7FFFFFFEu

/* ------------------------------------------------------------------ vectors */
#define VEC(T) struct { T* p; size_t n, cap; }
typedef struct { uint32_t* p; size_t n, cap; } u32vec;

#define VEC_PUSH(v, val, okflag)                                              \
    do {                                                                      \
        if ((v).n == (v).cap) {                                               \
            size_t nc_ = (v).cap ? (v).cap * 2 : 64;                          \
            void* np_ = realloc((v).p, nc_ * sizeof(*(v).p));                 \
            if (!np_) { (okflag) = 0; break; }                                \
            (v).p = np_; (v).cap = nc_;                                       \
        }                                                                     \
        (v).p[(v).n++] = (val);                                               \
    } while (0)

/* noncontiguous.rs:710-748 */
typedef struct { uint32_t sparse, dense, matches, fail, depth; } nstate;
/* noncontiguous.rs:769-775 */
typedef struct { uint8_t byte; uint32_t next, link; } ntrans;
/* noncontiguous.rs:807-811 */
typedef struct { uint32_t pid, link; } nmatch;
/* src/util/special.rs:10-28 */
typedef struct { uint32_t max_special_id, max_match_id, start_unanchored_id, start_anchored_id; } special_t;

typedef struct {
    int match_kind;
    VEC(nstate) states;
    VEC(ntrans) sparse;
    VEC(uint32_t) dense;
    VEC(nmatch) matches;
    VEC(uint32_t) pattern_lens;
    uint8_t byte_classes[256];
    size_t min_pattern_len, max_pattern_len;
    special_t special;
} nnfa_t;

typedef struct {
    uint32_t* trans; size_t trans_len;
    uint32_t* match_off; uint32_t* match_pid; size_t num_match_states;
    size_t state_len, alphabet_len, stride2;
    uint8_t byte_classes[256];
    special_t special;
} dfa_t;

typedef struct {
    VEC(uint32_t) repr;
    size_t state_len, alphabet_len;
    uint8_t byte_classes[256];
    special_t special;
} cnfa_t;

struct orc_ac {
    int kind, start_kind, match_kind;
    nnfa_t nnfa;
    dfa_t dfa;
    cnfa_t cnfa;
};

/* ------------------------------------------------------- byte class helpers */
/* alphabet.rs:45-49 */
static size_t alphabet_len(const uint8_t* classes) { return (size_t)classes[255] + 1; }
/* alphabet.rs:59-62 */
static size_t stride2_of(const uint8_t* classes) {
    size_t a = alphabet_len(classes), p = 1, z = 0;
    while (p < a) { p <<= 1; z++; }
    return z;
}
/* contiguous.rs:1079-1085 / noncontiguous.rs opposite_ascii_case (util) */
static uint8_t opposite_ascii_case(uint8_t b) {
    if (b >= 'A' && b <= 'Z') return (uint8_t)(b + 32);
    if (b >= 'a' && b <= 'z') return (uint8_t)(b - 32);
    return b;
}

/* --------------------------------------------------------- nNFA primitives */
/* noncontiguous.rs:527-533 */
static int alloc_transition(nnfa_t* n, uint32_t* id) {
    int ok = 1;
    if (n->sparse.n > SMALLINDEX_MAX) return ORC_ERR_STATE_ID_OVERFLOW;
    *id = (uint32_t)n->sparse.n;
    ntrans t = {0, 0, 0};
    VEC_PUSH(n->sparse, t, ok);
    return ok ? ORC_OK : ORC_ERR_NOMEM;
}
/* noncontiguous.rs:537-543 */
static int alloc_match(nnfa_t* n, uint32_t* id) {
    int ok = 1;
    if (n->matches.n > SMALLINDEX_MAX) return ORC_ERR_STATE_ID_OVERFLOW;
    *id = (uint32_t)n->matches.n;
    nmatch m = {0, 0};
    VEC_PUSH(n->matches, m, ok);
    return ok ? ORC_OK : ORC_ERR_NOMEM;
}
/* noncontiguous.rs:551-563 */
static int alloc_dense_state(nnfa_t* n, uint32_t* id) {
    int ok = 1;
    if (n->dense.n > SMALLINDEX_MAX) return ORC_ERR_STATE_ID_OVERFLOW;
    *id = (uint32_t)n->dense.n;
    size_t a = alphabet_len(n->byte_classes);
    for (size_t i = 0; i < a && ok; i++) VEC_PUSH(n->dense, FAIL, ok);
    return ok ? ORC_OK : ORC_ERR_NOMEM;
}
/* noncontiguous.rs:568-585 */
static int alloc_state(nnfa_t* n, size_t depth, uint32_t* id) {
    int ok = 1;
    if (n->states.n > SMALLINDEX_MAX) return ORC_ERR_STATE_ID_OVERFLOW;
    *id = (uint32_t)n->states.n;
    nstate s = {0, 0, 0, n->special.start_unanchored_id, (uint32_t)depth};
    VEC_PUSH(n->states, s, ok);
    return ok ? ORC_OK : ORC_ERR_NOMEM;
}

/* noncontiguous.rs:364-374 */
static inline uint32_t follow_transition_sparse(const nnfa_t* n, uint32_t sid, uint8_t byte) {
    uint32_t link = n->states.p[sid].sparse;
    while (link != 0) {
        const ntrans* t = &n->sparse.p[link];
        if (byte <= t->byte) {
            if (byte == t->byte) return t->next;
            break;
        }
        link = t->link;
    }
    return FAIL;
}
/* noncontiguous.rs:339-360 */
static inline uint32_t follow_transition(const nnfa_t* n, uint32_t sid, uint8_t byte) {
    const nstate* s = &n->states.p[sid];
    if (s->dense == 0) return follow_transition_sparse(n, sid, byte);
    return n->dense.p[s->dense + n->byte_classes[byte]];
}
/* noncontiguous.rs:601-626 */
static inline uint32_t nnfa_next_state(const nnfa_t* n, int anchored, uint32_t sid, uint8_t byte) {
    for (;;) {
        uint32_t next = follow_transition(n, sid, byte);
        if (next != FAIL) return next;
        if (anchored) return DEAD;
        sid = n->states.p[sid].fail;
    }
}

/* noncontiguous.rs:381-423 */
static int add_transition(nnfa_t* n, uint32_t prev, uint8_t byte, uint32_t next) {
    int rc;
    if (n->states.p[prev].dense != 0) {
        uint32_t dense = n->states.p[prev].dense;
        n->dense.p[dense + n->byte_classes[byte]] = next;
    }
    uint32_t head = n->states.p[prev].sparse;
    if (head == 0 || byte < n->sparse.p[head].byte) {
        uint32_t nl;
        if ((rc = alloc_transition(n, &nl))) return rc;
        n->sparse.p[nl].byte = byte; n->sparse.p[nl].next = next; n->sparse.p[nl].link = head;
        n->states.p[prev].sparse = nl;
        return ORC_OK;
    } else if (byte == n->sparse.p[head].byte) {
        n->sparse.p[head].next = next;
        return ORC_OK;
    }
    uint32_t link_prev = head, link_next = n->sparse.p[head].link;
    while (link_next != 0 && byte > n->sparse.p[link_next].byte) {
        link_prev = link_next;
        link_next = n->sparse.p[link_next].link;
    }
    if (link_next == 0 || byte < n->sparse.p[link_next].byte) {
        uint32_t link;
        if ((rc = alloc_transition(n, &link))) return rc;
        n->sparse.p[link].byte = byte; n->sparse.p[link].next = next; n->sparse.p[link].link = link_next;
        n->sparse.p[link_prev].link = link;
    } else {
        n->sparse.p[link_next].next = next;
    }
    return ORC_OK;
}

/* noncontiguous.rs:435-463 */
static int init_full_state(nnfa_t* n, uint32_t prev, uint32_t next) {
    int rc;
    uint32_t prev_link = 0;
    for (int byte = 0; byte <= 255; byte++) {
        uint32_t nl;
        if ((rc = alloc_transition(n, &nl))) return rc;
        n->sparse.p[nl].byte = (uint8_t)byte; n->sparse.p[nl].next = next; n->sparse.p[nl].link = 0;
        if (prev_link == 0) n->states.p[prev].sparse = nl;
        else n->sparse.p[prev_link].link = nl;
        prev_link = nl;
    }
    return ORC_OK;
}

/* noncontiguous.rs:466-484 */
static int add_match(nnfa_t* n, uint32_t sid, uint32_t pid) {
    int rc;
    uint32_t head = n->states.p[sid].matches;
    uint32_t link = head;
    while (n->matches.p[link].link != 0) link = n->matches.p[link].link;
    uint32_t nm;
    if ((rc = alloc_match(n, &nm))) return rc;
    n->matches.p[nm].pid = pid;
    if (link == 0) n->states.p[sid].matches = nm;
    else n->matches.p[link].link = nm;
    return ORC_OK;
}

/* noncontiguous.rs:490-523 */
static int copy_matches(nnfa_t* n, uint32_t src, uint32_t dst) {
    int ok = 1;
    uint32_t head_dst = n->states.p[dst].matches;
    uint32_t link_dst = head_dst;
    while (n->matches.p[link_dst].link != 0) link_dst = n->matches.p[link_dst].link;
    uint32_t link_src = n->states.p[src].matches;
    while (link_src != 0) {
        if (n->matches.n > SMALLINDEX_MAX) return ORC_ERR_STATE_ID_OVERFLOW;
        uint32_t nm = (uint32_t)n->matches.n;
        nmatch m = {n->matches.p[link_src].pid, 0};
        VEC_PUSH(n->matches, m, ok);
        if (!ok) return ORC_ERR_NOMEM;
        if (link_dst == 0) n->states.p[dst].matches = nm;
        else n->matches.p[link_dst].link = nm;
        link_dst = nm;
        link_src = n->matches.p[link_src].link;
    }
    return ORC_OK;
}

static inline int state_is_match(const nnfa_t* n, uint32_t sid) { return n->states.p[sid].matches != 0; }

/* ----------------------------------------------------------- byte class set */
/* alphabet.rs:224-230: boundaries at start-1 and end */
static void byteset_set_range(uint8_t* set /*[256] 0/1*/, uint8_t start, uint8_t end) {
    if (start > 0) set[start - 1] = 1;
    set[end] = 1;
}
/* alphabet.rs:235-250 */
static void byteset_byte_classes(const uint8_t* set, uint8_t* classes) {
    uint8_t cls = 0;
    int b = 0;
    for (;;) {
        classes[b] = cls;
        if (b == 255) break;
        if (set[b]) cls++;
        b++;
    }
}

/* ------------------------------------------------------------- the compiler */
typedef struct {
    int match_kind, ascii_case_insensitive;
    size_t dense_depth;
} nnfa_builder;

/* noncontiguous.rs:1057-1150 */
static int build_trie(nnfa_t* n, const nnfa_builder* b, uint8_t* byteset,
                      const uint8_t* const* pats, const size_t* lens, size_t npats) {
    int rc, ok = 1;
    for (size_t i = 0; i < npats; i++) {
        if (i > SMALLINDEX_MAX) return ORC_ERR_PATTERN_ID_OVERFLOW;
        uint32_t pid = (uint32_t)i;
        const uint8_t* pat = pats[i];
        size_t plen = lens[i];
        if (plen > SMALLINDEX_MAX) return ORC_ERR_PATTERN_TOO_LONG;
        if (plen < n->min_pattern_len) n->min_pattern_len = plen;
        if (plen > n->max_pattern_len) n->max_pattern_len = plen;
        VEC_PUSH(n->pattern_lens, (uint32_t)plen, ok);
        if (!ok) return ORC_ERR_NOMEM;

        uint32_t prev = n->special.start_unanchored_id;
        int saw_match = 0, skip = 0;
        for (size_t depth = 0; depth < plen; depth++) {
            uint8_t by = pat[depth];
            saw_match = saw_match || state_is_match(n, prev);
            if (b->match_kind == ORC_LEFTMOST_FIRST && saw_match) { skip = 1; break; }
            byteset_set_range(byteset, by, by);
            if (b->ascii_case_insensitive) {
                uint8_t ob = opposite_ascii_case(by);
                byteset_set_range(byteset, ob, ob);
            }
            uint32_t next = follow_transition(n, prev, by);
            if (next != FAIL) {
                prev = next;
            } else {
                if ((rc = alloc_state(n, depth, &next))) return rc;
                if ((rc = add_transition(n, prev, by, next))) return rc;
                if (b->ascii_case_insensitive) {
                    uint8_t ob = opposite_ascii_case(by);
                    if ((rc = add_transition(n, prev, ob, next))) return rc;
                }
                prev = next;
            }
        }
        if (skip) continue; /* continue 'PATTERNS */
        if ((rc = add_match(n, prev, pid))) return rc;
    }
    return ORC_OK;
}

/* noncontiguous.rs:1561-1586 */
static int set_anchored_start_state(nnfa_t* n) {
    uint32_t su = n->special.start_unanchored_id, sa = n->special.start_anchored_id;
    uint32_t ul = n->states.p[su].sparse, al = n->states.p[sa].sparse;
    while (ul != 0 && al != 0) {
        n->sparse.p[al].next = n->sparse.p[ul].next;
        ul = n->sparse.p[ul].link;
        al = n->sparse.p[al].link;
    }
    int rc = copy_matches(n, su, sa);
    if (rc) return rc;
    n->states.p[sa].fail = DEAD;
    return ORC_OK;
}

/* noncontiguous.rs:1597-1606 */
static void add_unanchored_start_state_loop(nnfa_t* n) {
    uint32_t su = n->special.start_unanchored_id;
    for (uint32_t link = n->states.p[su].sparse; link != 0; link = n->sparse.p[link].link)
        if (n->sparse.p[link].next == FAIL) n->sparse.p[link].next = su;
}

/* noncontiguous.rs:1500-1526 */
static int densify(nnfa_t* n, const nnfa_builder* b) {
    for (size_t i = 0; i < n->states.n; i++) {
        uint32_t sid = (uint32_t)i;
        if (sid == DEAD || sid == FAIL) continue;
        if ((size_t)n->states.p[sid].depth >= b->dense_depth) continue;
        uint32_t dense;
        int rc = alloc_dense_state(n, &dense);
        if (rc) return rc;
        for (uint32_t link = n->states.p[sid].sparse; link != 0; link = n->sparse.p[link].link) {
            const ntrans* t = &n->sparse.p[link];
            n->dense.p[dense + n->byte_classes[t->byte]] = t->next;
        }
        n->states.p[sid].dense = dense;
    }
    return ORC_OK;
}

/* noncontiguous.rs:1275-1374 (QueuedSet :1657-1689 is a bitmap here; it is
 * active only when ascii_case_insensitive, inert otherwise) */
static int fill_failure_transitions(nnfa_t* n, const nnfa_builder* b) {
    int rc = ORC_OK;
    int is_leftmost = b->match_kind != ORC_STANDARD;
    uint32_t start_uid = n->special.start_unanchored_id;
    size_t ns = n->states.n;
    /* without casei every non-start state is enqueued exactly once (one parent
     * edge per trie state), with casei the seen-set de-duplicates */
    uint32_t* queue = malloc((ns + 1) * sizeof(uint32_t));
    uint8_t* seen = b->ascii_case_insensitive ? calloc(ns, 1) : NULL;
    if (!queue || (b->ascii_case_insensitive && !seen)) { free(queue); free(seen); return ORC_ERR_NOMEM; }
    size_t qh = 0, qt = 0;

    for (uint32_t link = n->states.p[start_uid].sparse; link != 0; link = n->sparse.p[link].link) {
        ntrans t = n->sparse.p[link];
        if (start_uid == t.next || (seen && seen[t.next])) continue;
        queue[qt++] = t.next;
        if (seen) seen[t.next] = 1;
        if (is_leftmost && state_is_match(n, t.next)) n->states.p[t.next].fail = DEAD;
    }
    while (qh < qt) {
        uint32_t id = queue[qh++];
        for (uint32_t link = n->states.p[id].sparse; link != 0; link = n->sparse.p[link].link) {
            ntrans t = n->sparse.p[link];
            if (seen && seen[t.next]) continue;
            queue[qt++] = t.next;
            if (seen) seen[t.next] = 1;
            if (is_leftmost && state_is_match(n, t.next)) {
                n->states.p[t.next].fail = DEAD;
                continue;
            }
            uint32_t fail = n->states.p[id].fail;
            while (follow_transition(n, fail, t.byte) == FAIL) fail = n->states.p[fail].fail;
            fail = follow_transition(n, fail, t.byte);
            n->states.p[t.next].fail = fail;
            if ((rc = copy_matches(n, fail, t.next))) goto done;
        }
        if (!is_leftmost) {
            if ((rc = copy_matches(n, n->special.start_unanchored_id, id))) goto done;
        }
    }
done:
    free(queue);
    free(seen);
    return rc;
}

/* noncontiguous.rs:1620-1638 */
static void close_start_state_loop_for_leftmost(nnfa_t* n, const nnfa_builder* b) {
    uint32_t su = n->special.start_unanchored_id;
    uint32_t dense = n->states.p[su].dense;
    if (b->match_kind != ORC_STANDARD && state_is_match(n, su)) {
        for (uint32_t link = n->states.p[su].sparse; link != 0; link = n->sparse.p[link].link) {
            if (n->sparse.p[link].next == su) {
                n->sparse.p[link].next = DEAD;
                if (dense != 0) n->dense.p[dense + n->byte_classes[n->sparse.p[link].byte]] = DEAD;
            }
        }
    }
}

/* remapper.rs:104-115 */
static void remapper_swap(nnfa_t* n, uint32_t* map, uint32_t id1, uint32_t id2) {
    if (id1 == id2) return;
    nstate tmp = n->states.p[id1]; n->states.p[id1] = n->states.p[id2]; n->states.p[id2] = tmp;
    uint32_t t = map[id1]; map[id1] = map[id2]; map[id2] = t;
}

/* noncontiguous.rs:1399-1481 + remapper.rs:119-154 + noncontiguous.rs:261-278 */
static int shuffle(nnfa_t* n) {
    uint32_t old_su = n->special.start_unanchored_id, old_sa = n->special.start_anchored_id;
    size_t ns = n->states.n;
    uint32_t* map = malloc(ns * sizeof(uint32_t));
    uint32_t* oldmap = malloc(ns * sizeof(uint32_t));
    if (!map || !oldmap) { free(map); free(oldmap); return ORC_ERR_NOMEM; }
    for (size_t i = 0; i < ns; i++) map[i] = (uint32_t)i;
    uint32_t next_avail = 4;
    for (size_t i = 4; i < ns; i++) {
        uint32_t sid = (uint32_t)i;
        if (!state_is_match(n, sid)) continue;
        remapper_swap(n, map, sid, next_avail);
        next_avail++;
    }
    uint32_t new_sa = next_avail - 1;
    remapper_swap(n, map, old_sa, new_sa);
    uint32_t new_su = next_avail - 2;
    remapper_swap(n, map, old_su, new_su);
    n->special.max_match_id = next_avail - 3;
    n->special.start_unanchored_id = new_su;
    n->special.start_anchored_id = new_sa;
    if (state_is_match(n, n->special.start_anchored_id)) n->special.max_match_id = n->special.start_anchored_id;
    /* Remapper::remap */
    memcpy(oldmap, map, ns * sizeof(uint32_t));
    for (size_t i = 0; i < ns; i++) {
        uint32_t cur_id = (uint32_t)i;
        uint32_t new_id = oldmap[i];
        if (cur_id == new_id) continue;
        for (;;) {
            uint32_t id = oldmap[new_id];
            if (cur_id == id) { map[i] = new_id; break; }
            new_id = id;
        }
    }
    /* NFA::remap */
    size_t a = alphabet_len(n->byte_classes);
    for (size_t i = 0; i < ns; i++) {
        nstate* s = &n->states.p[i];
        s->fail = map[s->fail];
        for (uint32_t link = s->sparse; link != 0; link = n->sparse.p[link].link)
            n->sparse.p[link].next = map[n->sparse.p[link].next];
        if (s->dense != 0)
            for (size_t k = 0; k < a; k++) n->dense.p[s->dense + k] = map[n->dense.p[s->dense + k]];
    }
    free(map); free(oldmap);
    return ORC_OK;
}

/* noncontiguous.rs:963-1051 */
static int nnfa_compile(nnfa_t* n, const nnfa_builder* b, const uint8_t* const* pats,
                        const size_t* lens, size_t npats) {
    int rc, ok = 1;
    uint8_t byteset[256];
    memset(byteset, 0, sizeof byteset);
    memset(n, 0, sizeof *n);
    n->match_kind = b->match_kind;
    for (int i = 0; i < 256; i++) n->byte_classes[i] = (uint8_t)i; /* singletons */
    n->min_pattern_len = (size_t)-1;
    n->max_pattern_len = 0;
    ntrans t0 = {0, 0, 0}; nmatch m0 = {0, 0};
    VEC_PUSH(n->sparse, t0, ok);
    VEC_PUSH(n->matches, m0, ok);
    VEC_PUSH(n->dense, DEAD, ok);
    if (!ok) return ORC_ERR_NOMEM;
    uint32_t id;
    if ((rc = alloc_state(n, 0, &id))) return rc;                    /* DEAD */
    if ((rc = alloc_state(n, 0, &id))) return rc;                    /* FAIL */
    /* NB: alloc_state reads special.start_unanchored_id *before* it is set
     * for START_U itself, so DEAD/FAIL/START_U get fail=0 (noncontiguous.rs:577-583) */
    if ((rc = alloc_state(n, 0, &id))) return rc;
    n->special.start_unanchored_id = id;
    if ((rc = alloc_state(n, 0, &id))) return rc;
    n->special.start_anchored_id = id;
    /* init_unanchored_start_state :1549-1555 */
    if ((rc = init_full_state(n, n->special.start_unanchored_id, FAIL))) return rc;
    if ((rc = init_full_state(n, n->special.start_anchored_id, FAIL))) return rc;
    /* add_dead_state_loop :1643-1646 */
    if ((rc = init_full_state(n, DEAD, DEAD))) return rc;
    if ((rc = build_trie(n, b, byteset, pats, lens, npats))) return rc;
    byteset_byte_classes(byteset, n->byte_classes);
    if ((rc = set_anchored_start_state(n))) return rc;
    add_unanchored_start_state_loop(n);
    if ((rc = densify(n, b))) return rc;
    if ((rc = fill_failure_transitions(n, b))) return rc;
    close_start_state_loop_for_leftmost(n, b);
    if ((rc = shuffle(n))) return rc;
    /* no prefilter is ever built by the oracle => :1043-1045 */
    n->special.max_special_id = n->special.max_match_id;
    return ORC_OK;
}

static void nnfa_free(nnfa_t* n) {
    free(n->states.p); free(n->sparse.p); free(n->dense.p); free(n->matches.p); free(n->pattern_lens.p);
    memset(n, 0, sizeof *n);
}

static size_t nnfa_match_len(const nnfa_t* n, uint32_t sid) {
    size_t c = 0;
    for (uint32_t l = n->states.p[sid].matches; l != 0; l = n->matches.p[l].link) c++;
    return c;
}
static uint32_t nnfa_match_pattern(const nnfa_t* n, uint32_t sid, size_t index) {
    uint32_t l = n->states.p[sid].matches;
    while (index--) l = n->matches.p[l].link;
    return n->matches.p[l].pid;
}

/* ------------------------------------------------------------------- DFA */
typedef void (*sparse_iter_fn)(void* ctx, uint8_t byte, uint8_t cls, uint32_t next);

/* dfa.rs:801-835 */
static void sparse_iter(const nnfa_t* n, uint32_t oldsid, const uint8_t* classes,
                        sparse_iter_fn f, void* ctx) {
    int prev_class = -1;
    unsigned byte = 0;
    for (uint32_t link = n->states.p[oldsid].sparse; link != 0; link = n->sparse.p[link].link) {
        const ntrans* t = &n->sparse.p[link];
        while (byte < (unsigned)t->byte) {
            uint8_t rep = (uint8_t)byte;
            uint8_t cls = classes[rep];
            byte++;
            if (prev_class != (int)cls) { f(ctx, rep, cls, FAIL); prev_class = cls; }
        }
        uint8_t rep = t->byte;
        uint8_t cls = classes[rep];
        byte++;
        if (prev_class != (int)cls) { f(ctx, rep, cls, t->next); prev_class = cls; }
    }
    for (unsigned bb = byte; bb <= 255; bb++) {
        uint8_t rep = (uint8_t)bb;
        uint8_t cls = classes[rep];
        if (prev_class != (int)cls) { f(ctx, rep, cls, FAIL); prev_class = cls; }
    }
}

/* dfa.rs:171-184: matches as CSR; lists are appended in ascending DFA state
 * index order, so a two-pass CSR fill reproduces Vec<Vec<PatternID>>. */
typedef struct {
    const nnfa_t* n; dfa_t* d; int anchored; uint32_t newsid; const nstate* state;
} one_start_ctx;

/* dfa.rs:565-591 closure */
static void one_start_cb(void* vctx, uint8_t byte, uint8_t cls, uint32_t oldnext) {
    one_start_ctx* c = vctx;
    if (oldnext == FAIL) {
        if (c->anchored) oldnext = DEAD;
        else if (c->state->fail == DEAD) oldnext = DEAD;
        else oldnext = nnfa_next_state(c->n, 0, c->state->fail, byte);
    }
    c->d->trans[c->newsid + cls] = oldnext << c->d->stride2;
}

static int dfa_set_matches(dfa_t* d, u32vec* pidv, const nnfa_t* n, uint32_t newsid, uint32_t oldsid) {
    /* requires calls in ascending newsid order */
    int ok = 1;
    size_t index = (newsid >> d->stride2) - 2;
    d->match_off[index] = (uint32_t)pidv->n;
    for (uint32_t l = n->states.p[oldsid].matches; l != 0; l = n->matches.p[l].link) {
        VEC_PUSH(*pidv, n->matches.p[l].pid, ok);
        if (!ok) return ORC_ERR_NOMEM;
    }
    d->match_off[index + 1] = (uint32_t)pidv->n;
    return ORC_OK;
}

/* dfa.rs:544-607 */
static int finish_build_one_start(int anchored, const nnfa_t* n, dfa_t* d, u32vec* pidv) {
    size_t stride2 = d->stride2;
    for (size_t i = 0; i < n->states.n; i++) {
        uint32_t oldsid = (uint32_t)i;
        uint32_t newsid = oldsid << stride2;
        const nstate* st = &n->states.p[oldsid];
        if (state_is_match(n, oldsid)) {
            int rc = dfa_set_matches(d, pidv, n, newsid, oldsid);
            if (rc) return rc;
        }
        one_start_ctx ctx = {n, d, anchored, newsid, st};
        sparse_iter(n, oldsid, d->byte_classes, one_start_cb, &ctx);
    }
    d->special.max_special_id = n->special.max_special_id << stride2;
    d->special.max_match_id = n->special.max_match_id << stride2;
    if (anchored) {
        d->special.start_unanchored_id = DEAD;
        d->special.start_anchored_id = n->special.start_anchored_id << stride2;
    } else {
        d->special.start_unanchored_id = n->special.start_unanchored_id << stride2;
        d->special.start_anchored_id = DEAD;
    }
    return ORC_OK;
}

typedef struct {
    const nnfa_t* n; dfa_t* d; uint32_t newsid, unewsid, anewsid; const nstate* state;
} both_ctx;
/* dfa.rs:645-658 closure (start states) */
static void both_start_cb(void* vctx, uint8_t byte, uint8_t cls, uint32_t oldnext) {
    (void)byte;
    both_ctx* c = vctx;
    c->d->trans[c->newsid + cls] = (oldnext == FAIL) ? DEAD : oldnext;
}
/* dfa.rs:675-697 closure (ordinary states) */
static void both_other_cb(void* vctx, uint8_t byte, uint8_t cls, uint32_t oldnext) {
    both_ctx* c = vctx;
    if (oldnext == FAIL) {
        uint32_t nx = (c->state->fail == DEAD) ? DEAD : nnfa_next_state(c->n, 0, c->state->fail, byte);
        c->d->trans[c->unewsid + cls] = nx;
    } else {
        c->d->trans[c->unewsid + cls] = oldnext;
        c->d->trans[c->anewsid + cls] = oldnext;
    }
}

/* dfa.rs:617-724 */
static int finish_build_both_starts(const nnfa_t* n, dfa_t* d, u32vec* pidv) {
    size_t stride2 = d->stride2, stride = (size_t)1 << stride2, ns = n->states.n;
    uint32_t* remap_u = calloc(ns, sizeof(uint32_t));
    uint32_t* remap_a = calloc(ns, sizeof(uint32_t));
    uint8_t* is_anch = calloc(d->state_len, 1);
    if (!remap_u || !remap_a || !is_anch) { free(remap_u); free(remap_a); free(is_anch); return ORC_ERR_NOMEM; }
    int rc = ORC_OK;
    uint32_t newsid = DEAD;
    for (size_t i = 0; i < ns; i++) {
        uint32_t oldsid = (uint32_t)i;
        const nstate* st = &n->states.p[oldsid];
        if (oldsid == DEAD || oldsid == FAIL) {
            remap_u[oldsid] = newsid; remap_a[oldsid] = newsid;
            newsid += (uint32_t)stride;
        } else if (oldsid == n->special.start_unanchored_id || oldsid == n->special.start_anchored_id) {
            if (oldsid == n->special.start_unanchored_id) {
                remap_u[oldsid] = newsid; remap_a[oldsid] = DEAD;
            } else {
                remap_u[oldsid] = DEAD; remap_a[oldsid] = newsid;
                is_anch[newsid >> stride2] = 1;
            }
            if (state_is_match(n, oldsid))
                if ((rc = dfa_set_matches(d, pidv, n, newsid, oldsid))) goto done;
            both_ctx ctx = {n, d, newsid, 0, 0, st};
            sparse_iter(n, oldsid, d->byte_classes, both_start_cb, &ctx);
            newsid += (uint32_t)stride;
        } else {
            uint32_t unew = newsid; newsid += (uint32_t)stride;
            uint32_t anew = newsid; newsid += (uint32_t)stride;
            remap_u[oldsid] = unew; remap_a[oldsid] = anew;
            is_anch[anew >> stride2] = 1;
            if (state_is_match(n, oldsid)) {
                if ((rc = dfa_set_matches(d, pidv, n, unew, oldsid))) goto done;
                if ((rc = dfa_set_matches(d, pidv, n, anew, oldsid))) goto done;
            }
            both_ctx ctx = {n, d, 0, unew, anew, st};
            sparse_iter(n, oldsid, d->byte_classes, both_other_cb, &ctx);
        }
    }
    for (size_t i = 0; i < d->state_len; i++) {
        size_t sid = i << stride2;
        const uint32_t* rm = is_anch[i] ? remap_a : remap_u;
        for (size_t k = 0; k < stride; k++) d->trans[sid + k] = rm[d->trans[sid + k]];
    }
    d->special.max_special_id = remap_a[n->special.max_special_id];
    d->special.max_match_id = remap_a[n->special.max_match_id];
    d->special.start_unanchored_id = remap_u[n->special.start_unanchored_id];
    d->special.start_anchored_id = remap_a[n->special.start_anchored_id];
done:
    free(remap_u); free(remap_a); free(is_anch);
    return rc;
}

/* dfa.rs:431-540 */
static int dfa_build_from_noncontiguous(const nnfa_t* n, int start_kind, int byte_classes, dfa_t* d) {
    memset(d, 0, sizeof *d);
    if (byte_classes) memcpy(d->byte_classes, n->byte_classes, 256);
    else for (int i = 0; i < 256; i++) d->byte_classes[i] = (uint8_t)i;
    size_t state_len = (start_kind == ORC_START_BOTH) ? n->states.n * 2 - 4 : n->states.n;
    size_t s2 = stride2_of(d->byte_classes);
    size_t stride = (size_t)1 << s2;
    if (state_len > (((size_t)-1) >> s2)) return ORC_ERR_STATE_ID_OVERFLOW;
    size_t trans_len = state_len << s2;
    if (trans_len - stride > SMALLINDEX_MAX) return ORC_ERR_STATE_ID_OVERFLOW;
    size_t num_match_states = (size_t)n->special.max_match_id - 1;
    if (start_kind == ORC_START_BOTH) num_match_states *= 2;
    d->trans = calloc(trans_len ? trans_len : 1, sizeof(uint32_t));
    d->match_off = calloc(num_match_states + 1, sizeof(uint32_t));
    if (!d->trans || !d->match_off) return ORC_ERR_NOMEM;
    d->trans_len = trans_len;
    d->num_match_states = num_match_states;
    d->state_len = state_len;
    d->alphabet_len = alphabet_len(d->byte_classes);
    d->stride2 = s2;
    u32vec pidv = {0, 0, 0};
    int rc;
    if (start_kind == ORC_START_BOTH) rc = finish_build_both_starts(n, d, &pidv);
    else rc = finish_build_one_start(start_kind == ORC_START_ANCHORED, n, d, &pidv);
    /* match states that received no set_matches call cannot exist; make the CSR monotone anyway */
    for (size_t i = 1; i <= num_match_states; i++)
        if (d->match_off[i] < d->match_off[i - 1]) d->match_off[i] = d->match_off[i - 1];
    d->match_pid = pidv.p;
    if (!d->match_pid) d->match_pid = calloc(1, sizeof(uint32_t));
    return rc;
}

static void dfa_free(dfa_t* d) {
    free(d->trans); free(d->match_off); free(d->match_pid);
    memset(d, 0, sizeof *d);
}

/* ------------------------------------------------------ contiguous NFA */
#define KIND_DENSE 0xFFu
#define KIND_ONE 0xFEu
#define MAX_SPARSE_TRANSITIONS 127u

/* contiguous.rs:1079-1085 */
static inline size_t u32_len(size_t ntrans_) { return (ntrans_ % 4 == 0) ? (ntrans_ >> 2) : ((ntrans_ >> 2) + 1); }

/* contiguous.rs:686-820 (State::write + write_sparse_trans + write_dense_trans) */
static int cnfa_state_write(const nnfa_t* n, uint32_t oldsid, const uint8_t* classes,
                            cnfa_t* c, int force_dense, uint32_t* newsid) {
    int ok = 1;
    const nstate* old = &n->states.p[oldsid];
    if (c->repr.n > SMALLINDEX_MAX) return ORC_ERR_STATE_ID_OVERFLOW;
    *newsid = (uint32_t)c->repr.n;
    size_t old_len = 0;
    for (uint32_t l = old->sparse; l != 0; l = n->sparse.p[l].link) old_len++;
    uint32_t kind;
    int is_match = old->matches != 0;
    if (force_dense || old_len > MAX_SPARSE_TRANSITIONS) kind = KIND_DENSE;
    else if (old_len == 1 && !is_match) kind = KIND_ONE;
    else kind = (uint32_t)old_len;
    if (kind == KIND_DENSE) {
        VEC_PUSH(c->repr, kind, ok);
        VEC_PUSH(c->repr, old->fail, ok);
        size_t start = c->repr.n, a = alphabet_len(classes);
        for (size_t i = 0; i < a && ok; i++) VEC_PUSH(c->repr, FAIL, ok);
        if (!ok) return ORC_ERR_NOMEM;
        for (uint32_t l = old->sparse; l != 0; l = n->sparse.p[l].link)
            c->repr.p[start + classes[n->sparse.p[l].byte]] = n->sparse.p[l].next;
    } else if (kind == KIND_ONE) {
        const ntrans* t = &n->sparse.p[old->sparse];
        uint32_t cls = classes[t->byte];
        VEC_PUSH(c->repr, kind | (cls << 8), ok);
        VEC_PUSH(c->repr, old->fail, ok);
        VEC_PUSH(c->repr, t->next, ok);
    } else {
        VEC_PUSH(c->repr, kind, ok);
        VEC_PUSH(c->repr, old->fail, ok);
        uint8_t chunk[4] = {0, 0, 0, 0};
        size_t len = 0;
        for (uint32_t l = old->sparse; l != 0; l = n->sparse.p[l].link) {
            chunk[len++] = classes[n->sparse.p[l].byte];
            if (len == 4) {
                uint32_t w; memcpy(&w, chunk, 4); /* from_ne_bytes */
                VEC_PUSH(c->repr, w, ok);
                memset(chunk, 0, 4); len = 0;
            }
        }
        if (len > 0) {
            uint8_t repeat = chunk[len - 1];
            while (len < 4) chunk[len++] = repeat;
            uint32_t w; memcpy(&w, chunk, 4);
            VEC_PUSH(c->repr, w, ok);
        }
        for (uint32_t l = old->sparse; l != 0; l = n->sparse.p[l].link)
            VEC_PUSH(c->repr, n->sparse.p[l].next, ok);
    }
    if (is_match) {
        size_t ml = nnfa_match_len(n, oldsid);
        if (ml == 1) {
            uint32_t pid = n->matches.p[old->matches].pid;
            VEC_PUSH(c->repr, (1u << 31) | pid, ok);
        } else {
            VEC_PUSH(c->repr, (uint32_t)ml, ok);
            for (uint32_t l = old->matches; l != 0; l = n->matches.p[l].link)
                VEC_PUSH(c->repr, n->matches.p[l].pid, ok);
        }
    }
    return ok ? ORC_OK : ORC_ERR_NOMEM;
}

/* contiguous.rs:486-509 */
static void cnfa_state_remap(size_t alen, const uint32_t* old_to_new, uint32_t* state) {
    uint32_t kind = state[0] & 0xFF;
    if (kind == KIND_DENSE) {
        state[1] = old_to_new[state[1]];
        for (size_t i = 0; i < alen; i++) state[2 + i] = old_to_new[state[2 + i]];
    } else if (kind == KIND_ONE) {
        state[1] = old_to_new[state[1]];
        state[2] = old_to_new[state[2]];
    } else {
        size_t trans_len = kind, classes_len = u32_len(trans_len);
        state[1] = old_to_new[state[1]];
        for (size_t i = 0; i < trans_len; i++) state[2 + classes_len + i] = old_to_new[state[2 + classes_len + i]];
    }
}

/* contiguous.rs:937-1009 */
static int cnfa_build_from_noncontiguous(const nnfa_t* n, size_t dense_depth, int byte_classes, cnfa_t* c) {
    memset(c, 0, sizeof *c);
    if (byte_classes) memcpy(c->byte_classes, n->byte_classes, 256);
    else for (int i = 0; i < 256; i++) c->byte_classes[i] = (uint8_t)i;
    c->alphabet_len = alphabet_len(c->byte_classes);
    c->state_len = n->states.n;
    uint32_t* index_to_state_id = calloc(n->states.n, sizeof(uint32_t));
    if (!index_to_state_id) return ORC_ERR_NOMEM;
    int rc = ORC_OK;
    for (size_t i = 0; i < n->states.n; i++) {
        uint32_t oldsid = (uint32_t)i;
        if (oldsid == FAIL) { index_to_state_id[oldsid] = FAIL; continue; }
        int force_dense = (size_t)n->states.p[oldsid].depth < dense_depth;
        uint32_t newsid;
        if ((rc = cnfa_state_write(n, oldsid, c->byte_classes, c, force_dense, &newsid))) goto done;
        index_to_state_id[oldsid] = newsid;
    }
    for (size_t i = 0; i < n->states.n; i++) {
        uint32_t newsid = index_to_state_id[i];
        if (newsid == FAIL) continue;
        cnfa_state_remap(c->alphabet_len, index_to_state_id, &c->repr.p[newsid]);
    }
    c->special.max_special_id = index_to_state_id[n->special.max_special_id];
    c->special.max_match_id = index_to_state_id[n->special.max_match_id];
    c->special.start_unanchored_id = index_to_state_id[n->special.start_unanchored_id];
    c->special.start_anchored_id = index_to_state_id[n->special.start_anchored_id];
done:
    free(index_to_state_id);
    return rc;
}

static void cnfa_free(cnfa_t* c) { free(c->repr.p); memset(c, 0, sizeof *c); }

/* contiguous.rs:186-247 */
static inline uint32_t cnfa_next_state(const cnfa_t* c, int anchored, uint32_t sid, uint8_t byte) {
    const uint32_t* repr = c->repr.p;
    uint8_t cls = c->byte_classes[byte];
    for (;;) {
        size_t o = sid;
        uint32_t kind = repr[o] & 0xFF;
        if (kind == KIND_DENSE) {
            uint32_t next = repr[o + 2 + cls];
            if (next != FAIL) return next;
        } else if (kind == KIND_ONE) {
            if (cls == (uint8_t)((repr[o] >> 8) & 0xFF)) return repr[o + 2];
        } else {
            size_t trans_len = kind, classes_len = u32_len(trans_len);
            size_t trans_offset = o + 2 + classes_len;
            for (size_t i = 0; i < classes_len; i++) {
                uint8_t cl[4];
                memcpy(cl, &repr[o + 2 + i], 4); /* to_ne_bytes */
                if (cl[0] == cls) return repr[trans_offset + i * 4];
                if (cl[1] == cls) return repr[trans_offset + i * 4 + 1];
                if (cl[2] == cls) return repr[trans_offset + i * 4 + 2];
                if (cl[3] == cls) return repr[trans_offset + i * 4 + 3];
            }
        }
        if (anchored) return DEAD;
        sid = repr[o + 1];
    }
}
/* contiguous.rs:581-598 */
static inline size_t cnfa_match_len(const cnfa_t* c, uint32_t sid) {
    const uint32_t* state = &c->repr.p[sid];
    uint32_t kind = state[0] & 0xFF;
    size_t start;
    if (kind == KIND_DENSE) start = 2 + c->alphabet_len;
    else { size_t tl = kind; start = 2 + u32_len(tl) + tl; }
    uint32_t packed = state[start];
    return (packed & (1u << 31)) == 0 ? packed : 1;
}
/* contiguous.rs:611-633 */
static inline uint32_t cnfa_match_pattern(const cnfa_t* c, uint32_t sid, size_t index) {
    const uint32_t* state = &c->repr.p[sid];
    uint32_t kind = state[0] & 0xFF;
    size_t start;
    if (kind == KIND_DENSE) start = 2 + c->alphabet_len;
    else { size_t tl = kind; start = 2 + u32_len(tl) + tl; }
    uint32_t packed = state[start];
    if ((packed & (1u << 31)) == 0) return state[start + 1 + index];
    return packed & ~(1u << 31);
}

/* -------------------------------------------- Automaton-trait dispatch */
static inline const special_t* aut_special(const orc_ac* ac) {
    switch (ac->kind) {
        case ORC_KIND_DFA: return &ac->dfa.special;
        case ORC_KIND_CNFA: return &ac->cnfa.special;
        default: return &ac->nnfa.special;
    }
}
/* start_state: dfa.rs:190-215, contiguous.rs:177-183, noncontiguous.rs:593-598 */
static inline int aut_start_state(const orc_ac* ac, int anchored, uint32_t* sid) {
    const special_t* sp = aut_special(ac);
    uint32_t s = anchored ? sp->start_anchored_id : sp->start_unanchored_id;
    if (ac->kind == ORC_KIND_DFA && s == DEAD)
        return anchored ? ORC_ERR_INVALID_INPUT_ANCHORED : ORC_ERR_INVALID_INPUT_UNANCHORED;
    *sid = s;
    return ORC_OK;
}
static inline uint32_t aut_next_state(const orc_ac* ac, int anchored, uint32_t sid, uint8_t byte) {
    switch (ac->kind) {
        case ORC_KIND_DFA: /* dfa.rs:218-226 */
            return ac->dfa.trans[sid + ac->dfa.byte_classes[byte]];
        case ORC_KIND_CNFA: return cnfa_next_state(&ac->cnfa, anchored, sid, byte);
        default: return nnfa_next_state(&ac->nnfa, anchored, sid, byte);
    }
}
static inline int aut_is_special(const orc_ac* ac, uint32_t sid) { return sid <= aut_special(ac)->max_special_id; }
static inline int aut_is_dead(uint32_t sid) { return sid == DEAD; }
static inline int aut_is_match(const orc_ac* ac, uint32_t sid) { return sid != DEAD && sid <= aut_special(ac)->max_match_id; }
static inline size_t aut_match_len(const orc_ac* ac, uint32_t sid) {
    switch (ac->kind) {
        case ORC_KIND_DFA: { /* dfa.rs:275-279 */
            size_t off = (sid >> ac->dfa.stride2) - 2;
            return ac->dfa.match_off[off + 1] - ac->dfa.match_off[off];
        }
        case ORC_KIND_CNFA: return cnfa_match_len(&ac->cnfa, sid);
        default: return nnfa_match_len(&ac->nnfa, sid);
    }
}
static inline uint32_t aut_match_pattern(const orc_ac* ac, uint32_t sid, size_t index) {
    switch (ac->kind) {
        case ORC_KIND_DFA: { /* dfa.rs:282-286 */
            size_t off = (sid >> ac->dfa.stride2) - 2;
            return ac->dfa.match_pid[ac->dfa.match_off[off] + index];
        }
        case ORC_KIND_CNFA: return cnfa_match_pattern(&ac->cnfa, sid, index);
        default: return nnfa_match_pattern(&ac->nnfa, sid, index);
    }
}
/* automaton.rs:1540-1549 */
static inline orc_match get_match(const orc_ac* ac, uint32_t sid, size_t index, size_t at) {
    uint32_t pid = aut_match_pattern(ac, sid, index);
    size_t len = ac->nnfa.pattern_lens.p[pid];
    orc_match m = {pid, 0, (uint64_t)(at - len), (uint64_t)at};
    return m;
}

/* -------------------------------------------------------- search loops */
typedef struct {
    const uint8_t* hay; size_t hay_len; size_t start, end; int anchored, earliest;
} input_t;

/* automaton.rs:1259-1420 (prefilter arms omitted: no prefilter exists) */
static int try_find_fwd(const orc_ac* ac, const input_t* in, int* found, orc_match* out) {
    *found = 0;
    if (in->start > in->end) return ORC_OK; /* is_done, search.rs:627-629 */
    int earliest = ac->match_kind == ORC_STANDARD || in->earliest;
    int anchored = in->anchored;
    uint32_t sid;
    int rc = aut_start_state(ac, in->anchored, &sid);
    if (rc) return rc;
    size_t at = in->start;
    if (aut_is_match(ac, sid)) {
        *out = get_match(ac, sid, 0, at); *found = 1;
        if (earliest) return ORC_OK;
    }
    while (at < in->end) {
        sid = aut_next_state(ac, anchored, sid, in->hay[at]);
        if (aut_is_special(ac, sid)) {
            if (aut_is_dead(sid)) return ORC_OK;
            else if (aut_is_match(ac, sid)) {
                orc_match m = get_match(ac, sid, 0, at + 1);
                if (!(anchored && m.start > in->start)) {
                    *out = m; *found = 1;
                    if (earliest) return ORC_OK;
                }
            }
        }
        at++;
    }
    return ORC_OK;
}

/* automaton.rs:782-827 */
typedef struct {
    int has_mat; orc_match mat;
    int has_id; uint32_t id;
    size_t at;
    int has_nmi; size_t next_match_index;
} overlapping_state;

/* automaton.rs:1423-1537 */
static int try_find_overlapping_fwd(const orc_ac* ac, const input_t* in, overlapping_state* st) {
    st->has_mat = 0;
    if (in->start > in->end) return ORC_OK;
    uint32_t sid;
    if (!st->has_id) {
        int rc = aut_start_state(ac, in->anchored, &sid);
        if (rc) return rc;
        if (aut_is_match(ac, sid)) {
            size_t i = st->has_nmi ? st->next_match_index : 0;
            size_t len = aut_match_len(ac, sid);
            if (i < len) {
                st->has_nmi = 1; st->next_match_index = i + 1;
                st->mat = get_match(ac, sid, i, in->start); st->has_mat = 1;
                return ORC_OK;
            }
        }
        st->at = in->start;
        st->has_id = 1; st->id = sid;
        st->has_nmi = 0;
        st->has_mat = 0;
    } else {
        sid = st->id;
        if (st->has_nmi) {
            size_t i = st->next_match_index;
            size_t len = aut_match_len(ac, sid);
            if (i < len) {
                st->next_match_index = i + 1;
                st->mat = get_match(ac, sid, i, st->at + 1); st->has_mat = 1;
                return ORC_OK;
            }
            st->at += 1;
            st->has_nmi = 0;
            st->has_mat = 0;
        }
    }
    while (st->at < in->end) {
        sid = aut_next_state(ac, in->anchored, sid, in->hay[st->at]);
        if (aut_is_special(ac, sid)) {
            st->has_id = 1; st->id = sid;
            if (aut_is_dead(sid)) return ORC_OK;
            else if (aut_is_match(ac, sid)) {
                st->has_nmi = 1; st->next_match_index = 1;
                st->mat = get_match(ac, sid, 0, st->at + 1); st->has_mat = 1;
                return ORC_OK;
            }
        }
        st->at += 1;
    }
    st->has_id = 1; st->id = sid;
    return ORC_OK;
}

/* ahocorasick.rs:2778-2789 */
static int enforce_anchored_consistency(int have, int want_anchored) {
    switch (have) {
        case ORC_START_BOTH: return ORC_OK;
        case ORC_START_UNANCHORED: return want_anchored ? ORC_ERR_INVALID_INPUT_ANCHORED : ORC_OK;
        default: return want_anchored ? ORC_OK : ORC_ERR_INVALID_INPUT_UNANCHORED;
    }
}

/* search.rs:332-342 */
static int check_span(size_t hay_len, size_t start, size_t end) {
    if (!(end <= hay_len && start <= end + 1)) return ORC_ERR_INVALID_SPAN;
    return ORC_OK;
}

/* ------------------------------------------------------------ public API */
void orc_config_default(orc_config* c) {
    memset(c, 0, sizeof *c);
    c->match_kind = ORC_STANDARD;
    c->start_kind = ORC_START_UNANCHORED;
    c->kind = ORC_KIND_AUTO;
    c->ascii_case_insensitive = 0;
    c->byte_classes = 1;
    c->prefilter = 1;
    c->dense_depth_set = 0;
    c->dense_depth = 0;
}

/* ahocorasick.rs:2171-2261 */
int orc_build(const orc_config* cfg, const uint8_t* const* pats, const size_t* lens, size_t n, orc_ac** out) {
    *out = NULL;
    orc_ac* ac = calloc(1, sizeof *ac);
    if (!ac) return ORC_ERR_NOMEM;
    nnfa_builder nb;
    nb.match_kind = cfg->match_kind;
    nb.ascii_case_insensitive = cfg->ascii_case_insensitive;
    /* dense_depth defaults: nNFA 3 (noncontiguous.rs:855), cNFA 2 (contiguous.rs:904);
     * AhoCorasickBuilder::dense_depth sets both (ahocorasick.rs:2581-2585);
     * usize::MAX is expressed as UINT32_MAX */
    size_t dd_n = 3, dd_c = 2;
    if (cfg->dense_depth_set) {
        size_t dd = cfg->dense_depth == UINT32_MAX ? (size_t)-1 : cfg->dense_depth;
        dd_n = dd; dd_c = dd;
    }
    nb.dense_depth = dd_n;
    ac->start_kind = cfg->start_kind;
    ac->match_kind = cfg->match_kind;
    int rc = nnfa_compile(&ac->nnfa, &nb, pats, lens, n);
    if (rc) { orc_free(ac); return rc; }
    int kind = cfg->kind;
    if (kind == ORC_KIND_AUTO) { /* build_auto :2213-2261 */
        int try_dfa = cfg->start_kind != ORC_START_BOTH && ac->nnfa.pattern_lens.n <= 100;
        if (try_dfa) {
            rc = dfa_build_from_noncontiguous(&ac->nnfa, cfg->start_kind, cfg->byte_classes, &ac->dfa);
            if (rc == ORC_OK) { ac->kind = ORC_KIND_DFA; *out = ac; return ORC_OK; }
            dfa_free(&ac->dfa);
        }
        rc = cnfa_build_from_noncontiguous(&ac->nnfa, dd_c, cfg->byte_classes, &ac->cnfa);
        if (rc == ORC_OK) { ac->kind = ORC_KIND_CNFA; *out = ac; return ORC_OK; }
        cnfa_free(&ac->cnfa);
        ac->kind = ORC_KIND_NNFA; *out = ac; return ORC_OK;
    }
    if (kind == ORC_KIND_DFA) {
        rc = dfa_build_from_noncontiguous(&ac->nnfa, cfg->start_kind, cfg->byte_classes, &ac->dfa);
        if (rc) { orc_free(ac); return rc; }
    } else if (kind == ORC_KIND_CNFA) {
        rc = cnfa_build_from_noncontiguous(&ac->nnfa, dd_c, cfg->byte_classes, &ac->cnfa);
        if (rc) { orc_free(ac); return rc; }
    }
    ac->kind = kind;
    *out = ac;
    return ORC_OK;
}

void orc_free(orc_ac* ac) {
    if (!ac) return;
    nnfa_free(&ac->nnfa);
    dfa_free(&ac->dfa);
    cnfa_free(&ac->cnfa);
    free(ac);
}

int orc_kind(const orc_ac* ac) { return ac->kind; }
int orc_match_kind(const orc_ac* ac) { return ac->match_kind; }
int orc_start_kind(const orc_ac* ac) { return ac->start_kind; }
size_t orc_patterns_len(const orc_ac* ac) { return ac->nnfa.pattern_lens.n; }
size_t orc_min_pattern_len(const orc_ac* ac) { return ac->nnfa.min_pattern_len; }
size_t orc_max_pattern_len(const orc_ac* ac) { return ac->nnfa.max_pattern_len; }

/* memory_usage: dfa.rs:289-297, contiguous.rs:310-316, noncontiguous.rs:689-696
 * (size_of::<Vec<PatternID>>() == 24 on 64-bit; State 20 B; Transition 9 B; Match 8 B) */
size_t orc_memory_usage(const orc_ac* ac) {
    switch (ac->kind) {
        case ORC_KIND_DFA:
            return ac->dfa.trans_len * 4 + ac->dfa.num_match_states * 24 +
                   (size_t)ac->dfa.match_off[ac->dfa.num_match_states] * 4 + ac->nnfa.pattern_lens.n * 4;
        case ORC_KIND_CNFA:
            return ac->cnfa.repr.n * 4 + ac->nnfa.pattern_lens.n * 4;
        default:
            return ac->nnfa.states.n * 20 + ac->nnfa.sparse.n * 9 + ac->nnfa.matches.n * 8 +
                   ac->nnfa.dense.n * 4 + ac->nnfa.pattern_lens.n * 4;
    }
}

int orc_find(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start, size_t span_end,
             int anchored, int earliest, int* found, orc_match* m) {
    int rc = check_span(hay_len, span_start, span_end);
    if (rc) return rc;
    if ((rc = enforce_anchored_consistency(ac->start_kind, anchored))) return rc;
    input_t in = {hay, hay_len, span_start, span_end, anchored, earliest};
    return try_find_fwd(ac, &in, found, m);
}

/* automaton.rs:857-936 */
int orc_find_iter(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start, size_t span_end,
                  int anchored, orc_match* out, size_t cap, size_t* n_out) {
    return orc_find_iter_ex(ac, hay, hay_len, span_start, span_end, anchored, 0, out, cap, n_out);
}

/* FindIter keeps the caller's Input -- including Input::earliest -- and every step is try_find on it
 * (automaton.rs:864-883, :1266) */
int orc_find_iter_ex(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start, size_t span_end,
                     int anchored, int earliest, orc_match* out, size_t cap, size_t* n_out) {
    *n_out = 0;
    int rc = check_span(hay_len, span_start, span_end);
    if (rc) return rc;
    if ((rc = enforce_anchored_consistency(ac->start_kind, anchored))) return rc;
    uint32_t sid;
    if ((rc = aut_start_state(ac, anchored, &sid))) return rc; /* FindIter::new :861-870 */
    input_t in = {hay, hay_len, span_start, span_end, anchored, earliest};
    int has_last = 0; size_t last_match_end = 0;
    size_t n = 0;
    for (;;) {
        int found; orc_match m;
        if ((rc = try_find_fwd(ac, &in, &found, &m))) return rc;
        if (!found) break;
        if (m.start == m.end) { /* handle_overlapping_empty_match :910-920 */
            if (has_last && m.end == last_match_end) {
                in.start += 1;
                if ((rc = try_find_fwd(ac, &in, &found, &m))) return rc;
                if (!found) break;
            }
        }
        in.start = (size_t)m.end;
        has_last = 1; last_match_end = (size_t)m.end;
        if (n < cap) out[n] = m;
        n++;
    }
    *n_out = n;
    return ORC_OK;
}

/* automaton.rs:397-423 + :954-970 */
int orc_find_overlapping_iter(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start,
                              size_t span_end, int anchored, orc_match* out, size_t cap, size_t* n_out) {
    *n_out = 0;
    int rc = check_span(hay_len, span_start, span_end);
    if (rc) return rc;
    if ((rc = enforce_anchored_consistency(ac->start_kind, anchored))) return rc;
    if (ac->match_kind != ORC_STANDARD) return ORC_ERR_UNSUPPORTED_OVERLAPPING;
    if (anchored) return ORC_ERR_INVALID_INPUT_ANCHORED;
    uint32_t sid;
    if ((rc = aut_start_state(ac, anchored, &sid))) return rc;
    input_t in = {hay, hay_len, span_start, span_end, anchored, 0};
    overlapping_state st;
    memset(&st, 0, sizeof st);
    size_t n = 0;
    for (;;) {
        if ((rc = try_find_overlapping_fwd(ac, &in, &st))) return rc;
        if (!st.has_mat) break;
        if (n < cap) out[n] = st.mat;
        n++;
    }
    *n_out = n;
    return ORC_OK;
}

static inline uint64_t fnv_fold(uint64_t h, uint64_t w) {
    for (int i = 0; i < 8; i++) { h ^= (w >> (8 * i)) & 0xFF; h *= 0x100000001B3ull; }
    return h;
}

int orc_dfa_overlapping_count(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start,
                              size_t span_end, uint64_t* count, uint64_t* hash) {
    *count = 0; *hash = 0xCBF29CE484222325ull;
    if (ac->kind != ORC_KIND_DFA) return ORC_ERR_UNSUPPORTED_OVERLAPPING;
    int rc = check_span(hay_len, span_start, span_end);
    if (rc) return rc;
    if ((rc = enforce_anchored_consistency(ac->start_kind, 0))) return rc;
    if (ac->match_kind != ORC_STANDARD) return ORC_ERR_UNSUPPORTED_OVERLAPPING;
    if (span_start > span_end) return ORC_OK;
    const dfa_t* d = &ac->dfa;
    const uint32_t* trans = d->trans;
    const uint8_t* classes = d->byte_classes;
    const uint32_t* plen = ac->nnfa.pattern_lens.p;
    uint32_t max_special = d->special.max_special_id;
    uint32_t sid = d->special.start_unanchored_id;
    uint64_t c = 0, h = *hash;
    if (aut_is_match(ac, sid)) { /* automaton.rs:1456-1464 */
        size_t off = (sid >> d->stride2) - 2;
        for (uint32_t k = d->match_off[off]; k < d->match_off[off + 1]; k++) {
            uint32_t pid = d->match_pid[k];
            h = fnv_fold(fnv_fold(fnv_fold(h, pid), span_start - plen[pid]), span_start);
            c++;
        }
    }
    for (size_t at = span_start; at < span_end; at++) { /* automaton.rs:1491-1534 */
        sid = trans[sid + classes[hay[at]]];           /* dfa.rs:218-226 */
        if (sid <= max_special) {
            if (sid == DEAD) break;
            size_t off = (sid >> d->stride2) - 2;
            for (uint32_t k = d->match_off[off]; k < d->match_off[off + 1]; k++) {
                uint32_t pid = d->match_pid[k];
                h = fnv_fold(fnv_fold(fnv_fold(h, pid), at + 1 - plen[pid]), at + 1);
                c++;
            }
        }
    }
    *count = c; *hash = h;
    return ORC_OK;
}

/* The same loop over `threads` contiguous chunks of the span in parallel (pthreads), for the "all host cores" context
 * figure of bench.py.  Chunk i warms up on max_pattern_len-1 bytes left of its begin and owns the matches that END
 * inside it (the bound the reference's stream searcher keeps, src/automaton.rs:1108), so the counts add
 * up to the sequential count.  Count only (the order-sensitive hash needs the sequential pass). */
typedef struct { const orc_ac* ac; const uint8_t* hay; size_t lo, begin, end; uint64_t count; } par_job;

static void* par_worker(void* arg) {
    par_job* j = (par_job*)arg;
    const orc_ac* ac = j->ac;
    const dfa_t* d = &ac->dfa;
    const uint32_t* trans = d->trans;
    const uint8_t* classes = d->byte_classes;
    uint32_t max_special = d->special.max_special_id;
    uint32_t sid = d->special.start_unanchored_id;
    uint64_t c = 0;
    for (size_t at = j->lo; at < j->end; at++) {
        sid = trans[sid + classes[j->hay[at]]];
        if (sid <= max_special) {
            if (sid == DEAD) break;
            if (at >= j->begin) {
                size_t off = (sid >> d->stride2) - 2;
                c += d->match_off[off + 1] - d->match_off[off];
            }
        }
    }
    j->count = c;
    return NULL;
}

int orc_dfa_overlapping_count_parallel(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start,
                                       size_t span_end, unsigned threads, uint64_t* count) {
    *count = 0;
    if (ac->kind != ORC_KIND_DFA || ac->match_kind != ORC_STANDARD) return ORC_ERR_UNSUPPORTED_OVERLAPPING;
    int rc = check_span(hay_len, span_start, span_end);
    if (rc) return rc;
    if (span_start >= span_end || threads == 0) return ORC_OK;
    if (ac->nnfa.min_pattern_len == 0) return ORC_ERR_UNSUPPORTED_EMPTY;   /* keeps the seam rule simple */
    size_t halo = ac->nnfa.max_pattern_len ? ac->nnfa.max_pattern_len - 1 : 0;
    par_job* jobs = (par_job*)calloc(threads, sizeof *jobs);
    pthread_t* tid = (pthread_t*)calloc(threads, sizeof *tid);
    if (!jobs || !tid) { free(jobs); free(tid); return ORC_ERR_NOMEM; }
    size_t n = span_end - span_start;
    for (unsigned i = 0; i < threads; i++) {
        size_t b = span_start + (size_t)((unsigned __int128)n * i / threads);
        size_t e = span_start + (size_t)((unsigned __int128)n * (i + 1) / threads);
        size_t lo = b >= span_start + halo ? b - halo : span_start;
        jobs[i] = (par_job){ac, hay, lo, b, e, 0};
        pthread_create(&tid[i], NULL, par_worker, &jobs[i]);
    }
    uint64_t total = 0;
    for (unsigned i = 0; i < threads; i++) { pthread_join(tid[i], NULL); total += jobs[i].count; }
    free(jobs); free(tid);
    *count = total;
    return ORC_OK;
}

/* Chunk-parallel find_overlapping_iter(..).collect() for full-size parity checks: the reference loop
 * (automaton.rs:1491-1534 over the automaton's own next_state -- DFA dfa.rs:218-226, contiguous NFA
 * contiguous.rs:186-247, noncontiguous NFA noncontiguous.rs:601-626) runs over `threads` contiguous pieces of the span,
 * piece i warming up on max_pattern_len-1 bytes and owning the matches that END inside it; the pieces'
 * record lists are concatenated in piece order = the sequential iterator's order.  *hash folds (pid, start, end) of
 * every record in that order (FNV-1a, the fold orc_dfa_overlapping_count uses), so it is order-sensitive. */
typedef struct {
    const orc_ac* ac; const uint8_t* hay; size_t lo, begin, end;
    orc_match* recs; size_t n, cap; int oom;
} rec_job;

static void rec_push(rec_job* j, orc_match m) {
    if (j->n == j->cap) {
        size_t nc = j->cap ? j->cap * 2 : 256;
        orc_match* p = (orc_match*)realloc(j->recs, nc * sizeof *p);
        if (!p) { j->oom = 1; return; }
        j->recs = p; j->cap = nc;
    }
    j->recs[j->n++] = m;
}

static void* rec_worker(void* arg) {
    rec_job* j = (rec_job*)arg;
    const orc_ac* ac = j->ac;
    uint32_t sid;
    if (aut_start_state(ac, 0, &sid)) return NULL;
    if (ac->kind == ORC_KIND_DFA) {   /* the hot loop of the common case without the per-byte kind switch */
        const dfa_t* d = &ac->dfa;
        const uint32_t* trans = d->trans;
        const uint8_t* classes = d->byte_classes;
        const uint32_t max_special = d->special.max_special_id;
        for (size_t at = j->lo; at < j->end; at++) {
            sid = trans[sid + classes[j->hay[at]]];
            if (sid <= max_special) {
                if (sid == DEAD) break;
                if (at >= j->begin) {
                    size_t len = aut_match_len(ac, sid);
                    for (size_t i = 0; i < len && !j->oom; i++) rec_push(j, get_match(ac, sid, i, at + 1));
                }
            }
        }
        return NULL;
    }
    for (size_t at = j->lo; at < j->end; at++) {
        sid = aut_next_state(ac, 0, sid, j->hay[at]);
        if (aut_is_special(ac, sid)) {
            if (aut_is_dead(sid)) break;
            if (at >= j->begin && aut_is_match(ac, sid)) {
                size_t len = aut_match_len(ac, sid);
                for (size_t i = 0; i < len && !j->oom; i++) rec_push(j, get_match(ac, sid, i, at + 1));
            }
        }
    }
    return NULL;
}

int orc_find_overlapping_parallel(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start,
                                  size_t span_end, unsigned threads, orc_match* out, size_t cap, size_t* n_out,
                                  uint64_t* hash) {
    *n_out = 0;
    if (hash) *hash = 0xCBF29CE484222325ull;
    int rc = check_span(hay_len, span_start, span_end);
    if (rc) return rc;
    if ((rc = enforce_anchored_consistency(ac->start_kind, 0))) return rc;
    if (ac->match_kind != ORC_STANDARD) return ORC_ERR_UNSUPPORTED_OVERLAPPING;
    if (span_start >= span_end || threads == 0) return ORC_OK;
    if (ac->nnfa.min_pattern_len == 0 && ac->nnfa.pattern_lens.n) return ORC_ERR_UNSUPPORTED_EMPTY; /* seam rule needs non-empty patterns */
    size_t halo = ac->nnfa.max_pattern_len ? ac->nnfa.max_pattern_len - 1 : 0;
    size_t n = span_end - span_start;
    if (threads > n) threads = (unsigned)n;
    rec_job* jobs = (rec_job*)calloc(threads, sizeof *jobs);
    pthread_t* tid = (pthread_t*)calloc(threads, sizeof *tid);
    if (!jobs || !tid) { free(jobs); free(tid); return ORC_ERR_NOMEM; }
    for (unsigned i = 0; i < threads; i++) {
        size_t b = span_start + (size_t)((unsigned __int128)n * i / threads);
        size_t e = span_start + (size_t)((unsigned __int128)n * (i + 1) / threads);
        size_t lo = b >= span_start + halo ? b - halo : span_start;
        jobs[i] = (rec_job){ac, hay, lo, b, e, NULL, 0, 0, 0};
    }
    unsigned started = 0;
    for (; started < threads; started++)
        if (pthread_create(&tid[started], NULL, rec_worker, &jobs[started])) break;
    for (unsigned i = started; i < threads; i++) rec_worker(&jobs[i]);   /* thread limit reached: run inline */
    uint64_t h = 0xCBF29CE484222325ull;
    size_t total = 0;
    int oom = 0;
    for (unsigned i = 0; i < threads; i++) {
        if (i < started) pthread_join(tid[i], NULL);
        oom |= jobs[i].oom;
        for (size_t k = 0; k < jobs[i].n; k++) {
            const orc_match m = jobs[i].recs[k];
            h = fnv_fold(fnv_fold(fnv_fold(h, m.pattern), m.start), m.end);
            if (out && total < cap) out[total] = m;
            total++;
        }
        free(jobs[i].recs);
    }
    free(jobs); free(tid);
    if (oom) return ORC_ERR_NOMEM;
    *n_out = total;
    if (hash) *hash = h;
    return ORC_OK;
}

/* order-sensitive hash of a record list (the fold above), for comparing a device result with *hash */
uint64_t orc_hash_matches(const orc_match* m, size_t n) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (size_t k = 0; k < n; k++) h = fnv_fold(fnv_fold(fnv_fold(h, m[k].pattern), m[k].start), m[k].end);
    return h;
}

void orc_get_tables(const orc_ac* ac, orc_tables* t) {
    memset(t, 0, sizeof *t);
    t->nnfa_states = ac->nnfa.states.n;
    t->max_match_id = ac->nnfa.special.max_match_id;
    t->start_unanchored_id = ac->nnfa.special.start_unanchored_id;
    t->start_anchored_id = ac->nnfa.special.start_anchored_id;
    memcpy(t->byte_classes, ac->nnfa.byte_classes, 256);
    t->alphabet_len = alphabet_len(ac->nnfa.byte_classes);
    t->pattern_lens = ac->nnfa.pattern_lens.p;
    if (ac->kind == ORC_KIND_DFA) {
        t->dfa_state_len = ac->dfa.state_len; t->dfa_stride2 = ac->dfa.stride2;
        t->dfa_trans = ac->dfa.trans; t->dfa_trans_len = ac->dfa.trans_len;
        t->dfa_max_match_id = ac->dfa.special.max_match_id;
        t->dfa_start_unanchored_id = ac->dfa.special.start_unanchored_id;
        t->dfa_start_anchored_id = ac->dfa.special.start_anchored_id;
        t->dfa_match_off = ac->dfa.match_off; t->dfa_match_pid = ac->dfa.match_pid;
        t->dfa_num_match_states = ac->dfa.num_match_states;
    }
    if (ac->kind == ORC_KIND_CNFA) {
        t->cnfa_repr = ac->cnfa.repr.p; t->cnfa_repr_len = ac->cnfa.repr.n;
        t->cnfa_max_match_id = ac->cnfa.special.max_match_id;
        t->cnfa_start_unanchored_id = ac->cnfa.special.start_unanchored_id;
        t->cnfa_start_anchored_id = ac->cnfa.special.start_anchored_id;
    }
}

size_t orc_nnfa_state(const orc_ac* ac, uint32_t sid, uint32_t* fail, uint32_t* depth, uint32_t* pids, size_t cap) {
    const nnfa_t* n = &ac->nnfa;
    if (fail) *fail = n->states.p[sid].fail;
    if (depth) *depth = n->states.p[sid].depth;
    size_t c = 0;
    for (uint32_t l = n->states.p[sid].matches; l != 0; l = n->matches.p[l].link) {
        if (pids && c < cap) pids[c] = n->matches.p[l].pid;
        c++;
    }
    return c;
}

uint32_t orc_nnfa_next_state(const orc_ac* ac, int anchored, uint32_t sid, uint8_t byte) {
    return nnfa_next_state(&ac->nnfa, anchored, sid, byte);
}

/* ------------------------------------------------ synthetic generator */
/* SURVEY.md Appendix C */
uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void orc_gen_haystack(uint8_t* dst, uint64_t offset, size_t len, uint64_t seed, uint32_t lo, uint32_t span) {
    for (size_t i = 0; i < len; i++)
        dst[i] = (uint8_t)(lo + (uint32_t)(orc_splitmix64(seed ^ (offset + i)) % span));
}
typedef struct { uint8_t* dst; uint64_t offset; size_t len; uint64_t seed; uint32_t lo, span; } gen_job;
static void* gen_worker(void* arg) {
    gen_job* j = (gen_job*)arg;
    orc_gen_haystack(j->dst, j->offset, j->len, j->seed, j->lo, j->span);
    return NULL;
}
/* the same bytes, generated by `threads` pthreads (full-size parity tests: 8 GiB on one core would take a minute) */
void orc_gen_haystack_parallel(uint8_t* dst, uint64_t offset, size_t len, uint64_t seed, uint32_t lo, uint32_t span,
                               unsigned threads) {
    if (threads < 2 || len < (1u << 20)) { orc_gen_haystack(dst, offset, len, seed, lo, span); return; }
    gen_job* jobs = (gen_job*)calloc(threads, sizeof *jobs);
    pthread_t* tid = (pthread_t*)calloc(threads, sizeof *tid);
    if (!jobs || !tid) { free(jobs); free(tid); orc_gen_haystack(dst, offset, len, seed, lo, span); return; }
    unsigned started = 0;
    for (unsigned i = 0; i < threads; i++) {
        size_t b = (size_t)((unsigned __int128)len * i / threads), e = (size_t)((unsigned __int128)len * (i + 1) / threads);
        jobs[i] = (gen_job){dst + b, offset + b, e - b, seed, lo, span};
    }
    for (; started < threads; started++)
        if (pthread_create(&tid[started], NULL, gen_worker, &jobs[started])) break;
    for (unsigned i = started; i < threads; i++) gen_worker(&jobs[i]);
    for (unsigned i = 0; i < started; i++) pthread_join(tid[i], NULL);
    free(jobs); free(tid);
}
size_t orc_gen_patterns(uint8_t* buf, size_t cap, uint32_t* lens, size_t n, uint64_t seed, uint32_t lo, uint32_t span) {
    uint64_t ctr = 0;
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t x = orc_splitmix64(seed ^ ctr++);
        uint32_t L = 4 + (uint32_t)(x % 13);
        if (lens) lens[i] = L;
        for (uint32_t k = 0; k < L; k++) {
            x = orc_splitmix64(seed ^ ctr++);
            if (buf && pos < cap) buf[pos] = (uint8_t)(lo + (uint32_t)(x % span));
            pos++;
        }
    }
    return pos;
}
