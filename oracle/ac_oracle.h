/*
 * ac_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the reference crate's (BurntSushi/aho-corasick 1.1.3)
 * automaton construction and search loops. It exists so that the HIP product
 * path can be checked bit-for-bit against "what the reference computes".
 *
 * NOTHING in the product (aho-corasick_amd/, include/) may include, link or
 * call this code. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Parity pin: the reference is Rust and there is no rustc/cargo in this image,
 * so the reference itself cannot be run here. The oracle is pinned against
 * every golden vector of the reference's own test-suite (src/tests.rs:96-642,
 * transcribed by tests/golden/extract_vectors.py into
 * tests/golden/reference_vectors.json) under the same builder-config matrix
 * (src/tests.rs:723-1323), plus the doctest vectors of src/automaton.rs:756-779.
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef AC_ORACLE_H
#define AC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/util/search.rs:1052-1074 */
enum { ORC_STANDARD = 0, ORC_LEFTMOST_FIRST = 1, ORC_LEFTMOST_LONGEST = 2 };
/* src/util/search.rs:1133-1142 */
enum { ORC_START_BOTH = 0, ORC_START_UNANCHORED = 1, ORC_START_ANCHORED = 2 };
/* src/ahocorasick.rs:2627-2634 (+ AUTO = `None`) */
enum { ORC_KIND_AUTO = 0, ORC_KIND_NNFA = 1, ORC_KIND_CNFA = 2, ORC_KIND_DFA = 3 };

/* status codes: 0 ok; build errors src/util/error.rs:16-37; match errors :170-204 */
enum {
    ORC_OK = 0,
    ORC_ERR_STATE_ID_OVERFLOW = 1,
    ORC_ERR_PATTERN_ID_OVERFLOW = 2,
    ORC_ERR_PATTERN_TOO_LONG = 3,
    ORC_ERR_INVALID_INPUT_ANCHORED = 10,
    ORC_ERR_INVALID_INPUT_UNANCHORED = 11,
    ORC_ERR_UNSUPPORTED_STREAM = 12,
    ORC_ERR_UNSUPPORTED_OVERLAPPING = 13,
    ORC_ERR_UNSUPPORTED_EMPTY = 14,
    ORC_ERR_INVALID_SPAN = 20,
    ORC_ERR_NOMEM = 30
};

/* AhoCorasickBuilder knobs, src/ahocorasick.rs:2135-2141 and setters :2342-2616.
 * dense_depth: SIZE_MAX-like "all dense" is expressed as UINT32_MAX.
 * dense_depth_set == 0 keeps the per-automaton defaults (nNFA 3, cNFA 2). */
typedef struct {
    int match_kind;
    int start_kind;
    int kind;
    int ascii_case_insensitive;
    int byte_classes;   /* default 1 */
    int prefilter;      /* accepted, results-neutral, ignored */
    int dense_depth_set;
    uint32_t dense_depth;
} orc_config;

typedef struct {
    uint32_t pattern;
    uint32_t _pad;
    uint64_t start;
    uint64_t end;
} orc_match;

typedef struct orc_ac orc_ac;

void orc_config_default(orc_config* c);

int orc_build(const orc_config* cfg, const uint8_t* const* pats,
              const size_t* lens, size_t n, orc_ac** out);
void orc_free(orc_ac* ac);

/* getters: src/ahocorasick.rs:1867-2027 */
int orc_kind(const orc_ac* ac);
int orc_match_kind(const orc_ac* ac);
int orc_start_kind(const orc_ac* ac);
size_t orc_patterns_len(const orc_ac* ac);
size_t orc_min_pattern_len(const orc_ac* ac);
size_t orc_max_pattern_len(const orc_ac* ac);
size_t orc_memory_usage(const orc_ac* ac);

/* AhoCorasick::try_find (src/ahocorasick.rs:1021). *found = 0/1. */
int orc_find(const orc_ac* ac, const uint8_t* hay, size_t hay_len,
             size_t span_start, size_t span_end, int anchored, int earliest,
             int* found, orc_match* m);

/* find_iter(..).collect() (src/automaton.rs:857-936).
 * Writes up to cap matches, *n_out = total number of matches. */
int orc_find_iter(const orc_ac* ac, const uint8_t* hay, size_t hay_len,
                  size_t span_start, size_t span_end, int anchored,
                  orc_match* out, size_t cap, size_t* n_out);
/* the same with Input::earliest carried by the iterator (src/automaton.rs:864-883 keeps the caller's Input; :1266) */
int orc_find_iter_ex(const orc_ac* ac, const uint8_t* hay, size_t hay_len,
                     size_t span_start, size_t span_end, int anchored, int earliest,
                     orc_match* out, size_t cap, size_t* n_out);

/* find_overlapping_iter(..).collect() (src/automaton.rs:954-970, :1423-1537). */
int orc_find_overlapping_iter(const orc_ac* ac, const uint8_t* hay,
                              size_t hay_len, size_t span_start,
                              size_t span_end, int anchored, orc_match* out,
                              size_t cap, size_t* n_out);

/* Count-only fast form of the overlapping scan used by bench.py's
 * cpu_baseline leg: same loop as src/automaton.rs:1491-1534 on the DFA
 * (src/dfa.rs:218-226), no record materialisation. Returns number of matches
 * and folds (pid,start,end) into *hash (FNV-1a over the 3 u64 words). */
int orc_dfa_overlapping_count(const orc_ac* ac, const uint8_t* hay,
                              size_t hay_len, size_t span_start,
                              size_t span_end, uint64_t* count, uint64_t* hash);
/* chunk-parallel count of the same loop over `threads` pthreads (count only; bench.py's all-cores context figure) */
int orc_dfa_overlapping_count_parallel(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start,
                                       size_t span_end, unsigned threads, uint64_t* count);

/* chunk-parallel find_overlapping_iter(..).collect() for any automaton kind (full-size parity checks): `threads` pieces
 * of the span, each warming up on max_pattern_len-1 bytes and owning the matches that end inside it; records in the
 * sequential iterator's order (up to cap written, *n_out = total), *hash = order-sensitive FNV-1a over
 * (pid, start, end) of all of them.  Patterns must be non-empty. */
int orc_find_overlapping_parallel(const orc_ac* ac, const uint8_t* hay, size_t hay_len, size_t span_start,
                                  size_t span_end, unsigned threads, orc_match* out, size_t cap, size_t* n_out,
                                  uint64_t* hash);
uint64_t orc_hash_matches(const orc_match* m, size_t n);

/* --- table introspection (for table-parity tests against the product) --- */
typedef struct {
    /* noncontiguous NFA (always present) */
    size_t nnfa_states;
    uint32_t max_match_id, start_unanchored_id, start_anchored_id;
    uint8_t byte_classes[256];
    size_t alphabet_len;
    /* DFA (kind == DFA) */
    size_t dfa_state_len, dfa_stride2;
    const uint32_t* dfa_trans;
    size_t dfa_trans_len;
    uint32_t dfa_max_match_id, dfa_start_unanchored_id, dfa_start_anchored_id;
    const uint32_t* dfa_match_off; /* CSR over match-state index */
    const uint32_t* dfa_match_pid;
    size_t dfa_num_match_states;
    /* contiguous NFA (kind == CNFA) */
    const uint32_t* cnfa_repr;
    size_t cnfa_repr_len;
    uint32_t cnfa_max_match_id, cnfa_start_unanchored_id, cnfa_start_anchored_id;
    const uint32_t* pattern_lens;
} orc_tables;

void orc_get_tables(const orc_ac* ac, orc_tables* t);
/* nNFA per-state info: fail, depth, number of matches; match list copied to
 * pids (cap entries). Returns number of matches of that state. */
size_t orc_nnfa_state(const orc_ac* ac, uint32_t sid, uint32_t* fail,
                      uint32_t* depth, uint32_t* pids, size_t cap);
/* nNFA next_state (src/nfa/noncontiguous.rs:601-626) */
uint32_t orc_nnfa_next_state(const orc_ac* ac, int anchored, uint32_t sid,
                             uint8_t byte);

/* --- synthetic input generator --- */
uint64_t orc_splitmix64(uint64_t x);
void orc_gen_haystack(uint8_t* dst, uint64_t offset, size_t len, uint64_t seed,
                      uint32_t lo, uint32_t span);
void orc_gen_haystack_parallel(uint8_t* dst, uint64_t offset, size_t len, uint64_t seed, uint32_t lo, uint32_t span,
                               unsigned threads);
/* Generates n patterns back-to-back into buf (cap bytes), lengths into lens.
 * Returns total bytes needed. */
size_t orc_gen_patterns(uint8_t* buf, size_t cap, uint32_t* lens, size_t n,
                        uint64_t seed, uint32_t lo, uint32_t span);

#ifdef __cplusplus
}
#endif
#endif
