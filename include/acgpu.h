/*
 * acgpu.h -- C ABI of the MI355X-native Aho-Corasick search engine (libacgpu.so).
 *
 * This is the drop-in boundary for ONE path of BurntSushi/aho-corasick 1.1.3: the
 * DFA / contiguous-NFA byte-at-a-time transition walk behind
 * AhoCorasick::{find_overlapping_iter, find_iter, find, is_match}.  The reference
 * has no FFI of its own; the seam these entry points replace is the sealed
 * `Automaton` trait object the facade calls through
 * (src/ahocorasick.rs:2643-2772 -> src/automaton.rs:1259-1537).  Every entry point
 * cites the reference item it stands in for.  INTEGRATION.md shows the Rust
 * `extern "C"` block + safe wrapper a maintainer would add.
 *
 * Conventions (mirroring the reference):
 *   - the automaton is immutable after acgpu_build and may be shared between
 *     threads (AhoCorasick: Send + Sync, src/lib.rs:283-301); all per-search
 *     state lives in the call;
 *   - the haystack is borrowed for the duration of the call (Input<'h>,
 *     src/util/search.rs:83-88); match offsets are absolute haystack offsets even
 *     when a sub-span is searched (src/util/search.rs:41-48);
 *   - fallible `try_*` methods map to status codes; there is NO CPU fallback: if
 *     no HIP device is usable the search entry points return ACGPU_ERR_NO_DEVICE /
 *     ACGPU_ERR_HIP.
 *   - plain pointers and sizes only; device pointers are passed as plain
 *     pointers plus a flag.
 */
#ifndef ACGPU_H
#define ACGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: acgpu_config.engine / acgpu_profile.engine_used share ONE numbering (acgpu_engine below; version 1 had
 *    2 = LDS walk, 3 = prefix filter), and the acgpu_test_* hooks left this library (libacgpu_testhooks.so,
 *    include/acgpu_test.h).  A binding must refuse a library whose acgpu_abi_version() differs from the header it was
 *    written against. */
#define ACGPU_ABI_VERSION 3

/* BuildError kinds: src/util/error.rs:16-37; MatchErrorKind: src/util/error.rs:170-204 */
typedef enum acgpu_status {
    ACGPU_OK = 0,
    ACGPU_ERR_STATE_ID_OVERFLOW = 1,
    ACGPU_ERR_PATTERN_ID_OVERFLOW = 2,
    ACGPU_ERR_PATTERN_TOO_LONG = 3,
    ACGPU_ERR_INVALID_INPUT_ANCHORED = 10,
    ACGPU_ERR_INVALID_INPUT_UNANCHORED = 11,
    ACGPU_ERR_UNSUPPORTED_STREAM = 12,
    ACGPU_ERR_UNSUPPORTED_OVERLAPPING = 13,
    ACGPU_ERR_UNSUPPORTED_EMPTY = 14,
    ACGPU_ERR_INVALID_SPAN = 20,       /* Input::set_span assertion, src/util/search.rs:332-342 */
    ACGPU_ERR_BUFFER_TOO_SMALL = 21,   /* *n_out holds the required capacity */
    ACGPU_ERR_INVALID_ARGUMENT = 22,
    ACGPU_ERR_NOMEM = 30,
    ACGPU_ERR_HIP = 40,                /* a HIP runtime call failed; see acgpu_last_error() */
    ACGPU_ERR_NO_DEVICE = 41
} acgpu_status;

/* MatchKind, src/util/search.rs:1052-1074 */
typedef enum { ACGPU_MATCH_STANDARD = 0, ACGPU_MATCH_LEFTMOST_FIRST = 1, ACGPU_MATCH_LEFTMOST_LONGEST = 2 } acgpu_match_kind;
/* StartKind, src/util/search.rs:1133-1142 */
typedef enum { ACGPU_START_BOTH = 0, ACGPU_START_UNANCHORED = 1, ACGPU_START_ANCHORED = 2 } acgpu_start_kind;
/* Option<AhoCorasickKind>, src/ahocorasick.rs:2627-2634 (AUTO == None) */
typedef enum { ACGPU_KIND_AUTO = 0, ACGPU_KIND_NONCONTIGUOUS_NFA = 1, ACGPU_KIND_CONTIGUOUS_NFA = 2, ACGPU_KIND_DFA = 3 } acgpu_kind;

/* The device engines -- ONE numbering, used both to request an engine (acgpu_config.engine) and to report the one that
 * produced a result (acgpu_profile.engine_used).  All engines return identical results; see DESIGN.md section 3.
 *   AUTO           (request only) the library chooses, and may hand a scan from the prefix filter to another engine
 *   DFA_WALK       the transition walk sid = trans[sid + classes[byte]] (src/dfa.rs:218-226)
 *   CNFA_WALK      the failure-link walk over the contiguous NFA (src/nfa/contiguous.rs:186-247)
 *                  Requesting either walk asks for "the reference-faithful transition walk of this automaton": which of
 *                  the two runs follows from the tables the device holds (a full DFA whenever it has one).
 *   LDS_WALK       the DFA transition walk with the whole automaton in LDS (small automata)
 *   PREFIX_FILTER  occurrences enumerated by start position: LDS Bloom tables + exact trie walk of the survivors */
typedef enum {
    ACGPU_ENGINE_AUTO = 0, ACGPU_ENGINE_DFA_WALK = 1, ACGPU_ENGINE_CNFA_WALK = 2, ACGPU_ENGINE_LDS_WALK = 3,
    ACGPU_ENGINE_PREFIX_FILTER = 4
} acgpu_engine;

/* AhoCorasickBuilder, src/ahocorasick.rs:2135-2141; setters :2342-2616.
 * Initialise with acgpu_config_init (== AhoCorasickBuilder::new()). */
typedef struct acgpu_config {
    int32_t match_kind;             /* :2342  default STANDARD */
    int32_t start_kind;             /* :2437  default UNANCHORED */
    int32_t kind;                   /* :2527  default AUTO */
    int32_t ascii_case_insensitive; /* :2474  default 0 */
    int32_t byte_classes;           /* :2612  default 1 */
    int32_t prefilter;              /* :2547  default 1; accepted, results-neutral, unused on the GPU */
    int32_t dense_depth_set;        /* 0: keep per-automaton defaults (nNFA 3, cNFA 2) */
    uint32_t dense_depth;           /* :2581  UINT32_MAX == usize::MAX */
    /* --- GPU-side knobs (no reference counterpart) --- */
    uint32_t chunk_bytes;           /* bytes of haystack per wavefront lane; 0 = default */
    int32_t engine;                 /* acgpu_engine to use; ACGPU_ENGINE_AUTO (default) lets the library choose.  A request the
                                       automaton cannot honour fails the search with ACGPU_ERR_INVALID_ARGUMENT */
    int32_t gpu_dfa_fill;           /* 1: the DFA transition rows (src/dfa.rs:544-607) are computed on the device, one
                                       launch per trie depth (StartKind::Unanchored/Anchored; needs a HIP device at
                                       build time; the table is word-identical to the CPU fill).  default 0 */
    int32_t deterministic_routing;  /* 1: no adaptive hints -- by default the searches of one automaton leave each other small
                                       counters ("recent scans were abandoned by the prefix filter", "results were dense
                                       lately", "the last occurrence stream of find_iter had n records": the next call
                                       queues its whole pipeline sized by 2n and synchronises once) that steer the next calls'
                                       engine choice and save probes and host round trips; with this set the choice and the
                                       pipeline of every call follow from the automaton and the span alone (results are
                                       identical either way).  default 0 */
    uint32_t reserved[4];
} acgpu_config;

/* Match{pattern, span}, src/util/search.rs:825-830 */
typedef struct acgpu_match {
    uint32_t pattern;
    uint32_t _pad;
    uint64_t start;
    uint64_t end;
} acgpu_match;

/* Input, src/util/search.rs:83-88 */
typedef struct acgpu_input {
    const uint8_t* haystack;
    size_t haystack_len;
    size_t span_start;
    size_t span_end;
    int32_t anchored;            /* Anchored::{No=0,Yes=1} src/util/search.rs:784-792 */
    int32_t earliest;            /* Input::earliest, src/util/search.rs:300-305 */
    int32_t haystack_on_device;  /* 1: `haystack` is a device pointer on the automaton's device */
    int32_t out_on_device;       /* 1: the `out` buffer of the call is device memory */
    void* stream;                /* hipStream_t to enqueue on, or NULL for the default stream */
} acgpu_input;

/* Per-call timing of the device pipeline, filled when a non-NULL pointer is
 * passed to the *_ex entry points (HIP events on the call's stream). */
typedef struct acgpu_profile {
    float ms_scan;      /* the transition-walk / count kernel (the dominant kernel) */
    float ms_compact;   /* per-chunk count scan + active-chunk compaction (ballot/popc) */
    float ms_fill;      /* ordered match-record materialisation */
    float ms_total;     /* first launch to last completion */
    uint64_t bytes_scanned; /* algorithmic bytes: span length */
    uint64_t n_chunks;
    uint64_t n_active_chunks;
    uint64_t n_matches;
    uint32_t engine_used;   /* acgpu_engine that produced the result (never AUTO) */
    uint32_t routed;        /* 1: the prefix filter abandoned the scan (its cost model predicted engine_used to be faster on
                               this input) and the search was repeated by engine_used */
} acgpu_profile;

typedef struct acgpu_automaton acgpu_automaton;

/* Engine variants.  Every device engine exists in a few forms (table layouts, wave roles, with or without an auxiliary
 * table, ...) of which the library picks one; all return identical results.  Tests, the fuzzer and A/B measurements select a
 * form EXPLICITLY, per automaton, after acgpu_build and before its first upload or search -- there is no process-wide state:
 * the library reads no environment variable for this (it reads three in all: ACGPU_HOST_PIECE_MIB, the piece size of
 * pipelined host haystacks; ACGPU_MULTI_FORCE_RCCL / ACGPU_MULTI_NO_RCCL, the transport of acgpu_find_overlapping_multi;
 * plus ACGPU_GUARD_SHRINK in the bounds-checked debug flavour).  Names (aho-corasick_amd/csrc/host/variants.hpp):
 *   lw_flavour -1|0|1|2, lw_cls -1|0|1, lw_lane_chunk bytes            LDS walk: table flavour, class form, lane-chunk size
 *   lw_first 0|1, lw_events 0|1                                        small automata: the LDS walk before the prefix filter; records from its events
 *   pfx_min_patterns n, pfx_gate 0|1, pfx_tails 0|1|2, pfx_key8 0|1, pfx_key8_roles 12|14, pfx_key8_x2 0|1, pfx_short 0|1   large-set filter (tails: none | chains | small subtrees; short: one or two stragglers of 3..8 bytes in the long set's pass)
 *   walk_literal 0|1, walk_tri 0|1, tri_events 0|1                     transition walks
 *   pf_classic 0|1, routing 0|1, eo_fused 0|1                          prefix filter result form / hand-over of abandoned scans / the order pass's histogram inside the scan
 *   start_table 0|1, ss_window_kib n, find_iter_windows 0|1, find_iter_start_table 0|1, find_iter_disjoint 0|1, stream_split 0|1   non-overlapping forms
 * Unknown names and automata that are already on a device: ACGPU_ERR_INVALID_ARGUMENT. */
acgpu_status acgpu_set_variant(acgpu_automaton* aut, const char* name, int32_t value);

/* AhoCorasickBuilder::new(), src/ahocorasick.rs:2148 */
void acgpu_config_init(acgpu_config* cfg);

/* AhoCorasickBuilder::build, src/ahocorasick.rs:2171-2207.  Construction runs on
 * the CPU (nNFA -> DFA | contiguous NFA exactly as the reference orders it). */
acgpu_status acgpu_build(const acgpu_config* cfg, const uint8_t* const* patterns,
                         const size_t* pattern_lens, size_t n_patterns,
                         acgpu_automaton** out);
/* Drop of the Arc, src/ahocorasick.rs:176-180 */
void acgpu_free(acgpu_automaton* aut);

/* Getters, src/ahocorasick.rs:1867-2027 */
int32_t acgpu_kind_of(const acgpu_automaton* aut);
int32_t acgpu_match_kind_of(const acgpu_automaton* aut);
int32_t acgpu_start_kind_of(const acgpu_automaton* aut);
size_t acgpu_patterns_len(const acgpu_automaton* aut);
size_t acgpu_min_pattern_len(const acgpu_automaton* aut);
size_t acgpu_max_pattern_len(const acgpu_automaton* aut);
size_t acgpu_memory_usage(const acgpu_automaton* aut);

/* Replicates the built tables on HIP device `device` (no reference counterpart;
 * searches call it lazily for the current device). */
acgpu_status acgpu_upload(acgpu_automaton* aut, int device);

/* AhoCorasick::try_find_overlapping_iter(..).collect(), src/ahocorasick.rs:1350-1357
 * -> src/automaton.rs:397-423, :954-970, :1423-1537.  Writes the FULL ordered match
 * list (same triples, same order as the reference iterator).  If cap is too small
 * returns ACGPU_ERR_BUFFER_TOO_SMALL with *n_out = required count (out may be NULL
 * with cap 0 to size the buffer). */
acgpu_status acgpu_find_overlapping(acgpu_automaton* aut, const acgpu_input* input,
                                    acgpu_match* out, size_t cap, size_t* n_out);
acgpu_status acgpu_find_overlapping_ex(acgpu_automaton* aut, const acgpu_input* input,
                                       acgpu_match* out, size_t cap, size_t* n_out,
                                       acgpu_profile* prof);

/* One shard of the same search, for partitioning a haystack across GPUs: returns the
 * matches of the full-span overlapping search whose `end` lies in
 * (shard_begin, shard_end], plus -- when shard_begin == span_start -- the start-state
 * matches at span_start.  Concatenating the outputs of consecutive shards that
 * tile [span_start, span_end) reproduces acgpu_find_overlapping exactly.  Each shard
 * warms up on max_pattern_len-1 bytes left of shard_begin (clamped to span_start);
 * same bound the reference's stream searcher keeps, src/automaton.rs:1108. */
acgpu_status acgpu_find_overlapping_shard(acgpu_automaton* aut, const acgpu_input* input,
                                          size_t shard_begin, size_t shard_end,
                                          acgpu_match* out, size_t cap, size_t* n_out,
                                          acgpu_profile* prof);

/* Enqueue-only form of acgpu_find_overlapping_shard for pipelined callers (no reference counterpart): haystack,
 * `out` and `totals` are device memory; the call returns as soon as the kernels are enqueued on input->stream, so a
 * caller can queue the next buffer while this one is scanned and never pays a host round trip per call.
 *   totals[0] = number of match records, totals[1] = number of occurrences events (device, written in stream order).
 * The records are in out[0 .. totals[0]) iff totals[1] <= ACGPU_ENQUEUE_MAX_EVENTS and totals[0] <= cap.  Up to
 * ACGPU_ENQUEUE_MAX_EVENTS occurrences are ordered by the all-pairs rank queued behind the scan.  Beyond that the call
 * delivers through the bucket order pass (device/event_order.hip) IF it queued one -- it does while the automaton's recent
 * synchronous results were that dense -- and then resets totals[1] to 0, so the same test tells the caller the records
 * are there; if it did not (totals[1] still > ACGPU_ENQUEUE_MAX_EVENTS), or if totals[0] > cap, nothing usable was
 * written and the caller repeats the search with acgpu_find_overlapping_shard (which has no such limit and primes the
 * dense path for the next enqueue).  totals[1] == UINT64_MAX: the prefix filter abandoned the scan because its cost model
 * predicts another engine to be faster on this input (the synchronous call switches to it) -- or the order pass, queued as
 * the fused chain (its histogram inside the scan, DESIGN.md 3.1b: then it serves ANY number of occurrences and a delivered
 * call reports totals[1] = 0 whatever their number), could not deliver: more records than `cap`, or more occurrences than
 * its list holds.  The caller's test and reaction are the same in every case.
 * Automata the prefix-filter engine serves (Standard, unanchored start available, no empty pattern, up to 131072
 * patterns) take the event form above; every other automaton -- and any automaton when `flags` of the _ex form contains
 * ACGPU_ENQUEUE_CLASSIC (dense results expected) -- runs chunk counters -> scan -> fill, all reading their sizes on the
 * device: no occurrence limit, totals[1] = 0.  `slot` (0..63, or -1): HIP events of the calling stream's
 * context are recorded around the scan kernel of this call; read them with acgpu_enqueue_kernel_ms after the
 * stream has been synchronised.  The automaton, the haystack and both output buffers must stay alive until the
 * enqueued work has completed; calls on one stream must not be issued concurrently from several host threads
 * (different streams are independent: each has its own context). */
#define ACGPU_ENQUEUE_MAX_EVENTS 16384
acgpu_status acgpu_find_overlapping_enqueue(acgpu_automaton* aut, const acgpu_input* input,
                                            size_t shard_begin, size_t shard_end,
                                            acgpu_match* out, size_t cap, uint64_t* totals, int32_t slot);
#define ACGPU_ENQUEUE_CLASSIC 1u
acgpu_status acgpu_find_overlapping_enqueue_ex(acgpu_automaton* aut, const acgpu_input* input,
                                               size_t shard_begin, size_t shard_end,
                                               acgpu_match* out, size_t cap, uint64_t* totals, int32_t slot,
                                               uint32_t flags);
acgpu_status acgpu_enqueue_kernel_ms(acgpu_automaton* aut, void* stream, int32_t slot, float* ms);

/* Debug flavour of the library (make -C aho-corasick_amd/csrc guard -> libacgpu_guard.so, -DACGPU_GUARD): every haystack
 * access of every kernel is checked against the 16-byte-aligned hull of the searched span and violations are counted
 * (bounds-checked debug kernels; the reference relies on Rust's bounds checks).  Returns the
 * number of violations since the process started, or -1 in the normal build. */
long long acgpu_guard_violations(void);

/* One overlapping search partitioned over several devices of a node, from one host process (no reference counterpart:
 * the crate is single-threaded).  Shard i is a contiguous piece of the haystack that lives in the
 * memory of shards[i].device together with the max_pattern_len-1 bytes left of it (its warm-up; the same bound the
 * reference's stream searcher keeps, src/automaton.rs:1108):
 *     haystack / haystack_len      the device-resident bytes of this shard, halo included
 *     span_start, span_end         the searched span in the coordinates of that buffer (span_start = where the whole
 *                                  search begins if this is the first shard, else the start of the halo)
 *     shard_begin, shard_end       the shard owns the matches whose end lies in (shard_begin, shard_end] (buffer coordinates)
 *     global_offset                position of buffer byte 0 in the whole haystack: added to start/end of its records
 * Shards must be listed in haystack order.  All devices scan concurrently (one enqueue-only search per shard on a
 * per-device stream); the records are gathered in shard order -- which is the order of acgpu_find_overlapping over the
 * whole haystack -- into `out`, device memory of dst_device: with RCCL (ncclSend/ncclRecv over xGMI; librccl.so is
 * opened at run time) when every shard has its own device, with peer copies otherwise (several shards on one device are
 * allowed: "virtual shards").  *n_out = total records; shard_counts (optional, n_shards entries, host) = per shard.
 * ACGPU_ERR_BUFFER_TOO_SMALL with the required *n_out when they do not fit `cap`. */
typedef struct acgpu_shard {
    int32_t device;
    int32_t _pad;
    const uint8_t* haystack;
    size_t haystack_len;
    size_t span_start, span_end;
    size_t shard_begin, shard_end;
    uint64_t global_offset;
} acgpu_shard;
acgpu_status acgpu_find_overlapping_multi(acgpu_automaton* aut, const acgpu_shard* shards, size_t n_shards,
                                          int32_t dst_device, acgpu_match* out, size_t cap, size_t* n_out,
                                          uint64_t* shard_counts);
/* what moved the records of the last acgpu_find_overlapping_multi call of this process: 1 = device copies, 2 = RCCL */
int32_t acgpu_multi_last_transport(void);
const char* acgpu_multi_last_error(void);

/* Device memory for callers without a HIP binding of their own (a Rust host links only libacgpu.so):
 * kind: 0 host->device, 1 device->host, 2 device->device (same device). */
acgpu_status acgpu_device_count(int32_t* n);
acgpu_status acgpu_device_malloc(int32_t device, size_t bytes, void** out);
acgpu_status acgpu_device_free(int32_t device, void* p);
acgpu_status acgpu_device_copy(int32_t device, void* dst, const void* src, size_t bytes, int32_t kind);

/* AhoCorasick::try_find_iter(..).collect(), src/ahocorasick.rs:1275-1282
 * -> src/automaton.rs:857-936 (incl. the empty-match rule :910-920).  The device selects the iterator's matches from the
 * occurrence stream of the same patterns (all three MatchKinds), or -- leftmost kinds on occurrence-dense input -- from a
 * per-start candidate table; inputs neither form covers (anchored searches, empty patterns,
 * Input::earliest on a leftmost automaton) run the reference loop on one lane.  Same records in every case. */
acgpu_status acgpu_find_iter(acgpu_automaton* aut, const acgpu_input* input,
                             acgpu_match* out, size_t cap, size_t* n_out);
acgpu_status acgpu_find_iter_ex(acgpu_automaton* aut, const acgpu_input* input,
                                acgpu_match* out, size_t cap, size_t* n_out,
                                acgpu_profile* prof);

/* AhoCorasick::try_find, src/ahocorasick.rs:1021-1028 -> src/automaton.rs:1259-1420. */
acgpu_status acgpu_find(acgpu_automaton* aut, const acgpu_input* input, int32_t* found,
                        acgpu_match* m);
/* AhoCorasick::is_match, src/ahocorasick.rs:311-316 (earliest = true). */
acgpu_status acgpu_is_match(acgpu_automaton* aut, const acgpu_input* input, int32_t* is_match);

/* AhoCorasick::try_replace_all_bytes, src/ahocorasick.rs:1447-1457 -> Automaton::try_replace_all_bytes /
 * try_replace_all_with_bytes, src/automaton.rs:464-485, :530-550: every non-overlapping match of find_iter (per the
 * automaton's MatchKind) over the WHOLE haystack is replaced by replace_with[match.pattern].  n_replace must equal
 * the number of patterns (the reference asserts, :473-478) -> ACGPU_ERR_INVALID_ARGUMENT.  `input` must describe the
 * whole haystack, unanchored (the reference always uses Input::new(haystack)).  `out` receives *out_len bytes (host
 * memory, or device memory when input->out_on_device); ACGPU_ERR_BUFFER_TOO_SMALL reports the required *out_len.
 * flags: ACGPU_REPLACE_UTF8_BOUNDARIES = the &str variant (AhoCorasick::try_replace_all, src/ahocorasick.rs:1396-1406
 * -> src/automaton.rs:433-454, :493-522): matches whose start or end is not a UTF-8 char boundary are skipped. */
#define ACGPU_REPLACE_UTF8_BOUNDARIES 1u
acgpu_status acgpu_replace_all(acgpu_automaton* aut, const acgpu_input* input,
                               const uint8_t* const* replace_with, const size_t* replace_lens, size_t n_replace,
                               uint32_t flags, uint8_t* out, size_t cap, size_t* out_len);

/* --- stream search: AhoCorasick::try_stream_find_iter, src/ahocorasick.rs:1677-1683 -> StreamChunkIter,
 * src/automaton.rs:1036-1244.  A stream is fed chunk by chunk (each chunk = one std::io::Read refill, of any size);
 * every feed reports the matches that END inside that chunk, with absolute stream offsets, identical to
 * StreamFindIter's sequence.  Errors as the reference: ACGPU_ERR_UNSUPPORTED_STREAM unless MatchKind::Standard
 * (:1067-1069), ACGPU_ERR_UNSUPPORTED_EMPTY with an empty pattern (:1082-1084), ACGPU_ERR_INVALID_INPUT_UNANCHORED
 * for StartKind::Anchored automata.  Between feeds the object keeps max_pattern_len-1 bytes (the roll buffer's
 * minimum, src/util/buffer.rs) and the end of the last match. */
typedef struct acgpu_stream acgpu_stream;
acgpu_status acgpu_stream_begin(acgpu_automaton* aut, acgpu_stream** out);
acgpu_status acgpu_stream_feed(acgpu_stream* s, const uint8_t* bytes, size_t len, int32_t bytes_on_device,
                               void* hip_stream, size_t* n_matches);
/* the matches of the most recent feed (host memory) */
acgpu_status acgpu_stream_matches(const acgpu_stream* s, acgpu_match* out, size_t cap, size_t* n_out);
void acgpu_stream_end(acgpu_stream* s);

/* --- table introspection (host tables; used by the table-parity tests) --- */
typedef struct acgpu_tables {
    size_t nnfa_states;
    uint32_t nnfa_max_match_id, nnfa_start_unanchored_id, nnfa_start_anchored_id;
    uint8_t byte_classes[256];
    size_t alphabet_len;
    const uint32_t* nnfa_fail;       /* [nnfa_states] */
    const uint32_t* nnfa_depth;      /* [nnfa_states] */
    const uint32_t* nnfa_match_off;  /* [nnfa_states+1] */
    const uint32_t* nnfa_match_pid;
    const uint32_t* dfa_trans;       /* premultiplied, src/dfa.rs:95 */
    size_t dfa_trans_len, dfa_state_len, dfa_stride2;
    uint32_t dfa_max_match_id, dfa_start_unanchored_id, dfa_start_anchored_id;
    const uint32_t* dfa_match_off;   /* [dfa_num_match_states+1] */
    const uint32_t* dfa_match_pid;
    size_t dfa_num_match_states;
    const uint32_t* cnfa_repr;       /* src/nfa/contiguous.rs:96 */
    size_t cnfa_repr_len;
    uint32_t cnfa_max_match_id, cnfa_start_unanchored_id, cnfa_start_anchored_id;
    const uint32_t* pattern_lens;
} acgpu_tables;
void acgpu_get_tables(const acgpu_automaton* aut, acgpu_tables* t);

/* --- utilities --- */
/* Synthetic haystack: byte i = lo + splitmix64(seed ^ (offset+i)) % span,
 * generated on the device into dst[0..len). */
acgpu_status acgpu_gen_haystack(uint8_t* dst_device, uint64_t offset, size_t len,
                                uint64_t seed, uint32_t lo, uint32_t span, void* stream);
/* Measurement aid: reads src_device[0..len) (16-byte aligned) once with a plain streaming kernel, `iters` times, and
 * reports the best time in *ms_best -- the empirical ceiling of "read every haystack byte once" on this device, next to
 * which the scan kernels' rates are quoted. */
acgpu_status acgpu_stream_read(const uint8_t* src_device, size_t len, int32_t iters, float* ms_best, void* stream);
/* Thread-local text of the last HIP error seen by this library. */
const char* acgpu_last_error(void);
const char* acgpu_status_str(acgpu_status s);
uint32_t acgpu_abi_version(void);

/* --- environment variables ---
 * The library reads three, per call, none needed in production (all results are identical whatever they say):
 *     ACGPU_HOST_PIECE_MIB=<n>    host haystacks / stream feeds: size of the pieces copied under the scan (default 256)
 *     ACGPU_MULTI_FORCE_RCCL, ACGPU_MULTI_NO_RCCL   acgpu_find_overlapping_multi: transport of the gather
 * and ACGPU_GUARD_SHRINK in libacgpu_guard.so only (shrinks the permitted hull: the positive control of the guard test).
 * Every other choice between forms of an engine is an explicit, per-automaton VARIANT (acgpu_set_variant above): rounds
 * 1-4 read 36 environment variables for them; none is left.
 *   the Python binding: ACGPU_LIB=<path> loads another flavour of the library (guard / host-ASan / experiment builds). */

#ifdef __cplusplus
}
#endif
#endif /* ACGPU_H */
