/* Test hooks of libacgpu: NOT part of the product ABI (include/acgpu.h, libacgpu.so).  They live in
 * libacgpu_testhooks.so (make -C aho-corasick_amd/csrc testhooks; it links libacgpu.so) and in the host-ASan flavour of
 * the library.  Each builds the host-side tables of one device engine and replays that engine's decisions on the CPU,
 * so table construction and step logic are checked against the oracle without a GPU. */
#ifndef ACGPU_TEST_H
#define ACGPU_TEST_H
#include "acgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook, not a search path: applies the non-overlapping selection rule of the parallel find_iter to a HOST array
 * holding an ordered occurrence stream (what acgpu_find_overlapping returns for the MatchKind::Standard automaton of
 * the same patterns).  Lets the rule be checked against the oracle without a GPU. */
acgpu_status acgpu_test_select_host(const acgpu_match* stream, size_t n, int32_t match_kind, size_t span_start,
                                    size_t max_pattern_len, acgpu_match* out, size_t cap, size_t* n_out);

/* Test hook, not a search path: builds the LDS-walk engine's tables (one row per state, or dense rows + single-exception
 * handles + exception chains, device/lds_walk.hip) for a Standard / unanchored DFA-kind automaton on the host and walks
 * haystack[0..len) with the kernel's own step rules on the CPU (cold start at 0).  *n_matches = what the overlapping
 * search would count.  On entry info[0] = 1 + forced flavour (0 = the engine's choice; 1 narrow, 2 wide, 3 one row per
 * state), info[1] = 1 + forced class form (0 = the engine's choice; 1 LDS map, 2 computed).  On return
 * info[0..7] = {eligible, image bytes, dense rows, multi states, classes, states, dwords that took the exact path,
 * wide-row-index layout (bit 0) | one row per state (bit 1) | computed classes (bit 2) | monotone (bit 3) | disjoint (bit 4:
 * LwHostTables, host/lw_tables.hpp) | estimated share of exact-path dwords on pattern-like input in ppm << 8 (routing price)}.
 * Lets table construction and the fast-step / inline-count / exact-redo logic be checked against the oracle without a GPU. */
acgpu_status acgpu_test_lw_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint64_t* n_matches,
                                uint64_t* info);

/* Test hook, not a search path: the match records of the overlapping search over haystack[0..len) as the record fill of the
 * LDS walk's one-row-per-state form writes them (device/lds_walk.hip: k_lw_fill -- match lists {pattern, length} in the LDS
 * image, src/dfa.rs:275-279), produced on the host from the same tables.  *served = 0: the automaton has no such form. */
acgpu_status acgpu_test_lw_records_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, acgpu_match* out, size_t cap,
                                        size_t* n_out, int32_t* served);

/* Test hook, not a search path: the same records through a host model of the EVENT form of the LDS walk (device/lds_emit.hip):
 * lane-chunks of `chunk` bytes (a power of two >= 16), one event per dword that gained a record (byte masks on the ragged
 * edges), scan of the lane-chunk counts, re-walk of the events in reverse order of arrival. */
acgpu_status acgpu_test_lw_event_records_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint32_t chunk,
                                              acgpu_match* out, size_t cap, size_t* n_out, int32_t* served);

/* Test hook, not a search path: the routing rules the library applies (aho-corasick_amd/csrc/host/engine_plan.hpp) as pure
 * functions of explicit facts.  facts[0..6] = {device holds a DFA, prefix-filter tables, LDS-walk tables, large-set tables,
 * shortest pattern, requested engine as the pipelines test it (0 auto, 1 transition walk, 2 LDS walk, 3 prefix filter),
 * variant `routing`}; hints[0..1] = {probe_skip, route_hint} of the automaton; out[0..2] = {first engine (acgpu_engine; 0:
 * the request cannot be honoured), engine an abandoned scan is handed to (100 = the large-set filter), 0 scan | 1 probe
 * first | 2 take the alternative unasked}. */
acgpu_status acgpu_test_engine_plan(const uint64_t* facts, const int32_t* hints, uint64_t span_bytes, int32_t first_kernel_is_large_set,
                                    uint32_t* out);

/* Test hook: log2 of the bucket size the event order pass (device/event_order.hip) takes for a scan that may record up to
 * max_events events / max_records records over span_bytes haystack bytes. */
uint32_t acgpu_test_event_order_shift(uint64_t max_events, uint64_t max_records, uint64_t span_bytes);

/* Test hook, not a search path: builds the tables of the prefix-filter kernels (device/pf_scan.hip: kernel 0;
 * device/pfx_scan.hip with its 4-byte / long-prefix level 2: kernels 1 / 2; 3 = the long-prefix form with the
 * eight-byte level 1; 4 = ... probed at every other position) on the host and replays the kernels'
 * decisions over haystack[0..len) on the CPU (cold start at 0): *n_matches = the occurrences level 3 finds -- the
 * overlapping search's count if and only if the tables let every occurrence through.  info[0..7] = {two-type filter
 * serves the automaton, the requested large-set kernel does, level-1 survivors, level-2 survivors, exact prefix
 * depth of the long level 2, patterns, exact second table, 4-byte bit table in use}. */
acgpu_status acgpu_test_pf_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, int32_t kernel,
                                uint64_t* n_matches, uint64_t* info);

/* Test hook, not a search path: builds the tables of the contiguous-NFA walk kernel (device/cnfa_walk.hip: the states
 * held in LDS and the copy of `repr` that names them by slot) on the host and walks haystack[0..len) with the kernel's
 * step on the CPU (cold start at 0).  info[0..5] = {kernel serves the automaton, LDS slots, a dense state lives outside
 * LDS, sparse classes ascending, an LDS-resident state is a match state, patched words}. */
acgpu_status acgpu_test_cnfa_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint64_t* n_matches,
                                  uint64_t* info);

/* Test hook, not a search path: the tables of the contiguous-NFA shallow-skip walk (device/cnfa_tri.hip: trigram
 * bitmap of the trie nodes of depth 3, their 16-byte child entries, the copy of `repr` with the fail words into depth
 * <= 2 tagged) built on the host, and the kernel's walk over haystack[0..len) on the CPU (cold start at 0).
 * info[0..7] = {kernel serves the automaton, compact classes, bitmap words per pair, child granule, a state of depth
 * <= 2 is a match state, LDS bytes, gathers of the walk, FNV-1a hash over the (pattern, start, end) of the records the
 * walk's match events stand for, in output order (0: the events do not tile the output)}. */
acgpu_status acgpu_test_cnfa_tri_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint64_t* n_matches,
                                      uint64_t* info);

/* Test hook, not a search path: the tables of the DFA shallow-skip walk (device/dfa_tri.hip: trigram bitmap of the trie
 * nodes of depth 3, their state ids, the copy of the transition table with the targets of depth <= 2 tagged) built on
 * the host -- for NFA-kind automata from the DFA of the same noncontiguous NFA, as the upload derives it -- and the
 * kernel's walk over haystack[0..len) on the CPU (cold start at 0).  info[0..7] = {kernel serves the automaton,
 * compact classes, bitmap words per pair, child granule, a state of depth <= 2 is a match state, LDS bytes, steps of
 * the reference loop that start below depth 2, FNV-1a hash over the records of the walk's match events (0: the events
 * do not tile the output)}. */
acgpu_status acgpu_test_dfa_tri_host(const acgpu_automaton* aut, const uint8_t* haystack, size_t len, uint64_t* n_matches,
                                     uint64_t* info);

#ifdef __cplusplus
}
#endif
#endif /* ACGPU_TEST_H */
