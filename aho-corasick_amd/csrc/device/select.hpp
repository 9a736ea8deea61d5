// Non-overlapping match selection over the ordered stream of ALL pattern occurrences.
//
// The reference's FindIter (src/automaton.rs:857-936) restarts a search at the end of the previous match
// (src/automaton.rs:1285-1420).  For pattern sets without an empty pattern its observable result is a function
// of the occurrence set (reference docs src/util/search.rs:966-1049; checked against the
// oracle under hypothesis in tests/test_oracle_naive.py and tests/test_select_rule.py):
//   Standard         next = the first occurrence in overlapping-stream order (end ascending, then the match-list
//                    order of the state) whose start >= pos
//   LeftmostFirst    next = the occurrence with the smallest start >= pos; ties: lowest pattern id
//   LeftmostLongest  next = the occurrence with the smallest start >= pos; ties: greatest length, then lowest id
// then pos = next.end.  `S` is that stream (what acgpu_find_overlapping produces for the Standard automaton of
// the same patterns); `L` = max pattern length bounds how far ahead a better (earlier-starting) occurrence can
// still appear in an end-ordered stream.
#pragma once
#include <stdint.h>

#include "acgpu.h"

namespace acgpu {

template <class Emit>
__host__ __device__ inline uint64_t select_nonoverlapping(const acgpu_match* S, uint64_t M, int match_kind,
                                                          uint64_t span_start, uint64_t L, Emit emit) {
    uint64_t pos = span_start, n = 0;
    if (match_kind == ACGPU_MATCH_STANDARD) {
        for (uint64_t i = 0; i < M; i++) {
            if (S[i].start >= pos) { emit(n++, S[i]); pos = S[i].end; }
        }
        return n;
    }
    uint64_t i = 0;
    for (;;) {
        bool have = false;
        acgpu_match best{};
        for (uint64_t j = i; j < M; j++) {
            const acgpu_match m = S[j];
            if (have && m.end > best.start + L) break;  // no later occurrence can start at or before best.start
            if (m.start < pos) continue;
            bool better = !have || m.start < best.start;
            if (have && m.start == best.start) {
                if (match_kind == ACGPU_MATCH_LEFTMOST_FIRST) better = m.pattern < best.pattern;
                else {
                    const uint64_t lm = m.end - m.start, lb = best.end - best.start;
                    better = lm > lb || (lm == lb && m.pattern < best.pattern);
                }
            }
            if (better) { best = m; have = true; }
        }
        if (!have) break;
        emit(n++, best);
        pos = best.end;
        while (i < M && S[i].end <= pos) i++;  // an occurrence with start >= pos ends after pos (no empty patterns)
    }
    return n;
}

}  // namespace acgpu
