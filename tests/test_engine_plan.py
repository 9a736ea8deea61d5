"""not-gpu: the routing rules of the library (aho-corasick_amd/csrc/host/engine_plan.hpp -- the functions capi_overlap.cpp / capi_enqueue.cpp call) as
a table: which engine a search starts with, which one an abandoned prefix-filter scan is handed to, and what the adaptive
hints of an automaton change about that.  Every engine returns identical results; the plan decides cost only."""
import ctypes as C

import pytest

import aho_corasick_amd as ac

DFA, CNFA, LDS, PF, LARGE = 1, 2, 3, 4, 100
SCAN, PROBE, TAKE = 0, 1, 2


def plan(has_dfa=1, pf=1, lw=1, pfx=0, min_len=4, want=0, routing=1, probe_skip=0, route_hint=0, span=1 << 30, first_large=0, lw_full=0):
    L = ac.load_test_hooks()
    facts = (C.c_uint64 * 8)(has_dfa, pf, lw, pfx, min_len, want, routing, lw_full)
    hints = (C.c_int32 * 2)(probe_skip, route_hint)
    out = (C.c_uint32 * 3)()
    assert L.acgpu_test_engine_plan(facts, hints, C.c_uint64(span), first_large, out) == 0
    return tuple(out)


@pytest.mark.parametrize("facts,want", [
    # automatic choice: prefix filter first, handed to the LDS walk / the large-set filter / the global DFA walk
    (dict(), (PF, LDS, SCAN)),
    (dict(lw=0, pfx=1), (PF, LARGE, SCAN)),
    (dict(lw=0, pfx=0), (PF, DFA, SCAN)),
    (dict(routing=0), (PF, 0, SCAN)),
    # small automata (one row per state in LDS, match lists beside them): the LDS walk first, whatever the hints say
    (dict(lw_full=1), (LDS, 0, SCAN)),
    (dict(lw_full=1, route_hint=8, probe_skip=3), (LDS, 0, SCAN)),
    (dict(lw_full=1, min_len=0), (PF, DFA, SCAN)),      # an empty pattern: every state is a match state
    (dict(lw_full=1, want=3), (PF, 0, SCAN)),
    (dict(lw_full=1, want=1), (DFA, 0, SCAN)),
    # no prefix-filter tables (an empty pattern, > 131 072 patterns): the walks; the LDS walk is not offered with an empty pattern
    (dict(pf=0), (LDS, 0, SCAN)),
    (dict(pf=0, min_len=0), (DFA, 0, SCAN)),
    (dict(pf=0, lw=0), (DFA, 0, SCAN)),
    (dict(has_dfa=0, pf=0, lw=0), (CNFA, 0, SCAN)),
    # explicit requests are kept or refused, never routed
    (dict(want=1), (DFA, 0, SCAN)),
    (dict(want=1, has_dfa=0), (CNFA, 0, SCAN)),
    (dict(want=2), (LDS, 0, SCAN)),
    (dict(want=2, min_len=0), (LDS, 0, SCAN)),
    (dict(want=2, lw=0), (0, 0, SCAN)),
    (dict(want=3), (PF, 0, SCAN)),
    (dict(want=3, pf=0), (0, 0, SCAN)),
    (dict(want=3, has_dfa=0), (0, 0, SCAN)),
    # hints: recent scans were abandoned -> the probe decides; four probes in a row chose the alternative -> 32 searches take it unasked
    (dict(route_hint=8), (PF, LDS, PROBE)),
    (dict(route_hint=8, probe_skip=32), (PF, LDS, TAKE)),
    (dict(lw=0, pfx=1, probe_skip=5), (PF, LARGE, TAKE)),
    (dict(route_hint=8, span=(16 << 20) - 1), (PF, LDS, SCAN)),          # small shards: an abandoned pass costs less than a probe
    (dict(route_hint=8, probe_skip=3, first_large=1), (PF, LDS, SCAN)),  # the first kernel already is the large-set filter
    (dict(route_hint=8, routing=0), (PF, 0, SCAN)),
    (dict(route_hint=8, want=3), (PF, 0, SCAN)),
])
def test_plan_table(facts, want):
    assert plan(**facts) == want, facts


def test_deterministic_routing_is_the_hint_free_row():
    """acgpu_config.deterministic_routing makes every hint read 0: the plan of a call is the SCAN row of its facts."""
    for facts in (dict(), dict(lw=0, pfx=1), dict(pf=0)):
        assert plan(**facts)[2] == SCAN


def test_event_order_bucket_size_follows_the_event_count():
    """device/event_order.hip: 2 KiB buckets of end positions, or larger ones when the events are few for the span (config 5:
    45 k occurrences in 8 GiB paid for four million buckets) -- never fewer than max_events / 4 buckets, 2 KiB again from 2^31
    records on (a bucket's record count is a 32-bit word)."""
    L = ac.load_test_hooks()
    L.acgpu_test_event_order_shift.restype = C.c_uint32
    L.acgpu_test_event_order_shift.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    shift = L.acgpu_test_event_order_shift
    assert shift(45_000, 45_000, 8 << 30) == 19            # config 5: 16 Ki buckets of 512 KiB
    assert shift(1_032_356, 1_032_356, 1 << 30) == 12      # sherlock / words-5000: one event per KiB -> 4 KiB buckets
    assert shift(1_586_016, 1_586_016, 1 << 30) == 11      # en-huge / words-15000: 2 KiB
    assert shift((8 << 30) // 64, 1 << 26, 8 << 30) == 11  # the enqueue-only form's capacity: 2 KiB buckets
    assert shift(20_000, 1 << 31, 8 << 30) == 11           # too many records for large buckets
    assert shift(1, 1, 1 << 40) == 24                      # capped at 16 MiB
    assert shift(0, 0, 0) == 11 and shift(5, 5, 100) == 11
    import random
    rng = random.Random(5)
    for _ in range(2000):
        span = rng.randrange(1, 1 << rng.randrange(1, 40))
        ev = rng.randrange(1, 1 << rng.randrange(1, 27))
        sh = shift(ev, ev, span)
        assert 11 <= sh <= 24
        if sh > 11:
            assert (span >> sh) * 4 >= ev, (span, ev, sh)               # at least ev / 4 buckets
        if sh < 24:
            assert (span >> (sh + 1)) * 4 < ev, (span, ev, sh)          # ... and no larger bucket would do
