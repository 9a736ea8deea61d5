"""The chunk-parallel forms of the oracle (what the full-size GPU parity tests compare against) equal its sequential
reference loops: ordered records, order-sensitive hash, every automaton kind, odd thread counts, sub-spans."""
import numpy as np
import pytest

from oracle import orc


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("pattern", "start", "end"))


@pytest.mark.parametrize("kind", [orc.KIND_DFA, orc.KIND_CNFA, orc.KIND_NNFA])
def test_parallel_records_equal_sequential(kind):
    pats = [p[: 2 + i % 7] for i, p in enumerate(orc.gen_patterns(300, seed=0xAC01, lo=0x61, span=4))]
    hay = orc.gen_haystack(0, 1 << 17, seed=0xAC02, lo=0x61, span=4)
    o = orc.Oracle(pats, kind=kind)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert len(want) > 5000
    for t in (1, 2, 3, 7, 64, 1000):
        got, h = o.find_overlapping_parallel(hay, threads=t)
        assert same(got, want), t
        assert h == orc.hash_matches(want)
    span = (1234, 100001)
    got, h = o.find_overlapping_parallel(hay, threads=5, span=span)
    assert same(got, o.find_overlapping_iter(hay, span=span, as_numpy=True))
    # the hash is order-sensitive
    if len(want) > 2 and tuple(want[0]) != tuple(want[1]):
        sw = want.copy()
        sw[[0, 1]] = sw[[1, 0]]
        assert orc.hash_matches(sw) != orc.hash_matches(want)


def test_parallel_hash_equals_the_count_loop_hash():
    pats = orc.gen_patterns(1000, seed=0xAC01)
    hay = orc.gen_haystack(0, 1 << 22, seed=0xAC02)
    for k, pos in enumerate(range(4090, len(hay) - 64, 262139)):
        p = np.frombuffer(pats[k % len(pats)], dtype=np.uint8)
        hay[pos:pos + len(p)] = p
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    cnt, h_seq = o.dfa_overlapping_count(hay)
    got, h = o.find_overlapping_parallel(hay, threads=6)
    assert cnt == len(got) > 10 and h == h_seq


def test_parallel_refuses_what_the_seam_rule_does_not_cover():
    with pytest.raises(orc.OracleError) as e:
        orc.Oracle([b"", b"a"]).find_overlapping_parallel(b"aaaa", threads=2)
    assert e.value.kind == "UnsupportedEmpty"
    with pytest.raises(orc.OracleError) as e:
        orc.Oracle([b"a"], match_kind=orc.LEFTMOST_FIRST).find_overlapping_parallel(b"aaaa", threads=2)
    assert e.value.kind == "UnsupportedOverlapping"
    assert len(orc.Oracle([])  .find_overlapping_parallel(b"aaaa", threads=2)[0]) == 0


def test_parallel_generator_equals_sequential():
    a = orc.gen_haystack(12345, (64 << 20) + 77, seed=0xAC02)      # takes the pthread path
    b = np.empty_like(a)
    orc.lib().orc_gen_haystack(b.ctypes.data, 12345, len(b), 0xAC02, 0x20, 95)
    assert np.array_equal(a, b)
