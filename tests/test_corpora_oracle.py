"""The reference's benchmark inputs (tests/golden/corpora) through the oracle: fixtures intact, the chunk-parallel form
equals the sequential reference loop on natural text, and a naive matcher agrees on a sample."""
import numpy as np
import pytest

import corpora
from oracle import orc


def test_manifest_and_shapes():
    assert len(corpora.words("words-100")) == 100 and len(corpora.words("words-5000")) == 5000
    assert len(corpora.words("words-15000")) == 15000 and len(corpora.words("dictionary-15")) == 2663
    assert len(corpora.haystack("sherlock.txt")) == 594915 and len(corpora.haystack("en-huge.txt")) == 613357
    assert len(corpora.haystack("sherlock.txt", 1 << 20)) == 1 << 20


@pytest.mark.parametrize("pats", ["words-100", "words-5000", "dictionary-15"])
def test_parallel_equals_sequential_on_natural_text(pats):
    ws = corpora.words(pats)
    hay = corpora.haystack("sherlock.txt")
    for kind in (orc.KIND_DFA if len(ws) <= 5000 else orc.KIND_CNFA, orc.KIND_CNFA):
        o = orc.Oracle(ws, kind=kind)
        want = o.find_overlapping_iter(hay, as_numpy=True)
        got, h = o.find_overlapping_parallel(hay, threads=5)
        assert len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in ("pattern", "start", "end"))
        assert h == orc.hash_matches(want)


def test_naive_matcher_agrees_on_a_sample():
    ws = corpora.words("words-5000")
    text = corpora.raw("sherlock.txt")[100000:160000]
    want = sorted((e, -(e - s), p) for p, w in enumerate(ws) for s in _find_all(text, w) for e in [s + len(w)])
    got = orc.Oracle(ws, kind=orc.KIND_CNFA).find_overlapping_iter(text)
    assert [(e, -(e - s), p) for p, s, e in got] == want and len(want) > 20


def _find_all(text, w):
    i = text.find(w)
    while i >= 0:
        yield i
        i = text.find(w, i + 1)
