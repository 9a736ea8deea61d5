"""BASELINE.json configs[0] (C1): the reference's own ten literals "abcdef" ... "jklmno"
(benchmarks/definitions/random/misc.toml:37-41) over 1 MiB of a-z text (the SURVEY.md generator, seed 1),
MatchKind::Standard overlapping.  Not-gpu: the oracle's DFA / contiguous-NFA / noncontiguous-NFA streams against a
sliding-window count written with numpy; gpu: every engine against the oracle."""
import numpy as np
import pytest

from oracle import orc

PATS = [bytes(range(0x61 + i, 0x61 + i + 6)) for i in range(10)]   # abcdef, bcdefg, ..., jklmno


def c1_haystack():
    hay = orc.gen_haystack(0, 1 << 20, seed=1, lo=0x61, span=26)
    for k in range(200):   # (random a-z text holds a given 6-gram once per 3*10^8 bytes: plant the reference's patterns)
        p = np.frombuffer(PATS[k % 10], dtype=np.uint8)
        at = 5003 * k + 17
        hay[at:at + 6] = p
    hay[300000:300015] = np.frombuffer(b"abcdefghijklmno", dtype=np.uint8)   # all ten, overlapping
    return hay


def windows_equal(hay, p):
    v = np.lib.stride_tricks.sliding_window_view(hay, len(p))
    return np.nonzero((v == np.frombuffer(p, dtype=np.uint8)).all(axis=1))[0]


def test_c1_oracle_against_sliding_windows():
    hay = c1_haystack()
    want = sorted((int(s) + 6, i, int(s)) for i, p in enumerate(PATS) for s in windows_equal(hay, p))
    assert len(want) >= 210
    for kind in (orc.KIND_DFA, orc.KIND_CNFA, orc.KIND_NNFA):
        got = orc.Oracle(PATS, kind=kind).find_overlapping_iter(hay, as_numpy=True)
        assert [(int(e), int(p), int(s)) for p, s, e in zip(got["pattern"], got["start"], got["end"])] == want


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["auto", "pf", "hot", "walk"])
def test_c1_every_engine(engine):
    import torch
    from gpu_util import assert_same, build_pair
    hay = c1_haystack()
    for kind in ("dfa", "cnfa"):
        if kind == "cnfa" and engine in ("pf", "hot"):
            continue
        a, o = build_pair(PATS, "standard", {"kind": kind}, engine=engine)
        want = o.find_overlapping_iter(hay, as_numpy=True)
        assert_same(a.find_overlapping_iter(torch.from_numpy(hay).cuda(), as_numpy=True), want, f"C1 {kind} {engine}")
        assert_same(a.find_overlapping_iter(hay, as_numpy=True), want, f"C1 {kind} {engine}, host haystack")
