"""-m gpu: one-shot searches (acgpu_find / acgpu_is_match) through the parallel windowed path vs the oracle's
try_find_fwd (src/automaton.rs:1259-1420): first matches far into large haystacks, at window seams, and where the
leftmost rule needs bytes beyond the window."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import build_pair
from oracle import orc

pytestmark = pytest.mark.gpu
W0 = 16 << 20   # first window of find_parallel


def same(a, o, hay, dh, span=None):
    inp = ac.Input(dh)
    if span is not None:
        inp.range(*span)
    got = a.find(inp)
    want = o.find(hay, span=span)
    assert (got is None) == (want is None), (got, want, span)
    if got is not None:
        assert got.as_tuple() == tuple(want), (got, want, span)
    assert a.is_match(inp) == (want is not None)


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
def test_find_far_and_at_window_seams(mk):
    n = 40 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC09, lo=0x30, span=10)          # digits: no letter pattern occurs
    pats = [b"needle", b"needles", b"need", b"eedlesx", b"dle", b"xyzzyxyzzyxyzzy"]
    a, o = build_pair(pats, mk)
    dh = torch.from_numpy(hay).cuda()
    same(a, o, hay, dh)                                                    # no match at all: scans everything
    for pos in (W0 - 3, W0 - 7, W0, W0 + 1, 2 * W0 + 11, n - 7, 123):  # straddling / just past the first seam, far, end
        h2 = hay.copy()
        h2[pos:pos + 7] = np.frombuffer(b"needles", dtype=np.uint8)
        d2 = torch.from_numpy(h2).cuda()
        same(a, o, h2, d2)
        same(a, o, h2, d2, span=(pos - 50, min(n, pos + 5)))               # span cuts the longer alternatives
        same(a, o, h2, d2, span=(pos + 1, n))                              # starts inside the occurrence
    # the leftmost choice needs bytes beyond the first window: "need" ends inside, "needles" ends outside
    h3 = hay.copy()
    h3[W0 - 4:W0 + 3] = np.frombuffer(b"needles", dtype=np.uint8)
    same(a, o, h3, torch.from_numpy(h3).cuda())


def test_find_dense_small():
    rng = np.random.default_rng(17)
    for case in range(60):
        sigma = int(rng.integers(2, 5))
        pats = [bytes(rng.integers(0x61, 0x61 + sigma, size=int(rng.integers(1, 6)), dtype=np.uint8))
                for _ in range(int(rng.integers(1, 9)))]
        mk = ["standard", "leftmost_first", "leftmost_longest"][case % 3]
        a, o = build_pair(pats, mk, {"kind": [None, "dfa", "cnfa", "nnfa"][case % 4]})
        hay = rng.integers(0x61, 0x61 + sigma + 2, size=int(rng.integers(0, 400)), dtype=np.uint8)
        dh = torch.from_numpy(hay).cuda() if len(hay) else torch.zeros(0, dtype=torch.uint8, device="cuda")
        same(a, o, hay, dh)
        if len(hay) > 4:
            s = int(rng.integers(0, len(hay)))
            e = int(rng.integers(s, len(hay) + 1))
            same(a, o, hay, dh, span=(s, e))
