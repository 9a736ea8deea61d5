"""-m gpu: acgpu_replace_all (device segmented copy) vs the oracle's restatement of
Automaton::try_replace_all_with_bytes (src/automaton.rs:530-550)."""
import numpy as np
import pytest
import torch

import aho_corasick_amd as ac
from gpu_util import build_pair, plant
from oracle import orc

pytestmark = pytest.mark.gpu


def test_reference_doc_examples():
    a = ac.AhoCorasick.new(["fox", "brown", "quick"])                         # src/ahocorasick.rs:163-175
    assert a.replace_all("The quick brown fox.", ["sloth", "grey", "slow"]) == "The slow grey sloth."
    a = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build(["append", "appendage", "app"])
    assert a.replace_all("append the app to the appendage", ["x", "y", "z"]) == "x the z to the xage"   # :636-650
    assert a.replace_all_bytes(b"append the app to the appendage", ["x", "y", "z"]) == b"x the z to the xage"  # :680-692
    with pytest.raises(ValueError):
        a.replace_all("append", ["x"])                                        # src/automaton.rs:442-447


def test_closure_forms():
    a = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build(["append", "appendage", "app"])
    dst = []                                                                  # src/ahocorasick.rs:736-746
    a.replace_all_with("append the app to the appendage", dst, lambda m, t, d: d.append(str(m.pattern())) or True)
    assert "".join(dst) == "0 the 2 to the 0age"
    dst = []                                                                  # stop early, :755-763
    a.replace_all_with("append the app to the appendage", dst, lambda m, t, d: d.append(str(m.pattern())) or False)
    assert "".join(dst) == "0 the app to the appendage"
    out = bytearray()
    a.replace_all_with_bytes(b"append the app to the appendage", out, lambda m, t, d: d.extend(t.upper()) or True)
    assert bytes(out) == b"APPEND the APP to the APPENDage"


def test_utf8_boundary_rule():
    a = ac.AhoCorasick.new([b"\xa9", b"a"])
    assert a.replace_all_bytes("aéa".encode(), ["X", "Y"]) == b"Y\xc3XY"
    assert a.replace_all("aéa", ["X", "Y"]) == "YéY"


@pytest.mark.parametrize("mk", ["standard", "leftmost_first", "leftmost_longest"])
def test_replace_matches_oracle(mk):
    n = 1 << 20
    hay = orc.gen_haystack(0, n, seed=0xAC06, lo=0x61, span=26)
    pats = [p[:k] for p, k in zip(orc.gen_patterns(300, seed=9, lo=0x61, span=26), [3, 4, 5, 6] * 75)]
    a, o = build_pair(pats, mk)
    rng = np.random.default_rng(5)
    repl = [bytes(rng.integers(0x41, 0x5B, size=int(rng.integers(0, 12)), dtype=np.uint8)) for _ in pats]
    want = orc.replace_all_bytes(o, hay, repl)
    assert want != hay.tobytes()
    assert a.replace_all_bytes(hay, repl) == want                              # host in, host out
    got = a.replace_all_bytes(torch.from_numpy(hay).cuda(), repl)              # device in, device out
    assert got.is_cuda and bytes(got.cpu().numpy()) == want
    for m in (1, 5, 17):                                                        # misaligned device haystack
        buf = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        buf[m:m + n] = torch.from_numpy(hay).cuda()
        assert bytes(a.replace_all_bytes(buf[m:m + n], repl).cpu().numpy()) == want
    # only deletions / only growth / nothing to do
    assert a.replace_all_bytes(hay, [b""] * len(pats)) == orc.replace_all_bytes(o, hay, [b""] * len(pats))
    big = [b"<" + p + b">" * 20 for p in pats]
    assert a.replace_all_bytes(hay, big) == orc.replace_all_bytes(o, hay, big)
    none = orc.gen_haystack(0, 5000, seed=1, lo=0x30, span=10)
    assert a.replace_all_bytes(none, repl) == none.tobytes()
    assert a.replace_all_bytes(b"", repl) == b""


def test_replace_dense_and_tiny():
    rng = np.random.default_rng(11)
    for case in range(40):
        sigma = int(rng.integers(2, 5))
        pats = [bytes(rng.integers(0x61, 0x61 + sigma, size=int(rng.integers(1, 5)), dtype=np.uint8))
                for _ in range(int(rng.integers(1, 12)))]
        mk = ["standard", "leftmost_first", "leftmost_longest"][case % 3]
        a, o = build_pair(pats, mk)
        hay = rng.integers(0x61, 0x61 + sigma + 1, size=int(rng.integers(0, 3000)), dtype=np.uint8)
        repl = [bytes(rng.integers(0x41, 0x5B, size=int(rng.integers(0, 6)), dtype=np.uint8)) for _ in pats]
        assert a.replace_all_bytes(hay, repl) == orc.replace_all_bytes(o, hay, repl), f"case {case} {mk} {pats}"
