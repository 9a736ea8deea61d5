"""Helpers shared by the -m gpu parity tests."""
import numpy as np

import aho_corasick_amd as ac
from oracle import orc

MK = {"standard": 0, "leftmost_first": 1, "leftmost_longest": 2}
KIND = {None: None, "nnfa": ac.AhoCorasickKind.NoncontiguousNFA, "cnfa": ac.AhoCorasickKind.ContiguousNFA,
        "dfa": ac.AhoCorasickKind.DFA}
OKIND = {None: orc.KIND_AUTO, "nnfa": orc.KIND_NNFA, "cnfa": orc.KIND_CNFA, "dfa": orc.KIND_DFA}
SK = {"both": 0, "unanchored": 1, "anchored": 2}


def build_pair(pats, mk="standard", kw=None, chunk=0, engine="auto", variants=None):
    """variants: {name: value} engine variants of the automaton (acgpu_set_variant) -- the explicit forms tests force."""
    kw = kw or {}
    b = ac.AhoCorasick.builder().match_kind(MK[mk]).start_kind(SK[kw.get("start_kind", "unanchored")]) \
        .kind(KIND[kw.get("kind")]).ascii_case_insensitive(kw.get("ascii_case_insensitive", False)) \
        .byte_classes(kw.get("byte_classes", True)).prefilter(kw.get("prefilter", True)) \
        .gpu_chunk_bytes(chunk).gpu_engine(engine)
    if kw.get("dense_depth") is not None:
        b.dense_depth(kw["dense_depth"])
    for name, value in {**(kw.get("variants") or {}), **(variants or {})}.items():
        b.gpu_variant(name, value)
    a = b.build(pats)
    o = orc.Oracle(pats, match_kind=MK[mk], start_kind=SK[kw.get("start_kind", "unanchored")],
                   kind=OKIND[kw.get("kind")], ascii_case_insensitive=kw.get("ascii_case_insensitive", False),
                   byte_classes=kw.get("byte_classes", True), dense_depth=kw.get("dense_depth"))
    return a, o


def triples(arr):
    return [(int(p), int(s), int(e)) for p, s, e in zip(arr["pattern"], arr["start"], arr["end"])]


def assert_same(got, want, ctx=""):
    """got/want: numpy MATCH_DTYPE arrays; bit-exact including order."""
    assert len(got) == len(want), f"{ctx}: {len(got)} matches vs oracle {len(want)}"
    for f in ("pattern", "start", "end"):
        if not np.array_equal(got[f], want[f]):
            i = int(np.nonzero(got[f] != want[f])[0][0])
            raise AssertionError(f"{ctx}: first difference at match {i}: got {tuple(got[i])} want {tuple(want[i])}")


def plant(hay, pats, positions):
    """Overwrite hay (numpy uint8) with pats[i % n] at each position (clipped)."""
    for i, pos in enumerate(positions):
        p = np.frombuffer(pats[i % len(pats)], dtype=np.uint8)
        if pos < 0 or pos + len(p) > len(hay):
            continue
        hay[pos:pos + len(p)] = p
    return hay
