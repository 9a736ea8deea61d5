"""not-gpu: the contiguous-NFA walk kernel's tables (device/cnfa_walk.hip) -- which states live in LDS, and the patched
copy of the reference's `repr` that names them by slot -- built by the host code the upload uses (csrc/host/cnfa_tables.cpp)
and walked by the kernel's own step on the CPU (acgpu_test_cnfa_host): the count must equal the oracle's contiguous-NFA
overlapping count."""
import ctypes as C

import numpy as np
import pytest

import aho_corasick_amd as ac
from oracle import orc


def walk(pats, hay, **kw):
    b = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA)
    if kw.get("casei"):
        b.ascii_case_insensitive(True)
    if kw.get("byte_classes") is False:
        b.byte_classes(False)
    if kw.get("dense_depth") is not None:
        b.dense_depth(kw["dense_depth"])
    a = b.build(pats)
    L = ac.load_test_hooks()
    L.acgpu_test_cnfa_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    h = np.ascontiguousarray(hay, dtype=np.uint8)
    n, info = C.c_uint64(), (C.c_uint64 * 8)()
    assert L.acgpu_test_cnfa_host(a._h, C.c_void_p(h.ctypes.data), len(h), C.byref(n), info) == 0
    return n.value, dict(served=int(info[0]), slots=int(info[1]), dense_outside=int(info[2]), sorted=int(info[3]),
                         slot_matches=int(info[4]), patched=int(info[5]))


def want(pats, hay, **kw):
    o = orc.Oracle(pats, kind=orc.KIND_CNFA, ascii_case_insensitive=bool(kw.get("casei")),
                   byte_classes=kw.get("byte_classes", True), dense_depth=kw.get("dense_depth"))
    return len(o.find_overlapping_iter(hay, as_numpy=True))


def planted(pats, n, seed, lo=0x20, span=95, every=499):
    hay = orc.gen_haystack(0, n, seed=seed, lo=lo, span=span)
    rng = np.random.default_rng(seed)
    for at in range(3, n - 64, every):
        p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
        hay[at:at + len(p)] = p
    return hay


@pytest.mark.parametrize("npat", [1000, 30000])
def test_random_sets(npat):
    pats = orc.gen_patterns(npat, seed=0xAC04)
    hay = planted(pats, 1 << 19, npat)
    n, info = walk(pats, hay)
    assert info["served"] and info["slots"] == 96 and not info["dense_outside"] and info["sorted"]   # start + its 95 children in LDS, their children second tier
    assert info["patched"] > npat and n == want(pats, hay) > 1000


def test_layout_variants():
    """256 classes (byte_classes off): fewer rows fit LDS, the other dense states become second-tier states; dense_depth
    4 with 20 000 patterns: more dense states than both tiers hold, the speculative load path runs; 1-byte and empty
    patterns: LDS-resident match states; 2-byte patterns: second-tier match states; case-insensitive."""
    pats = orc.gen_patterns(3000, seed=0xAC07)
    hay = planted(pats, 1 << 18, 7)
    n, info = walk(pats, hay, byte_classes=False)
    assert info["served"] and info["slots"] < 96 and n == want(pats, hay, byte_classes=False)
    n, info = walk(pats, hay, dense_depth=3)
    assert info["served"] and not info["dense_outside"] and n == want(pats, hay, dense_depth=3)
    big = orc.gen_patterns(20000, seed=0xAC08)
    hb = planted(big, 1 << 18, 8)
    n, info = walk(big, hb, dense_depth=4)
    assert info["served"] and info["dense_outside"] and n == want(big, hb, dense_depth=4)
    two = [p[:2] for p in pats[:400]] + pats[400:]
    n, info = walk(two, hay)
    assert info["served"] and n == want(two, hay) > 3000
    short = [b"a", b"ab", b"b", b"abc", b"ca", b"", b"bb"]
    h2 = np.frombuffer(b"abcabbacabcbbabca" * 500, dtype=np.uint8).copy()
    n, info = walk(short, h2)
    assert info["served"] and info["slot_matches"] and n == want(short, h2) > len(h2)
    ci = [b"Needle", b"hAy", b"stack", b"NEEDLES", b"x"]
    text = np.frombuffer(b"a needle in a HAYSTACK of NeEdLeS and hay; xX. " * 300, dtype=np.uint8).copy()
    n, info = walk(ci, text, casei=True)
    assert info["served"] and n == want(ci, text, casei=True) > 1500


@pytest.mark.parametrize("seed", range(10))
def test_random_automata(seed):
    rng = np.random.default_rng(2000 + seed)
    asz = int(rng.choice([2, 4, 26, 95, 200]))
    lo = 0x61 if asz <= 26 else (0x20 if asz == 95 else 0x10)
    npat = int(rng.choice([1, 7, 80, 900]))
    pats = []
    for _ in range(npat):
        if pats and rng.random() < 0.25:
            b = pats[int(rng.integers(len(pats)))]
            p = b[: int(rng.integers(1, len(b) + 1))] + bytes(rng.integers(lo, lo + asz, size=int(rng.integers(0, 3)), dtype=np.uint8))
        else:
            p = bytes(rng.integers(lo, lo + asz, size=int(rng.integers(1, 10)), dtype=np.uint8))
        pats.append(p)
    n = 1 << 14
    hay = rng.integers(lo, lo + asz, size=n, dtype=np.uint8)
    for at in range(5, n - 32, 131):
        p = np.frombuffer(pats[int(rng.integers(npat))], dtype=np.uint8)
        hay[at:at + len(p)] = p
    kw = {"byte_classes": bool(rng.random() < 0.7), "dense_depth": int(rng.choice([0, 1, 2, 3]))}
    got, info = walk(pats, hay, **kw)
    if info["served"]:
        assert got == want(pats, hay, **kw), (seed, kw, info)


@pytest.mark.parametrize("words", ["words-100", "words-5000"])
def test_reference_corpora_natural_text(words):
    import corpora
    pats = corpora.words(words)
    hay = corpora.haystack("sherlock.txt")
    n, info = walk(pats, hay)
    assert info["served"] and n == want(pats, hay) >= 10
