"""not-gpu: the contiguous-NFA shallow-skip walk (device/cnfa_tri.hip) -- its tables (trigram bitmap of the depth-3 trie
nodes, their 16-byte child entries, the copy of `repr` with the fail words into depth <= 2 tagged: csrc/host/
cnfa_tri_tables.cpp) and the kernel's OWN per-piece code (device/cnfa_tri_step.hpp, compiled for the host) run lane by
lane over the chunk grid of a search (acgpu_test_cnfa_tri_host): the match count must equal the oracle's contiguous-NFA
overlapping count, and the records the walk's match events stand for (what k_cnfa_tri_emit writes) must equal the oracle's
record list, in order (FNV-1a over pattern, start, end)."""
import ctypes as C

import numpy as np
import pytest

import aho_corasick_amd as ac
from oracle import orc


def fnv(rec):
    h = 0xCBF29CE484222325
    for p, s_, e_ in zip(rec["pattern"].tolist(), rec["start"].tolist(), rec["end"].tolist()):
        for w in (p, s_, e_):
            for i in range(8):
                h = ((h ^ ((w >> (8 * i)) & 0xFF)) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def walk(pats, hay, chunk=None, **kw):
    b = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA)
    if kw.get("casei"):
        b.ascii_case_insensitive(True)
    if kw.get("byte_classes") is False:
        b.byte_classes(False)
    if kw.get("dense_depth") is not None:
        b.dense_depth(kw["dense_depth"])
    if chunk:
        b.gpu_chunk_bytes(chunk)
    a = b.build(pats)
    L = ac.load_test_hooks()
    L.acgpu_test_cnfa_tri_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    h = np.ascontiguousarray(hay, dtype=np.uint8)
    n, info = C.c_uint64(), (C.c_uint64 * 8)()
    assert L.acgpu_test_cnfa_tri_host(a._h, C.c_void_p(h.ctypes.data), len(h), C.byref(n), info) == 0
    return n.value, dict(served=int(info[0]), classes=int(info[1]), bw=int(info[2]), granule=int(info[3]),
                         shallow_matches=int(info[4]), lds=int(info[5]), gathers=int(info[6]), hash=int(info[7]))


def want(pats, hay, **kw):
    o = orc.Oracle(pats, kind=orc.KIND_CNFA, ascii_case_insensitive=bool(kw.get("casei")),
                   byte_classes=kw.get("byte_classes", True), dense_depth=kw.get("dense_depth"))
    r = o.find_overlapping_iter(hay, as_numpy=True)
    return len(r), fnv(r)


def check(pats, hay, chunk=None, **kw):
    n, info = walk(pats, hay, chunk=chunk, **kw)
    wn, wh = want(pats, hay, **kw)
    if info["served"]:
        assert (n, info["hash"]) == (wn, wh), (info, n, wn)
    return n, info


def planted(pats, n, seed, lo=0x20, span=95, every=499):
    hay = orc.gen_haystack(0, n, seed=seed, lo=lo, span=span)
    rng = np.random.default_rng(seed)
    for at in range(3, n - 64, every):
        p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
        hay[at:at + len(p)] = p
    return hay


@pytest.mark.parametrize("npat", [1000, 30000, 100000])
def test_random_sets(npat):
    pats = orc.gen_patterns(npat, seed=0xAC04)
    hay = planted(pats, 1 << 19, npat)
    n, info = check(pats, hay)
    assert info["served"] and info["classes"] == 95 and info["bw"] == 3 and not info["shallow_matches"] and n > 1000
    # what the walk costs in gathers: the point of the kernel (1.14 per byte for the LDS-row walk at 100 000 patterns)
    assert info["gathers"] / len(hay) < {1000: 0.06, 30000: 0.12, 100000: 0.22}[npat]


def test_layout_variants():
    """byte_classes off (256 classes, only those in use count); deeper dense states (dense records below depth 2: the
    walk reads their transition rows); 2-byte, 1-byte and empty patterns (matches in the shallow regime: counted from the
    per-pair table, their records through the pair's state); case-insensitive; odd chunk sizes (seams inside pieces)."""
    pats = orc.gen_patterns(3000, seed=0xAC07)
    hay = planted(pats, 1 << 18, 7)
    assert check(pats, hay, byte_classes=False)[1]["served"]
    assert check(pats, hay, dense_depth=3)[1]["served"]
    assert check(pats, hay, dense_depth=0)[1]["served"]
    big = orc.gen_patterns(20000, seed=0xAC08)
    assert check(big, planted(big, 1 << 18, 8), dense_depth=4)[1]["served"]
    two = [p[:2] for p in pats[:400]] + pats[400:]
    n, info = check(two, hay)
    assert info["served"] and info["shallow_matches"] and n > 3000
    short = [b"a", b"ab", b"b", b"abc", b"ca", b"", b"bb"]
    h2 = np.frombuffer(b"abcabbacabcbbabca" * 500, dtype=np.uint8).copy()
    n, info = check(short, h2)
    assert info["served"] and info["shallow_matches"] and n > len(h2)
    ci = [b"Needle", b"hAy", b"stack", b"NEEDLES", b"x"]
    text = np.frombuffer(b"a needle in a HAYSTACK of NeEdLeS and hay; xX. " * 300, dtype=np.uint8).copy()
    n, info = check(ci, text, casei=True)
    assert info["served"] and n > 1500
    for chunk in (64, 192, 4096):
        assert check(pats, hay[: 1 << 16], chunk=chunk)[1]["served"]
        check(short, h2, chunk=chunk)


@pytest.mark.parametrize("seed", range(16))
def test_random_automata(seed):
    rng = np.random.default_rng(3000 + seed)
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
    _, info = check(pats, hay, chunk=int(rng.choice([64, 128, 2048])), **kw)

    # (served unless the alphabet is too large for the pair tables in LDS: a regression that stops serving small automata
    # would otherwise pass here unnoticed)
    assert info["served"] or asz == 200, (seed, asz, npat, info)

@pytest.mark.parametrize("words", ["words-100", "words-5000"])
def test_reference_corpora_natural_text(words):
    import corpora
    pats = corpora.words(words)
    hay = corpora.haystack("sherlock.txt")
    n, info = check(pats, hay)
    assert info["served"] and n >= 10


def test_long_case_insensitive_pattern_builds_in_linear_time():
    """Under ascii_case_insensitive every trie node is reached through two byte classes ('a' and 'A'): a traversal that
    follows every edge visits a 160-byte pattern's states 2^160 times (a hang found on the GPU box by
    tests/test_gpu_dfa_fill.py).  The table builders visit states, not paths."""
    pats = [b"a", b"ab", b"abc", b"bc", b"c", b"abcd" * 40]
    hay = orc.gen_haystack(0, 20000, seed=5, lo=0x61, span=4)
    n, info = check(pats, hay, casei=True)
    assert info["served"] and n > 10000


def test_alphabet_too_large_is_refused():
    """200 classes in use: the pair tables would not fit LDS; the kernel must say so (the LDS-row walk serves)."""
    rng = np.random.default_rng(9)
    pats = [bytes(rng.integers(0, 256, size=6, dtype=np.uint8)) for _ in range(4000)]
    hay = rng.integers(0, 256, size=1 << 14, dtype=np.uint8)
    n, info = walk(pats, hay)
    assert not info["served"]
