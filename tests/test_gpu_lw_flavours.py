"""-m gpu: every table flavour of the LDS walk (device/lds_walk.hip: one row per state | narrow | wide handles) under
both class forms (LDS map | computed clamp), forced through the engine variants lw_flavour / lw_cls (acgpu_set_variant),
against the oracle's ordered overlapping stream.  Host model of the same tables: tests/test_lw_tables.py."""
import numpy as np
import pytest
import torch

from gpu_util import assert_same, build_pair, plant
from oracle import orc

pytestmark = pytest.mark.gpu

FLAVOURS = {"narrow": "0", "wide": "1", "full": "2"}
CLS = {"lds": "0", "computed": "1"}


def forced(pats, flavour, cls, chunk=0, kw=None):
    variants = {}
    if flavour is not None:
        variants["lw_flavour"] = int(FLAVOURS[flavour])
    if cls is not None:
        variants["lw_cls"] = int(CLS[cls])
    a, o = build_pair(pats, "standard", dict({"kind": "dfa"}, **(kw or {})), chunk=chunk, engine="hot", variants=variants)
    a.upload()   # the variants are read here
    return a, o


def check(pats, hay, flavour, cls, chunk=0, kw=None, must_fit=True):
    try:
        a, o = forced(pats, flavour, cls, chunk, kw)
        got = a.find_overlapping_iter(torch.from_numpy(hay).cuda(), as_numpy=True)
    except Exception as e:   # the forced form does not fit this automaton: the engine is refused, nothing falls back
        assert not must_fit and "invalid argument" in str(e).lower(), (flavour, cls, e)
        return None
    want = o.find_overlapping_iter(hay, as_numpy=True)
    assert_same(got, want, f"lw {flavour}/{cls} chunk={chunk}")
    return len(want)


@pytest.mark.parametrize("cls", ["lds", "computed"])
@pytest.mark.parametrize("chunk", [0, 4096])
def test_headline_set_narrow(cls, chunk):
    pats = orc.gen_patterns(1000, seed=0xAC01)
    n = 8 << 20   # most wave tasks interior (fast step, inline counts), first / last regions on the edge walk
    hay = orc.gen_haystack(0, n, seed=0xAC02)
    plant(hay, pats, [0, n - 16, n - 5] + [4090 + 65521 * k for k in range(120)])
    hay[1 << 20:(1 << 20) + 4096] = np.frombuffer(bytes(range(256)) * 16, dtype=np.uint8)   # every byte value, both clamp sides
    assert check(pats, hay, "narrow", cls, chunk) > 100


@pytest.mark.parametrize("flavour", ["narrow", "wide", None])
@pytest.mark.parametrize("cls", ["lds", "computed"])
def test_az_set_match_dense(flavour, cls):
    pats = orc.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
    hay = orc.gen_haystack(0, 8 << 20, seed=0xAC02, lo=0x61, span=26)
    hay[12345:12345 + 512] = np.frombuffer(bytes(range(256)) * 2, dtype=np.uint8)
    n = check(pats, hay, flavour, cls, must_fit=flavour != "narrow")   # 254 rows: the exception chains outgrow the 16 K slots
    assert n is None or n > 1000


@pytest.mark.parametrize("flavour", ["full", "narrow", "wide", None])
@pytest.mark.parametrize("cls", ["lds", "computed"])
def test_small_sets_every_byte_matches(flavour, cls):
    """The reference's teddy / memchr / same families in miniature: a handful of 1-4 byte patterns, nested and
    duplicated, over a haystack in which most bytes end one or more of them."""
    rng = np.random.default_rng(31)
    for case in range(3):
        alphabet = np.array([0x61, 0x62, 0x63, 0x65, 0x74, 0x20][: 3 + case], dtype=np.uint8)
        pats = [bytes(rng.choice(alphabet, size=int(rng.integers(1, 5)))) for _ in range(4 + 12 * case)] + [b"a", b"a"]
        hay = rng.choice(np.concatenate([alphabet, np.array([0x0A, 0x7A, 0xC3], dtype=np.uint8)]), size=4 << 20).astype(np.uint8)
        assert check(pats, hay, flavour, cls) > 100_000


@pytest.mark.parametrize("cls", ["lds", "computed"])
def test_full_flavour_on_natural_text(cls):
    import corpora
    pats = [b"Sherlock", b"Holmes", b"Watson", b"the", b"he", b"e", b"\n"]
    hay = np.tile(corpora.haystack("sherlock.txt"), 8)
    assert check(pats, hay, "full", cls) > 100_000


def test_forms_that_do_not_fit_are_refused():
    pats = orc.gen_patterns(1000, seed=0xAC01)
    hay = orc.gen_haystack(0, 1 << 16, seed=5)
    assert check(pats, hay, "full", None, must_fit=False) is None          # 9 287 rows do not fit LDS
    assert check(pats, hay, "wide", None, must_fit=False) is None          # 96 classes > 64
    assert check(pats, hay, None, "computed", kw={"ascii_case_insensitive": True}, must_fit=False) is None
