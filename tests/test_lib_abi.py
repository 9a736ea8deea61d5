"""The C-ABI library loads, exports every symbol include/acgpu.h declares, builds automata on the CPU,
and fails loudly (no CPU fallback) when asked to search without a GPU."""
import ctypes as C
import os
import re

import pytest

import aho_corasick_amd as ac
from aho_corasick_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "acgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(acgpu_[a-z_0-9]+)\s*\(", src)))


def test_exports_every_declared_symbol():
    L = ac.load_library()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in acgpu.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms
    assert L.acgpu_abi_version() == 3


def test_test_hooks_live_outside_the_product_library():
    """include/acgpu_test.h is not part of the product ABI: libacgpu.so exports none of its symbols,
    libacgpu_testhooks.so (which links libacgpu.so) all of them."""
    src = open(os.path.join(ROOT, "include", "acgpu_test.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = sorted(set(re.findall(r"\b(acgpu_test_[a-z_0-9]+)\s*\(", src)))
    assert declared == sorted(_lib.TEST_SYMBOLS) and len(declared) >= 6
    product = C.CDLL(os.path.join(ROOT, "aho-corasick_amd", "lib", "libacgpu.so"))
    for s in declared:
        assert not hasattr(product, s), f"{s} leaked into libacgpu.so"
    hooks = ac.load_test_hooks()
    for s in declared:
        assert hasattr(hooks, s)


def test_config_defaults():  # AhoCorasickBuilder::new, dfa.rs:395-403, contiguous.rs:900-908
    L = ac.load_library()
    cfg = _lib.Config()
    L.acgpu_config_init(C.byref(cfg))
    assert (cfg.match_kind, cfg.start_kind, cfg.kind) == (0, 1, 0)
    assert cfg.ascii_case_insensitive == 0 and cfg.byte_classes == 1 and cfg.prefilter == 1
    assert cfg.dense_depth_set == 0


def test_build_and_getters_on_cpu():
    a = ac.AhoCorasick.new([b"append", b"appendage", b"app"])
    assert a.kind() == ac.AhoCorasickKind.DFA  # <= 100 patterns: build_auto picks the DFA
    assert a.match_kind() == ac.MatchKind.Standard and a.start_kind() == ac.StartKind.Unanchored
    assert (a.patterns_len(), a.min_pattern_len(), a.max_pattern_len()) == (3, 3, 9)
    assert a.memory_usage() > 0
    big = ac.AhoCorasick.new([bytes([65 + i % 26, 65 + i // 26, 66]) for i in range(101)])
    assert big.kind() == ac.AhoCorasickKind.ContiguousNFA  # > 100 patterns
    both = ac.AhoCorasick.builder().start_kind(ac.StartKind.Both).build([b"a"])
    assert both.kind() == ac.AhoCorasickKind.ContiguousNFA  # StartKind::Both never auto-picks the DFA
    forced = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.NoncontiguousNFA).build([b"a"])
    assert forced.kind() == ac.AhoCorasickKind.NoncontiguousNFA


def test_match_and_input_types():
    m = ac.Match.must(2, (3, 7))
    assert (m.pattern(), m.start(), m.end(), m.len(), m.is_empty()) == (2, 3, 7, 4, False)
    assert m == (2, 3, 7)
    i = ac.Input(b"foobar")
    assert i.get_span() == (0, 6) and not i.is_done()
    i.range(6, 6)
    assert not i.is_done()
    i.range(7, 6)
    assert i.is_done()
    with pytest.raises(ValueError):
        ac.Input(b"foobar").range(0, 7)
    with pytest.raises(ValueError):
        ac.Input(b"foobar").range(5, 3)


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful without a GPU")
def test_search_fails_loudly_without_gpu():
    a = ac.AhoCorasick.new([b"abc"])
    with pytest.raises(RuntimeError) as e:
        list(a.find_overlapping_iter(b"xxabcxx"))
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(RuntimeError):
        list(a.find_iter(b"xxabcxx"))
    with pytest.raises(RuntimeError):
        a.is_match(b"abc")


def test_argument_errors_precede_device_work():
    # these are decided on the host exactly where the reference decides them
    a = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build([])
    with pytest.raises(ac.MatchError) as e:
        a.find_overlapping_iter(b"")
    assert e.value.kind == "UnsupportedOverlapping"  # automaton.rs:404-408
    a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build([b"foo"])
    with pytest.raises(ac.MatchError) as e:
        a.try_find(ac.Input(b"foo").anchored(ac.Anchored.Yes))
    assert e.value.kind == "InvalidInputAnchored"  # ahocorasick.rs:2778-2789
    a = ac.AhoCorasick.builder().start_kind(ac.StartKind.Anchored).build([b"foo"])
    with pytest.raises(ac.MatchError) as e:
        a.try_find(b"foo")
    assert e.value.kind == "InvalidInputUnanchored"
    a = ac.AhoCorasick.new([b"foo"])
    with pytest.raises(ac.MatchError) as e:
        a.try_find_overlapping_iter(ac.Input(b"foo").anchored(ac.Anchored.Yes))
    assert e.value.kind == "InvalidInputAnchored"


def test_workload_generator_matches_the_oracle_generator():
    """bench.py builds its pattern sets with the product's own generator (aho_corasick_amd.workload); the tests'
    oracle restates the same Appendix-C generator in C -- they must agree byte for byte."""
    import numpy as np
    import aho_corasick_amd as ac
    from aho_corasick_amd.workload import gen_haystack_host
    from oracle import orc
    for seed, n in ((0xAC01, 1000), (0xAC04, 3000), (7, 33)):
        assert ac.gen_patterns(n, seed=seed) == orc.gen_patterns(n, seed=seed)
    assert ac.gen_patterns(50, seed=9, lo=0x61, span=26) == orc.gen_patterns(50, seed=9, lo=0x61, span=26)
    assert np.array_equal(gen_haystack_host(12345, 70000), orc.gen_haystack(12345, 70000))


def test_engine_variants_are_per_automaton_and_named():
    """acgpu_set_variant (include/acgpu.h): explicit forms of the device engines for tests and A/B runs -- state of ONE
    automaton (and of the automata searched on its behalf), refused for unknown names; the library reads no environment
    variable for them."""
    a = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).gpu_variant("find_iter_start_table", 1).build([b"ab", b"b"])
    a.set_variant("lw_cls", 0).set_variant("pfx_min_patterns", 1)
    with pytest.raises(Exception) as e:
        a.set_variant("no_such_variant", 1)
    assert "unknown variant" in str(e.value)
    import re
    srcs = []
    root = os.path.join(ROOT, "aho-corasick_amd", "csrc")
    for d, _, files in os.walk(root):
        srcs += [os.path.join(d, f) for f in files if f.endswith((".cpp", ".hip", ".hpp"))]
    names = sorted({m for p in srcs for m in re.findall(r'getenv\("(ACGPU_[A-Z0-9_]+)"\)', open(p).read())})
    assert names == ["ACGPU_GUARD_SHRINK", "ACGPU_HOST_PIECE_MIB", "ACGPU_MULTI_FORCE_RCCL", "ACGPU_MULTI_NO_RCCL"], names
