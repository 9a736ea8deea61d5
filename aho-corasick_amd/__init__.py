"""acgpu: MI355X-native Aho-Corasick search (drop-in for the search path of BurntSushi/aho-corasick).

Host-side mirror of the reference facade (src/ahocorasick.rs): AhoCorasick, AhoCorasickBuilder,
MatchKind, StartKind, AhoCorasickKind, Match, Input -- same names, argument meaning and error
behaviour -- on top of the C ABI in include/acgpu.h (lib/libacgpu.so, hand-written HIP kernels).
There is no CPU search path: without the HIP library / a GPU the search calls raise.
"""
from .api import (AhoCorasick, AhoCorasickBuilder, AhoCorasickKind, Anchored, BuildError, Input, Match,  # noqa: F401
                  MatchError, MatchKind, StartKind, gen_haystack, stream_read_gbps, MATCH_DTYPE)
from ._lib import build_library, library_path, load_library, load_test_hooks  # noqa: F401
from .workload import gen_patterns  # noqa: F401

__all__ = ["AhoCorasick", "AhoCorasickBuilder", "AhoCorasickKind", "Anchored", "BuildError", "Input", "Match",
           "MatchError", "MatchKind", "StartKind", "gen_haystack", "stream_read_gbps", "gen_patterns", "build_library", "library_path", "load_library", "load_test_hooks",
           "MATCH_DTYPE"]
