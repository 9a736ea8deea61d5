"""CPU: the oracle's replace_all restatement against the reference's documented examples
(src/ahocorasick.rs:163-175, :636-650, :680-692; README.md:85-99)."""
from oracle import orc

LEFTMOST_FIRST = 1


def test_doc_example_struct_level():
    # src/ahocorasick.rs:163-175
    o = orc.Oracle([b"fox", b"brown", b"quick"])
    assert orc.replace_all_bytes(o, b"The quick brown fox.", ["sloth", "grey", "slow"]) == b"The slow grey sloth."


def test_doc_example_leftmost_first():
    # src/ahocorasick.rs:636-650 and :680-692
    o = orc.Oracle([b"append", b"appendage", b"app"], match_kind=LEFTMOST_FIRST)
    assert orc.replace_all_bytes(o, b"append the app to the appendage", ["x", "y", "z"]) == b"x the z to the xage"


def test_utf8_boundary_rule():
    # src/automaton.rs:505-513: a match that splits a code point is skipped and the text passes through
    o = orc.Oracle([b"\xa9", b"a"])
    hay = "aéa".encode()          # 61 C3 A9 61 ; pattern 0 matches the continuation byte A9
    assert orc.replace_all_bytes(o, hay, ["X", "Y"]) == b"Y\xc3XY"
    assert orc.replace_all_bytes(o, hay, ["X", "Y"], utf8_boundaries=True) == "YéY".encode()


def test_no_match_and_empty():
    o = orc.Oracle([b"zzz"])
    assert orc.replace_all_bytes(o, b"abc", ["q"]) == b"abc"
    assert orc.replace_all_bytes(o, b"", ["q"]) == b""
