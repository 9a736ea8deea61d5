#!/usr/bin/env python3
"""Transcribe the reference's golden vectors into tests/golden/reference_vectors.json.

Runs only in the build container (needs /root/reference, which does not exist
on the GPU box). It parses the `t!(name, &[patterns], "haystack", &[(pid,s,e)..])`
tables of /root/reference/src/tests.rs (collections :96-642) verbatim and adds
the doctest vectors of src/automaton.rs:756-779, src/ahocorasick.rs (find_iter
examples) and README.md:36-77. The JSON is committed; tests read the JSON only.

Usage: python tests/golden/extract_vectors.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")


def strip_comments(src: str) -> str:
    # block comments first (tests.rs:58-64, :555-576 hold retired vectors), then line comments
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = []
    for line in src.split("\n"):
        # no string literal in tests.rs contains "//"
        i = line.find("//")
        out.append(line if i < 0 else line[:i])
    return "\n".join(out)


def parse_rust_string(s: str, i: int):
    """s[i] == '"'. Returns (bytes, next_index). Handles \\xNN, \\n, \\\\, \\", \\0, \\t."""
    assert s[i] == '"'
    i += 1
    out = bytearray()
    while s[i] != '"':
        c = s[i]
        if c == "\\":
            n = s[i + 1]
            if n == "x":
                out.append(int(s[i + 2:i + 4], 16))
                i += 4
            else:
                out.append({"n": 10, "t": 9, "r": 13, "0": 0, "\\": 92, '"': 34, "'": 39}[n])
                i += 2
        else:
            out += c.encode("utf-8")
            i += 1
    return bytes(out), i + 1


def skip_ws(s, i):
    while s[i] in " \t\r\n":
        i += 1
    return i


def parse_t(s: str, i: int):
    """s[i:] starts right after 't!('. Returns (dict, next_index)."""
    i = skip_ws(s, i)
    m = re.match(r"[A-Za-z_][A-Za-z_0-9]*", s[i:])
    name = m.group(0)
    i += len(name)
    i = skip_ws(s, i)
    assert s[i] == ","
    i = skip_ws(s, i + 1)
    assert s[i:i + 2] == "&[", s[i:i + 20]
    i += 2
    pats = []
    while True:
        i = skip_ws(s, i)
        if s[i] == "]":
            i += 1
            break
        if s[i] == ",":
            i += 1
            continue
        b, i = parse_rust_string(s, i)
        pats.append(b)
    i = skip_ws(s, i)
    assert s[i] == ","
    i = skip_ws(s, i + 1)
    hay, i = parse_rust_string(s, i)
    i = skip_ws(s, i)
    assert s[i] == ","
    i = skip_ws(s, i + 1)
    assert s[i:i + 2] == "&["
    i += 2
    matches = []
    while True:
        i = skip_ws(s, i)
        if s[i] == "]":
            i += 1
            break
        if s[i] == ",":
            i += 1
            continue
        m = re.match(r"\(\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*\)", s[i:])
        assert m, s[i:i + 40]
        matches.append([int(m.group(1)), int(m.group(2)), int(m.group(3))])
        i += m.end()
    i = skip_ws(s, i)
    if s[i] == ",":
        i = skip_ws(s, i + 1)
    assert s[i] == ")", s[i:i + 20]
    return {"name": name, "patterns": pats, "haystack": hay, "matches": matches}, i + 1


def enc(b: bytes) -> str:
    return b.hex()


def main():
    src = strip_comments(open(os.path.join(REF, "src/tests.rs"), encoding="utf-8").read())
    groups = {}
    for m in re.finditer(r"const\s+([A-Z_]+)\s*:\s*&'static\s*\[SearchTest\]\s*=\s*&\[", src):
        gname = m.group(1)
        i = m.end()
        tests = []
        while True:
            i = skip_ws(src, i)
            if src[i] == "]":
                break
            if src[i] == ",":
                i += 1
                continue
            assert src[i:i + 3] == "t!(", (gname, src[i:i + 30])
            t, i = parse_t(src, i + 3)
            tests.append(t)
        groups[gname] = tests
    collections = {}
    for m in re.finditer(r"const\s+([A-Z_]+)\s*:\s*TestCollection\s*=\s*&\[(.*?)\];", src, flags=re.S):
        collections[m.group(1)] = [x.strip() for x in m.group(2).replace("\n", " ").split(",") if x.strip()]

    # Doctest / README vectors (hand-cited; tiny)
    doctests = [
        {"name": "automaton_rs_756_overlapping", "cite": "src/automaton.rs:756-779",
         "api": "find_overlapping_iter", "config": {},
         "patterns": [b"append", b"appendage", b"app"], "haystack": b"append the app to the appendage",
         "matches": [[2, 0, 3], [0, 0, 6], [2, 11, 14], [2, 22, 25], [0, 22, 28], [1, 22, 31]]},
        {"name": "readme_36_find_iter", "cite": "README.md:36-50",
         "api": "find_iter", "config": {},
         "patterns": [b"apple", b"maple", b"Snapple"],
         "haystack": b"Nobody likes maple in their apple flavored Snapple.",
         "matches": [[1, 13, 18], [0, 28, 33], [2, 43, 50]]},
        {"name": "readme_63_casei", "cite": "README.md:63-77",
         "api": "find_iter", "config": {"ascii_case_insensitive": True},
         "patterns": [b"apple", b"maple", b"snapple"],
         "haystack": b"Nobody likes maple in their apple flavored Snapple.",
         "matches": [[1, 13, 18], [0, 28, 33], [2, 43, 50]]},
    ]

    def ser(t):
        d = dict(t)
        d["patterns"] = [enc(p) for p in t["patterns"]]
        d["haystack"] = enc(t["haystack"])
        return d

    out = {
        "_comment": "hex-encoded byte strings; transcribed verbatim from the reference by extract_vectors.py",
        "source": "BurntSushi/aho-corasick 1.1.3 src/tests.rs:96-642",
        "groups": {g: [ser(t) for t in ts] for g, ts in groups.items()},
        "collections": collections,
        "doctests": [ser(t) for t in doctests],
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    n = sum(len(v) for v in groups.values())
    print(f"wrote {OUT}: {len(groups)} groups, {n} vectors, {len(collections)} collections")
    for g, ts in sorted(groups.items()):
        print(f"  {g}: {len(ts)}")


if __name__ == "__main__":
    main()
