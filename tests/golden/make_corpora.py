#!/usr/bin/env python3
"""Copies the reference's own benchmark inputs (reusable test DATA, not source code) into
tests/golden/corpora/ as gzip files, with a manifest of their SHA-256 sums.  Run in the build container, where
/root/reference exists; the GPU box only sees the committed fixtures.

    sherlock.txt, opensubtitles/en-huge.txt   haystacks of the reference's `curated` / `opensubtitles` benchmark groups
                                              (benchmarks/definitions/*.toml)
    words-100 / words-5000 / words-15000      its dictionary pattern sets (benchmarks/regexes/)
    dictionary-15                             dictionary/english/length-15.txt, the `dictionary-15` curated benchmark
    en-sampled / ru-sampled / zh-sampled,     the remaining haystacks of benchmarks/definitions/{teddy,curated,random/*}.toml
    random.txt, random10x.txt                 (Cyrillic and CJK text: bytes >= 0x80)
    dictionary-10, dictionary-sorted          dictionary/english/{length-10,sorted}.txt: 123 115 words, 2 004 of them
                                              shorter than 4 bytes
    bench_defs.json                           every [[bench]] of definitions/{sherlock,teddy,curated,same,jetscii}.toml
                                              and definitions/random/{many,misc,memchr}.toml: pattern list (or the
                                              words file it names), haystack (file, or contents x repeat + append), the
                                              reference's expected match counts per engine and its engine list --
                                              golden vectors the reference holds for exactly these inputs

They are the inputs on which a 4-byte prefix filter behaves differently from random ASCII: natural text searched for
dictionary words."""
import gzip
import hashlib
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "corpora")
FILES = {
    "sherlock.txt": "benchmarks/haystacks/sherlock.txt",
    "en-huge.txt": "benchmarks/haystacks/opensubtitles/en-huge.txt",
    "words-100": "benchmarks/regexes/words-100",
    "words-5000": "benchmarks/regexes/words-5000",
    "words-15000": "benchmarks/regexes/words-15000",
    "dictionary-15": "benchmarks/regexes/dictionary/english/length-15.txt",
    "dictionary-10": "benchmarks/regexes/dictionary/english/length-10.txt",
    "dictionary-sorted": "benchmarks/regexes/dictionary/english/sorted.txt",
    "en-sampled.txt": "benchmarks/haystacks/opensubtitles/en-sampled.txt",
    "ru-sampled.txt": "benchmarks/haystacks/opensubtitles/ru-sampled.txt",
    "zh-sampled.txt": "benchmarks/haystacks/opensubtitles/zh-sampled.txt",
    "random.txt": "benchmarks/haystacks/random.txt",
    "random10x.txt": "benchmarks/haystacks/random10x.txt",
}
DEFS = ["sherlock", "teddy", "curated", "same", "jetscii", "random/many", "random/misc", "random/memchr"]

os.makedirs(OUT, exist_ok=True)
manifest = {}
for name, rel in FILES.items():
    data = open(os.path.join(REF, rel), "rb").read()
    with open(os.path.join(OUT, name + ".gz"), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, compresslevel=9) as g:   # mtime=0: reproducible bytes
            g.write(data)
    manifest[name] = {"source": rel, "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()}
json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(manifest, indent=1))


# ---- the benchmark definitions themselves: patterns, haystack, expected counts (the reference's own golden numbers)
import tomli  # noqa: E402



def bstr_unescape(s):
    """bstr::ByteVec::unescape_bytes, which the reference's benchmark harness applies to every pattern
    (benchmarks/shared/lib.rs:69): \\0 \\\\ \\r \\n \\t \\xNN; any other backslash stays."""
    out, b, i = bytearray(), s.encode("utf-8"), 0
    simple = {ord("0"): 0, ord("\\"): 0x5C, ord("r"): 13, ord("n"): 10, ord("t"): 9}
    hexd = b"0123456789abcdefABCDEF"
    while i < len(b):
        if b[i] == 0x5C and i + 1 < len(b):
            c = b[i + 1]
            if c in simple:
                out.append(simple[c]); i += 2; continue
            if c == ord("x") and i + 3 < len(b) and b[i + 2] in hexd and b[i + 3] in hexd:
                out.append(int(b[i + 2:i + 4], 16)); i += 4; continue
        out.append(b[i]); i += 1
    return bytes(out)


by_source = {rel.split("benchmarks/", 1)[1].split("/", 1)[1]: name for name, rel in FILES.items()}   # path under haystacks/ or regexes/ -> fixture
defs = {}
for d in DEFS:
    doc = tomli.load(open(os.path.join(REF, "benchmarks", "definitions", d + ".toml"), "rb"))
    out = []
    for b in doc.get("bench", []):
        rx, hs = b["regex"], b["haystack"]
        e = {"name": b["name"], "model": b["model"], "engines": b["engines"], "count": b["count"]}
        if isinstance(rx, dict):
            assert rx.get("per-line") == "pattern", rx
            e["patterns_file"] = by_source[rx["path"]]
        else:
            e["patterns_hex"] = [bstr_unescape(x).hex() for x in ([rx] if isinstance(rx, str) else list(rx))]
        if "path" in hs:
            if hs["path"] not in by_source:   # (catalog.data.gov/*.xml is not vendored in the reference repository: only its README)
                assert not os.path.exists(os.path.join(REF, "benchmarks", "haystacks", hs["path"])), hs
                continue
            e["haystack_file"] = by_source[hs["path"]]
        else:
            e["haystack"] = {"contents": hs["contents"], "repeat": hs.get("repeat", 1), "append": hs.get("append", "")}
        for k in b:
            assert k in ("name", "model", "engines", "count", "regex", "haystack", "analysis"), k
        out.append(e)
    defs[d] = out
json.dump(defs, open(os.path.join(OUT, "bench_defs.json"), "w"), indent=1, sort_keys=True, ensure_ascii=True)
print({k: len(v) for k, v in defs.items()})
