#!/usr/bin/env python3
"""A/B of the large-set filter's eight-byte level 1 (engine variants pfx_key8 / pfx_key8_roles / pfx_key8_x2 / pfx_tails: one automaton per form) on natural text: same automaton, same
1 GiB haystack, results compared record for record (CRC of the ordered records), kernel and call times of both."""
import os, sys, time, json, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
pairs = (("sherlock.txt", "words-5000"), ("en-huge.txt", "words-15000")) if len(sys.argv) <= 2 else (("sherlock.txt", "words-5000"),)
VARIANTS = tuple(os.environ.get("KEY8_VARIANTS", "0,8,12").split(","))
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for hay_name, words_name in pairs:
    text = corpora.haystack(hay_name)
    nat = torch.from_numpy(np.tile(text, -(-n // len(text)))[:n].copy()).cuda()
    p = _lib.CProfile()
    res = {}
    for key8 in VARIANTS:   # "0": the 4-byte level 1; else the long-key one with that many producer wavefronts
        name = key8
        plain = key8.endswith("p")      # "12p": the long-key level 1 probing EVERY position (pfx_key8_x2 = 0)
        key8 = key8.rstrip("p")
        no_tails = key8.endswith("n")   # "12n": without the chain-tail records behind the prefix map (level 3 walks the trie)
        b = ac.AhoCorasick.builder().match_kind(ac.MatchKind.Standard).gpu_engine("pf").gpu_variant("pfx_min_patterns", 1)
        b.gpu_variant("pfx_key8_x2", 0 if plain else 1).gpu_variant("pfx_tails", 0 if no_tails else 1)
        b.gpu_variant("pfx_key8", 0 if key8.rstrip("n") == "0" else 1)
        if key8.rstrip("n") != "0":
            b.gpu_variant("pfx_key8_roles", int(key8.rstrip("n")))
        a = b.build(corpora.words(words_name))
        for _ in range(2):
            m, ok = a.overlapping_device(nat, out=out, profile=p)
        torch.cuda.synchronize()
        ks, t0 = [], time.perf_counter()
        for _ in range(5):
            m, ok = a.overlapping_device(nat, out=out, profile=p); ks.append(p.ms_scan)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        crc = zlib.crc32(out[: int(m) * 24].cpu().numpy().tobytes())
        res[name] = {"matches": int(m), "crc": crc, "call_ms": round(dt * 1e3, 3), "kernel_ms": round(float(np.mean(ks)), 3), "engine": int(p.engine_used)}
    print(json.dumps({"haystack": hay_name, "words": words_name, "mib": n >> 20, **{("key4" + k[1:] if k[0] == "0" else "key8_p" + k): res[k] for k in res},
                      "identical": len({(r["crc"], r["matches"]) for r in res.values()}) == 1}), flush=True)
