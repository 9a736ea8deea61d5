#!/usr/bin/env python3
"""Throughput of every count engine (and the automatic choice) over the input kinds DESIGN.md section 4 tabulates:
random ASCII, a-z, natural English text (the reference's sherlock.txt tiled) with random / dictionary patterns, and
adversarial haystacks made of pattern prefixes.  One JSON line per (input, engine); whole synchronous call incl. the
ordered records, and the count kernel alone."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=1.0)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--engines", default="auto,pf,hot,walk")
ap.add_argument("--only", default="")
args = ap.parse_args()
N = int(args.gib * (1 << 30)) // 4096 * 4096


def tiled(a):
    reps = -(-N // len(a))
    return torch.from_numpy(np.tile(a, reps)[:N].copy()).cuda()


def rand_dev(lo, span, seed=0xAC02):
    t = torch.empty(N, dtype=torch.uint8, device="cuda")
    ac.gen_haystack(t, offset=0, seed=seed, lo=lo, span=span)
    return t


def adversarial(pats, k):
    rng = np.random.default_rng(3)
    pieces = [pats[int(i)][:k] for i in rng.integers(0, len(pats), size=(16 << 20) // max(1, min(k, 8)))]
    return tiled(np.frombuffer(b"".join(pieces), dtype=np.uint8)[: 16 << 20])


P1K = ac.gen_patterns(1000, seed=0xAC01)
PAZ = ac.gen_patterns(1000, seed=0xAC01, lo=0x61, span=26)
CASES = [
    ("random ASCII / 1k random patterns", lambda: rand_dev(0x20, 95), P1K, "dfa"),
    ("a-z / 1k a-z patterns", lambda: rand_dev(0x61, 26), PAZ, "dfa"),
    ("English (sherlock) / 1k random patterns", lambda: tiled(corpora.haystack("sherlock.txt")), P1K, "dfa"),
    ("English (sherlock) / words-100", lambda: tiled(corpora.haystack("sherlock.txt")), corpora.words("words-100"), None),
    ("English (sherlock) / words-5000", lambda: tiled(corpora.haystack("sherlock.txt")), corpora.words("words-5000"), None),
    ("English (sherlock) / dictionary-15", lambda: tiled(corpora.haystack("sherlock.txt")), corpora.words("dictionary-15"), None),
    ("English (en-huge subtitles) / words-15000", lambda: tiled(corpora.haystack("en-huge.txt")), corpora.words("words-15000"), None),
    ("adversarial: 4-byte pattern prefixes / 1k random patterns", lambda: adversarial(P1K, 4), P1K, "dfa"),
    ("adversarial: 8-byte pattern prefixes / 1k random patterns", lambda: adversarial(P1K, 8), P1K, "dfa"),
    ("adversarial: whole patterns back to back / 1k random patterns", lambda: adversarial(P1K, 99), P1K, "dfa"),
]
out = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for name, mk, pats, kind in CASES:
    if args.only and args.only not in name:
        continue
    hay = mk()
    for eng in args.engines.split(","):
        b = ac.AhoCorasick.builder().gpu_engine(eng)
        if kind == "dfa":
            b.kind(ac.AhoCorasickKind.DFA)
        aut = b.build(pats)
        prof = _lib.CProfile()
        try:
            for _ in range(2):
                m, ok = aut.overlapping_device(hay, out=out, profile=prof)
            torch.cuda.synchronize()
            ks, t0 = [], time.perf_counter()
            for _ in range(args.steps):
                m, ok = aut.overlapping_device(hay, out=out, profile=prof)
                ks.append(prof.ms_scan)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            k = float(np.mean(ks))
            print(json.dumps({"input": name, "engine": eng, "engine_used": int(prof.engine_used), "routed": int(prof.routed), "matches": int(m), "fits": bool(ok),
                              "call_ms": round(dt * 1e3, 3), "call_GBps": round(N / dt / 1e9, 1), "kernel_ms": round(k, 3),
                              "kernel_GBps": round(N / k / 1e6, 1), "frac_hbm": round(N / k / 1e6 / 8000, 4)}), flush=True)
        except RuntimeError as e:
            print(json.dumps({"input": name, "engine": eng, "error": str(e)[:120]}), flush=True)
        del aut
    del hay
