#!/usr/bin/env python3
"""Per-call launch timelines of selected benchmark definitions (run under rocprofv3 --kernel-trace; call_timeline.py reads
the trace).  Each measured call is preceded by a marker launch (a torch fill: its kernel name does not start with acgpu)
and followed by a synchronise + 1 ms sleep, so the trace splits into calls; stdout carries one JSON line per call with the
wall time the host saw, in trace order.
usage: trace_defs.py <mib> <name,name,...> [variant=value,...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora
n = int(sys.argv[1]) << 20
only = [x for x in sys.argv[2].split(",") if x]
variants = dict((v.split("=")[0], int(v.split("=")[1])) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else {}
out = torch.empty(int(6.1 * (1 << 30)), dtype=torch.uint8, device="cuda")
marker = torch.zeros(64, dtype=torch.int32, device="cuda")
defs = corpora.bench_defs()
defs["dictionary"] = [{"name": "sorted.txt", "patterns_file": "dictionary-sorted", "haystack_file": "sherlock.txt"}]


def build(b):
    for k_, v_ in variants.items():
        b.gpu_variant(k_, v_)
    return b


def timed(kind, name, call, reps=3):
    for _ in range(4):   # warm: adaptive hints settle
        call()
    torch.cuda.synchronize()
    for r in range(reps):
        time.sleep(0.001)
        marker.fill_(r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m, ok = call()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(json.dumps({"bench": name, "kind": kind, "rep": r, "records": int(m), "ok": bool(ok), "wall_us": round(dt * 1e6, 1),
                          "GBps": round(n / dt / 1e9, 1)}), flush=True)


for family, benches in defs.items():
    for b in benches:
        full = family + ":" + b["name"]   # names repeat across families (random/memchr:onebyte-match, same:onebyte-match)
        if not any(o in (b["name"], full) or (o.endswith("*") and (b["name"].startswith(o[:-1]) or full.startswith(o[:-1]))) for o in only):
            continue
        pats, hay = corpora.bench_patterns(b), corpora.bench_haystack(b)
        d = torch.from_numpy(np.tile(hay, -(-n // len(hay)))[:n].copy()).cuda()
        a = build(ac.AhoCorasick.builder()).build(pats)
        p = _lib.CProfile()
        timed("ov", full, lambda: a.overlapping_device(d, out=out, profile=p))
        lf = build(ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst)).build(pats)
        timed("lf", full, lambda: lf.find_iter_device(d, out))
        del d
