#!/usr/bin/env python3
"""Times the count kernel of one engine on the headline workload (8 GiB resident haystack, 1k patterns); the LDS walk
engine's forms are selected with --variant name=value (acgpu_set_variant: lw_flavour, lw_cls, lw_lane_chunk, ...)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--engine", default="hot")
ap.add_argument("--gib", type=float, default=8.0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--patterns", type=int, default=1000)
ap.add_argument("--casei", action="store_true")
ap.add_argument("--alpha", default="ascii", choices=["ascii", "az", "none"])   # none: a haystack of bytes no pattern contains (patterns stay ascii)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--variant", action="append", default=[], help="name=value engine variant of the automaton (repeatable)")
args = ap.parse_args()
lo, span = (0x61, 26) if args.alpha == "az" else (0x20, 95)
pats = ac.gen_patterns(args.patterns, seed=0xAC01, lo=lo, span=span)
if args.alpha == "none":
    lo, span = 0x01, 2
b = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).gpu_engine(args.engine).gpu_chunk_bytes(args.chunk)
if args.casei:
    b.ascii_case_insensitive(True)
for v in args.variant:
    b.gpu_variant(v.split("=")[0], int(v.split("=")[1]))
aut = b.build(pats)
n = int(args.gib * (1 << 30)) // 64 * 64
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02, lo=lo, span=span)
out = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
prof = _lib.CProfile()
for _ in range(3):
    m, ok = aut.overlapping_device(buf, out=out, profile=prof)
torch.cuda.synchronize()
ks, t0 = [], time.perf_counter()
for _ in range(args.steps):
    m, ok = aut.overlapping_device(buf, out=out, profile=prof)
    ks.append(prof.ms_scan)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
k = float(np.mean(ks))
print(json.dumps({"engine": args.engine, "engine_used": int(prof.engine_used), "variants": args.variant,
                  "casei": args.casei, "alpha": args.alpha, "patterns": args.patterns, "matches": int(m),
                  "kernel_ms": round(k, 4), "kernel_min_ms": round(min(ks), 4), "GBps": round(n / k / 1e6, 1), "frac_hbm": round(n / k / 1e6 / 8000, 4),
                  "step_ms": round(dt * 1e3, 4)}))
