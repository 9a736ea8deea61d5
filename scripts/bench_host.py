#!/usr/bin/env python3
"""PCIe-inclusive rate: the caller hands a HOST haystack (pageable numpy / pinned torch) to find_overlapping_iter."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aho_corasick_amd as ac

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30))
pats = ac.gen_patterns(1000, seed=0xAC01)
a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build(pats)
def bench_all():
    for name, h in (("pageable numpy", host), ("pinned torch tensor", pinned), ("device tensor", dev)):
        a.find_overlapping_iter(h, as_numpy=True)
        t0 = time.perf_counter()
        k = 3
        for _ in range(k):
            m = a.find_overlapping_iter(h, as_numpy=True)
        dt = (time.perf_counter() - t0) / k
        print(f"{name:20s}: {gib} GiB in {dt*1e3:8.2f} ms = {n/dt/1e9:7.1f} GB/s ({len(m)} matches)")


dev = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(dev, offset=0, seed=0xAC02)
host = dev.cpu().numpy()
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
pinned.copy_(dev)
stage = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, h in (("pageable numpy", torch.from_numpy(host)), ("pinned torch tensor", pinned)):   # the copy alone, for reference
    stage.copy_(h); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        stage.copy_(h)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"copy only, {name:20s}: {gib} GiB in {dt*1e3:8.2f} ms = {n/dt/1e9:7.1f} GB/s")
for piece in ("256", "100000"):   # 100000 MiB pieces = one piece = copy, then scan (the round-1 behaviour)
    os.environ["ACGPU_HOST_PIECE_MIB"] = piece
    print(f"-- ACGPU_HOST_PIECE_MIB={piece}")
    bench_all()
