#!/usr/bin/env python3
"""Clock profile of the large-set filter's wave roles on config 4 (100 000 patterns, random text; 4-byte level 1 with the
bit-table gate): a -DPFX_PROF=1 build of pfx_scan.hip (ACGPU_LIB), one timed call over 2 GiB."""
import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
lib = ctypes.CDLL(os.environ["ACGPU_LIB"])
lib.acgpu_debug_pfx_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
n = 2 << 30
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
ac.gen_haystack(buf, offset=0, seed=0xAC02)
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).build(ac.gen_patterns(100000, seed=0xAC04))
p = _lib.CProfile()
for _ in range(3):
    a.overlapping_device(buf, out=out, profile=p)
torch.cuda.synchronize()
v = (ctypes.c_ulonglong * 16)()
lib.acgpu_debug_pfx_prof(v, 1)
m, ok = a.overlapping_device(buf, out=out, profile=p)
torch.cuda.synchronize()
lib.acgpu_debug_pfx_prof(v, 0)
v = list(v)
cus = torch.cuda.get_device_properties(0).multi_processor_count
np_, nv = cus * 12, cus * 4
us = lambda c, w: round(c / w / 2400.0, 1)
print(json.dumps({"workload": "config 4, 2 GiB", "kernel_ms": round(p.ms_scan, 3), "matches": int(m), "engine": int(p.engine_used),
                  "producer_us": us(v[0], np_), "producer_wait_us": us(v[1], np_), "waits_per_producer": round(v[2] / np_, 1),
                  "verifier_us": us(v[3], nv), "verifier_idle_us": us(v[4], nv), "verifier_levels12_us": us(v[5], nv), "verifier_level3_us": us(v[6], nv),
                  "rounds_per_verifier": round(v[7] / nv, 1), "survivors_per_round": round(v[8] / max(1, v[7]), 1),
                  "level3_batches_per_verifier": round(v[9] / nv, 1), "hits_per_batch": round(v[10] / max(1, v[9]), 1),
                  "us_per_round": round(v[5] / max(1, v[7]) / 2400.0, 2), "us_per_level3_batch": round(v[6] / max(1, v[9]) / 2400.0, 2)}), flush=True)
