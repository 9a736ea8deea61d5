#!/usr/bin/env python3
"""Where the wavefronts of the large-set filter spend their clocks (a -DPFX_PROF=1 build of pfx_scan.hip in lib/exp,
ACGPU_LIB pointing at it): natural text, 1 GiB, one timed call; per-wave averages in microseconds at 2.4 GHz."""
import os, sys, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import aho_corasick_amd as ac
from aho_corasick_amd import _lib
import corpora
lib = ctypes.CDLL(os.environ["ACGPU_LIB"])
lib.acgpu_debug_pfx_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
n = 1024 << 20
out = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for hay_name, words_name in (("sherlock.txt", "words-5000"), ("en-huge.txt", "words-15000")):
    text = corpora.haystack(hay_name)
    nat = torch.from_numpy(np.tile(text, -(-n // len(text)))[:n].copy()).cuda()
    p = _lib.CProfile()
    for roles in os.environ.get("KEY8_VARIANTS", "12").split(","):
        a = (ac.AhoCorasick.builder().match_kind(ac.MatchKind.Standard).gpu_engine("pf").gpu_variant("pfx_min_patterns", 1)
             .gpu_variant("pfx_key8_roles", int(roles)).build(corpora.words(words_name)))
        for _ in range(3):
            a.overlapping_device(nat, out=out, profile=p)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        lib.acgpu_debug_pfx_prof(buf, 1)
        m, ok = a.overlapping_device(nat, out=out, profile=p)
        torch.cuda.synchronize()
        lib.acgpu_debug_pfx_prof(buf, 0)
        v = list(buf)
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        P, V = int(roles), 16 - int(roles)
        np_, nv = cus * P, cus * V
        us = lambda c, w: round(c / w / 2400.0, 1)
        print(json.dumps({"haystack": hay_name, "roles": roles, "kernel_ms": round(p.ms_scan, 3), "matches": int(m),
                          "producer_us": us(v[0], np_), "producer_wait_us": us(v[1], np_), "waits_per_producer": round(v[2] / np_, 1),
                          "verifier_us": us(v[3], nv), "verifier_idle_us": us(v[4], nv), "verifier_levels12_us": us(v[5], nv),
                          "verifier_level3_us": us(v[6], nv), "rounds_per_verifier": round(v[7] / nv, 1), "survivors_per_round": round(v[8] / max(1, v[7]), 1),
                          "level3_batches_per_verifier": round(v[9] / nv, 1), "hits_per_batch": round(v[10] / max(1, v[9]), 1),
                          "flush_us": us(v[11], nv), "slow_us": us(v[13], nv), "slow_batches_per_verifier": round(v[14] / nv, 1), "hits_per_slow_batch": round(v[15] / max(1, v[14]), 1),
                          "us_per_slow_batch": round(v[13] / max(1, v[14]) / 2400.0, 2), "walk_trips_per_batch": round(v[12] / max(1, v[9]), 2),
                          "us_per_round": round(v[5] / max(1, v[7]) / 2400.0, 2), "us_per_level3_batch": round(v[6] / max(1, v[9]) / 2400.0, 2)}), flush=True)
