#!/usr/bin/env python3
"""Time-bounded randomized differential run of the device path against the oracle (not a pytest: run by hand on a
GPU box, `python scripts/fuzz_gpu.py [seconds] [first_seed]`).  Random pattern sets (alphabet, count, lengths,
duplicates, prefixes of each other), automaton configurations, haystack sizes / densities, sub-spans; every call is
compared bit-exactly with the oracle.  Prints the seed of every mismatch and exits non-zero if there was one."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import aho_corasick_amd as ac
from oracle import orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_util import KIND, OKIND, assert_same  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
fails, runs, calls = [], 0, 0
seed = seed0
only = [int(x) for x in os.environ["FUZZ_SEEDS"].split(",")] if os.environ.get("FUZZ_SEEDS") else None   # replay just these seeds
variants = {}
while (only is not None and only) or (only is None and time.time() < t_end):
    if only is not None:
        seed = only.pop(0)
    rng = np.random.default_rng(seed)
    ctx = f"seed {seed} (setup)"
    try:
        asz = int(rng.choice([2, 3, 4, 8, 26, 95, 200]))
        lo = 0x61 if asz <= 26 else (0x20 if asz == 95 else 0x10)
        npat = int(rng.choice([1, 2, 5, 40, 300, 1000, 3000, 9000] + ([26000] if rng.random() < 0.08 else [])))
        maxlen = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 40]))
        # a third of the larger automata have a shortest pattern of 5..11 bytes (the large-set filter's long-prefix level 2; from 8: its 8-byte level 1)
        minlen = int(rng.integers(5, 12)) if (rng.random() < 0.35 and npat >= 300) else 1
        maxlen = max(maxlen, minlen + int(rng.integers(0, 6)))
        pats = []
        for _ in range(npat):
            if pats and rng.random() < 0.15:  # prefix / extension / duplicate of an earlier pattern
                base = pats[int(rng.integers(len(pats)))]
                k = int(rng.integers(min(minlen, len(base)), len(base) + 1))
                p = base[:k] + bytes(rng.integers(lo, lo + asz, size=int(rng.integers(0, 3)), dtype=np.uint8))
            else:
                p = bytes(rng.integers(lo, lo + asz, size=int(rng.integers(minlen, maxlen + 1)), dtype=np.uint8))
            pats.append(p)
        # per-seed knob (read per call by the library): host haystacks searched piece by piece behind the copy
        os.environ["ACGPU_HOST_PIECE_MIB"] = "1" if rng.random() < 0.5 else "256"
        # per-seed engine variants of the automaton (acgpu_set_variant): the large-set filter for every set it can serve;
        # its long-key level 1 without chain tails (a fifth of the seeds), with 14 + 2 wave roles (a third), probing every
        # position (a third) or not at all (a tenth), without the bit-table gate (a fifth); the LDS walk's table flavour and
        # class form at random (forms that do not fit an automaton leave it without that engine, never with another result);
        # leftmost find_iter from the per-start table whatever the density (half of the seeds), in small windows
        # split sets (capi.cpp: a thousand long patterns and a few short ones are searched as two automata, merged on the device)
        if npat >= 1000 and minlen >= 9 and rng.random() < 0.5:
            for _ in range(int(rng.integers(1, 6))):   # (one or two distinct ones of 3..8 bytes: short mode of the large-set filter instead)
                pats.append(bytes(rng.integers(lo, lo + asz, size=int(rng.integers(1, 9)), dtype=np.uint8)))
        variants = {"pfx_min_patterns": 1 if rng.random() < 0.4 else 10000}
        if rng.random() < 0.4:
            variants["pfx_tails"] = int(rng.integers(0, 2))   # (default 2: one record per pattern end of a small subtree)
        if rng.random() < 0.3:
            variants["eo_fused"] = 0                          # (the order pass as separate launches)
        if rng.random() < 0.3:
            variants["pfx_short"] = 0                         # (one or two stragglers beside long patterns: the set as a whole instead of short mode)
        if rng.random() < 0.33:
            variants["pfx_key8_roles"] = 14
        if rng.random() < 0.33:
            variants["pfx_key8_x2"] = 0
        if rng.random() < 0.1:
            variants["pfx_key8"] = 0
        if rng.random() < 0.2:
            variants["pfx_gate"] = 0
        if rng.random() < 0.3:
            variants["lw_cls"] = int(rng.integers(0, 2))
        if rng.random() < 0.15:
            variants["tri_events"] = 0
        if rng.random() < 0.5:
            variants["find_iter_start_table"] = 1
            if rng.random() < 0.6:
                variants["ss_window_kib"] = int(rng.choice([1, 2, 8, 64]))
        # round 6: small automata go to the LDS walk first (or not), its records come from events (or from the chunk fill),
        # find_iter of sets whose occurrences cannot overlap is the overlapping search (or the selection)
        if rng.random() < 0.25:
            variants["lw_first"] = 0
        if rng.random() < 0.25:
            variants["lw_events"] = 0
        if rng.random() < 0.3:
            variants["find_iter_disjoint"] = 0
        deterministic = bool(rng.random() < 0.5)
        mk = int(rng.integers(0, 3))
        kind = [None, "dfa", "cnfa", "nnfa"][int(rng.integers(0, 4))]
        casei = bool(rng.random() < 0.25) and asz in (26, 95)
        bc = bool(rng.random() < 0.8)
        engine = "auto" if rng.random() < 0.7 else ["walk", "hot", "pf"][int(rng.integers(0, 3))]
        sk = int(rng.choice([1, 1, 1, 0]))   # StartKind: mostly Unanchored, sometimes Both (then anchored searches too)
        b = (ac.AhoCorasick.builder().match_kind(mk).start_kind(sk).kind(KIND[kind]).ascii_case_insensitive(casei)
             .byte_classes(bc).gpu_chunk_bytes(int(rng.choice([0, 64, 256, 4096]))).gpu_deterministic_routing(deterministic))
        for vname, vval in variants.items():
            b.gpu_variant(vname, vval)
        try:   # an explicitly requested engine that this automaton cannot have is an error status at search time
            a = b.gpu_engine(engine).build(pats)
            a.find_iter(np.frombuffer(b"probe", dtype=np.uint8), as_numpy=True)
            if mk == 0:
                a.find_overlapping_iter(np.frombuffer(b"probe", dtype=np.uint8), as_numpy=True)
        except Exception:
            engine = "auto"
            a = b.gpu_engine("auto").build(pats)
        if os.environ.get("FUZZ_VERBOSE"):
            print(f"seed {seed}: npat={npat} minlen={minlen} maxlen={maxlen} asz={asz} mk={mk} sk={sk} kind={kind} eng={engine} t={time.time() - (t_end - budget):.1f}", flush=True)
        o = orc.Oracle(pats, match_kind=mk, start_kind=sk, kind=OKIND[kind], ascii_case_insensitive=casei, byte_classes=bc)
        for rep in range(3):
            n = int(rng.choice([0, 1, 17, 1000, 65536, 1 << 20, 3 << 20]))
            # (expected occurrences per byte: thousands of duplicates of short patterns over a 3-letter alphabet would
            # make the ORACLE produce gigabytes of records)
            per_byte = sum(float(asz) ** -len(p) for p in pats)
            while n > 1000 and n * per_byte > 3e6:
                n //= 16
            hay = rng.integers(lo, lo + asz, size=n, dtype=np.uint8)
            if casei and n:
                flip = rng.random(n) < 0.3
                hay = np.where(flip & (hay >= 0x61) & (hay <= 0x7A), hay - 32, hay).astype(np.uint8)
            for _ in range(int(rng.integers(0, 40))):   # planted occurrences
                p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
                if n > len(p):
                    at = int(rng.integers(0, n - len(p)))
                    hay[at:at + len(p)] = p
            if n and rng.random() < 0.3:
                s = int(rng.integers(0, n)); e = int(rng.integers(s, n + 1)); span = (s, e)
            else:
                span = None
            d = torch.from_numpy(hay).cuda() if (n and rng.random() < 0.7) else hay
            inp = ac.Input(d) if span is None else ac.Input(d).range(*span)
            ctx = f"seed {seed} rep {rep} n={n} span={span} npat={npat} mk={mk} sk={sk} kind={kind} casei={casei} eng={engine}"
            if mk == 0:
                want_ov = o.find_overlapping_iter(hay, span=span, as_numpy=True)
                assert_same(a.find_overlapping_iter(inp, as_numpy=True), want_ov, "overlapping " + ctx)
                calls += 1
                if n and len(want_ov) < (1 << 20) and sk == 1:
                    dd = d if torch.is_tensor(d) else torch.from_numpy(hay).cuda()
                    outb = torch.zeros(len(want_ov) * 24 + 240, dtype=torch.uint8, device="cuda")
                    tot = torch.zeros(2, dtype=torch.int64, device="cuda")
                    classic = bool(rng.random() < 0.5)
                    a.overlapping_enqueue(dd, outb, tot, span=span, classic=classic)   # enqueue-only form, either pipeline
                    torch.cuda.synchronize()
                    th = tot.cpu().numpy().view(np.uint64)
                    if int(th[1]) <= a.ENQUEUE_MAX_EVENTS:   # (else: event overflow / abandoned scan -> caller repeats synchronously)
                        assert int(th[0]) == len(want_ov), f"enqueue count {ctx} classic={classic}: {th} vs {len(want_ov)}"
                        assert_same(outb[: len(want_ov) * 24].cpu().numpy().view(ac.MATCH_DTYPE), want_ov, f"enqueue classic={classic} " + ctx)
                    calls += 1
                    # device-to-device synchronous calls, repeated: from the second on a dense result takes the enqueue machinery with
                    # the fused order chain (event_order.hip), whose bucket words are re-zeroed behind each call
                    for again in range(int(rng.integers(1, 4))):
                        outb.fill_(0xEE)
                        m, ok = a.overlapping_device(dd, span=span, out=outb)
                        assert ok and m == len(want_ov), f"device call {again} {ctx}: {m} vs {len(want_ov)}"
                        assert_same(outb[: m * 24].cpu().numpy().view(ac.MATCH_DTYPE), want_ov, f"device call {again} " + ctx)
                        calls += 1
                    if span is None and n > 4 * a.max_pattern_len() + 8 and len(pats) and min(map(len, pats)) > 0:
                        halo = a.max_pattern_len() - 1   # virtual shards on this device through the multi entry point
                        cuts = sorted({0, n} | {int(x) for x in rng.integers(halo + 1, n, size=int(rng.integers(1, 4)))})
                        sh = [dd[(b0 - (halo if i else 0)):e0].clone() for i, (b0, e0) in enumerate(zip(cuts[:-1], cuts[1:]))]
                        if all(c2 - c1 >= 0 for c1, c2 in zip(cuts[:-1], cuts[1:])) and all(b0 >= halo for b0 in cuts[1:-1]):
                            m, _ = a.find_overlapping_multi(sh, outb)
                            assert m == len(want_ov), f"multi count {ctx} cuts={cuts}"
                            assert_same(outb[: m * 24].cpu().numpy().view(ac.MATCH_DTYPE), want_ov, f"multi cuts={cuts} " + ctx)
                            calls += 1
            assert_same(a.find_iter(inp, as_numpy=True), o.find_iter(hay, span=span, as_numpy=True), "find_iter " + ctx)
            w = o.find(hay, span=span)
            g = a.find(inp)
            assert (g is None and w is None) or (g is not None and w is not None and
                                                 (g.pattern(), g.start(), g.end()) == tuple(w)), f"find {ctx}: {g} vs {w}"
            assert a.is_match(inp) == (o.find(hay, span=span, earliest=True) is not None), "is_match " + ctx
            calls += 3
            if sk == 0:   # StartKind::Both: the anchored side uses the automaton's own two-start tables
                ainp = (ac.Input(d) if span is None else ac.Input(d).range(*span)).anchored(ac.Anchored.Yes)
                assert_same(a.find_iter(ainp, as_numpy=True), o.find_iter(hay, span=span, anchored=True, as_numpy=True),
                            "anchored find_iter " + ctx)
                calls += 1
            if rng.random() < 0.2 and n <= (1 << 20) and span is None and all(len(p) for p in pats):
                repl = [bytes([0x41 + (i % 26)]) * (i % 4) for i in range(len(pats))]
                got = a.replace_all_bytes(d, repl)
                got = bytes(got.cpu().numpy()) if hasattr(got, "cpu") else bytes(got)
                assert got == orc.replace_all_bytes(o, hay, repl), "replace_all " + ctx
                calls += 1
        # repeated find_iter calls of ONE automaton over haystacks whose occurrence streams differ by 0.1x .. 10x (the guessed
        # pipeline of capi_find.cpp holds, misses and recovers; the adaptive hints flip), host and device records, with
        # Input::earliest on the way; then the stream search (Standard kinds)
        if all(len(p) for p in pats):
            import io
            per_byte = sum(float(asz) ** -len(p) for p in pats)
            n2 = int(rng.choice([65536, 1 << 20, 4 << 20]))
            while n2 > 4096 and n2 * per_byte > 1e6:
                n2 //= 4
            hays = []
            for dens in (0.0, float(rng.choice([0.001, 0.01, 0.1]))):
                h2 = rng.integers(lo, lo + asz, size=n2, dtype=np.uint8)
                if dens and n2 * dens * per_byte < 3e6:
                    for at in rng.integers(0, max(1, n2 - maxlen - 1), size=int(n2 * dens / max(1, maxlen))):
                        pp = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
                        if at + len(pp) <= n2:
                            h2[at:at + len(pp)] = pp
                hays.append((h2, torch.from_numpy(h2).cuda(), o.find_iter(h2, as_numpy=True)))
            outb = torch.zeros(max(len(w) for _, _, w in hays) * 24 + 240, dtype=torch.uint8, device="cuda")
            for step in range(int(rng.integers(3, 7))):
                h2, d2, want2 = hays[int(rng.integers(0, 2))] if step else hays[0]
                ctx = f"seed {seed} repeated find_iter step {step} n={n2} want={len(want2)} npat={npat} mk={mk} det={deterministic}"
                if rng.random() < 0.5:
                    assert_same(a.find_iter(d2, as_numpy=True), want2, ctx)
                else:
                    m2, ok2 = a.find_iter_device(d2, outb)
                    assert ok2 and m2 == len(want2), ctx
                    assert_same(outb[: m2 * 24].cpu().numpy().view(ac.MATCH_DTYPE), want2, ctx + " device records")
                calls += 1
            h2, d2, _ = hays[1]
            ctx = f"seed {seed} earliest n={n2} npat={npat} mk={mk}"
            assert_same(a.find_iter(ac.Input(d2).earliest(True), as_numpy=True), o.find_iter(h2, earliest=True, as_numpy=True), "find_iter " + ctx)
            w = o.find(h2, earliest=True)
            g = a.find(ac.Input(d2).earliest(True))
            assert (g is None and w is None) or (g is not None and w is not None and (g.pattern(), g.start(), g.end()) == tuple(w)), f"find {ctx}"
            calls += 2
            if mk == 0 and n2 <= (1 << 20) and rng.random() < 0.5:
                ctx = f"seed {seed} stream search n={n2} npat={npat}"
                got = [(m.pattern(), m.start(), m.end()) for m in a.stream_find_iter(io.BytesIO(h2.tobytes()), chunk_bytes=int(rng.choice([4099, 65536, 300001])))]
                want = [(int(p_), int(s_), int(e_)) for p_, s_, e_ in zip(hays[1][2]["pattern"], hays[1][2]["start"], hays[1][2]["end"])]
                assert got == want, ctx
                calls += 1
        runs += 1
    except AssertionError as e:
        fails.append(str(e)[:400])
        print("MISMATCH:", str(e)[:400], "variants", variants, "piece MiB", os.environ.get("ACGPU_HOST_PIECE_MIB"), flush=True)
    except Exception as e:   # an error status where the oracle has a result is a failure too
        fails.append(f"{type(e).__name__}: {e} [{ctx}]"[:400])
        print("ERROR:", fails[-1], "variants", variants, "piece MiB", os.environ.get("ACGPU_HOST_PIECE_MIB"), flush=True)
    seed += 1
print(f"fuzz: {runs} automata, {calls} device calls compared with the oracle, seeds {seed0}..{seed - 1}, {len(fails)} mismatches")
sys.exit(1 if fails else 0)
