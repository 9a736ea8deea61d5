#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

Metric: GB/s of haystack scanned (and % of HBM peak), 1 000-pattern full DFA, MatchKind::Standard
overlapping search, 8 GiB of synthetic random-ASCII haystack per GPU, bit-exact ordered matches.

A "step" is one complete find_overlapping pass over the GPU-resident haystack, ending with the ordered match
records in device memory: by default the prefix-filter engine over the DFA's pattern set (k_pf_count: two LDS
Bloom tables + exact trie walk of the survivors -> k_ev_rank -> k_ev_write), or with
--engine hot|walk the byte-at-a-time DFA transition walk (count kernel -> count scan + compaction -> ordered
match fill); then (N>1) the gather of the match records to rank 0 over RCCL.  At N>1 the haystack is
N x 8 GiB, partitioned contiguously across the ranks (weak scaling, BASELINE config 3 at N=8); each rank warms
up on max_pattern_len-1 bytes left of its seam.

At N=1 the JSON line additionally carries, measured on the same resident haystack after the timed region:
  "engines": the three count engines of the headline workload side by side (kernel ms, GB/s, fraction of HBM peak),
  "also":    the reference's own corpora (English prose / 5 000 dictionary words) and
             BASELINE configs 4 (100k patterns, contiguous NFA; default engine and the literal failure-link walk)
             and 5 (casei LeftmostFirst find_iter), each with its own roofline / kernel_ms.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gib G] [--engine auto|walk|hot] [--chunk B]

N>1 is launched by the driver through torch.distributed.run (one rank per GPU).  Rank 0 prints ONE JSON
line.  `roofline` is the dominant kernel (count/scan walk) measured with HIP events on the launch
stream inside the library; `cpu_baseline` is the oracle's restatement of the reference DFA loop
(src/dfa.rs:218-226 + src/automaton.rs:1491-1534) on one host core over a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def committed_traffic(tag, kernel, gib):
    """HBM/fabric bytes per launch of `kernel` from the newest committed counter file profiles/r*_<tag>_pmc.json (a separate
    rocprofv3 --pmc pass: scripts/pmc_traffic.sh; PMC passes cannot run inside a timed bench).  The file must name the same
    kernel (template arguments included) and the same launch size, else the line carries no traffic figure."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
            if "hbm_read_bytes" not in d or "_kernel" not in d or "_gib" not in d:   # (files of rounds 1-4 name neither)
                continue
            if kernel not in (d["_kernel"] or "") or abs(float(d["_gib"]) - gib) > 1e-9:
                continue
            return float(d["hbm_read_bytes"]), os.path.relpath(path, ROOT) + " (TCC_EA0_RDREQ_{32B,64B,128B}, separate --pmc pass)", d.get("_kernel")
        except Exception:
            continue
    return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gib", type=float, default=8.0, help="haystack GiB per GPU (BASELINE: 8)")
    ap.add_argument("--engine", default="auto")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--patterns", type=int, default=1000)
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5"],
                    help="c2 = headline (BASELINE configs[1]); c4 = 100k patterns, contiguous-NFA failure-link walk; "
                         "c5 = 1k patterns, ascii_case_insensitive + LeftmostFirst find_iter (parity-test configs, "
                         "timed here for the record only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the engines / configs 4+5 side measurements (N=1)")
    ap.add_argument("--also-steps", type=int, default=10)
    ap.add_argument("--cpu-sample-mib", type=int, default=1024)
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc pass; default: the committed "
                         "profiles/*_pf_pmc.json of the same kernel/config (PMC passes cannot run inside a timed bench)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import aho_corasick_amd as ac
    from aho_corasick_amd import _lib
    from aho_corasick_amd.distributed import MatchGatherer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test knobs for validating the N>1 code path on a 1-GPU box (gloo, every rank on cuda:0); never set by the driver
    backend = os.environ.get("ACGPU_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("ACGPU_BENCH_ONE_DEVICE") == "1" else local_rank
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")   # where collective tensors live

    # ---- automaton: 1 000 random 4-16 byte patterns over printable ASCII
    if args.workload == "c4":
        args.patterns = 100000 if args.patterns == 1000 else args.patterns
        pats = ac.gen_patterns(args.patterns, seed=0xAC04)
        aut = (ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).match_kind(ac.MatchKind.Standard)
               .gpu_chunk_bytes(args.chunk).build(pats))
    elif args.workload == "c5":
        pats = ac.gen_patterns(args.patterns, seed=0xAC01)
        aut = (ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(ac.MatchKind.LeftmostFirst)
               .ascii_case_insensitive(True).gpu_engine(args.engine).gpu_chunk_bytes(args.chunk).build(pats))
    else:
        pats = ac.gen_patterns(args.patterns, seed=0xAC01)
        aut = (ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(ac.MatchKind.Standard)
               .gpu_engine(args.engine).gpu_chunk_bytes(args.chunk).build(pats))
    aut.upload(dev_index)
    L = aut.max_pattern_len()
    halo = L - 1

    # ---- haystack: this rank's shard of the global synthetic haystack, generated on the device
    shard = int(args.gib * (1 << 30)) // 64 * 64
    g_begin = rank * shard            # global offset of the shard
    left = halo if rank > 0 else 0    # warm-up bytes that belong to the left neighbour
    buf = torch.empty(left + shard, dtype=torch.uint8, device=dev)
    ac.gen_haystack(buf, offset=g_begin - left, seed=0xAC02)
    # planted occurrences (identical global positions on every rank): straddling every shard seam and
    # a spread of lane-chunk seams, so seams are exercised although random matches are sparse
    total = shard * world
    planted = []
    for k in range(1, world):
        for j, d in enumerate((1, 3, 8, 15)):
            planted.append((k * shard - d, pats[(k * 4 + j) % len(pats)]))
    for j in range(64):
        planted.append((((j + 1) * total // 65) // 4096 * 4096 - (j % 16), pats[(7 * j) % len(pats)]))
    for pos, p in planted:
        lo, hi = pos - (g_begin - left), pos - (g_begin - left) + len(p)
        if lo >= 0 and hi <= buf.numel():
            buf[lo:hi] = torch.frombuffer(bytearray(p), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()

    out = torch.empty(64 << 20, dtype=torch.uint8, device=dev)  # room for 2.8M records
    # N>1: one all_gather of fixed-size payloads per step; rank 0 adds each shard's coordinate offset on the host
    gatherer = MatchGatherer(cap=32768, dst=0, device=coll_dev) if world > 1 else None
    shard_offsets = [r * shard - (halo if r > 0 else 0) for r in range(world)]
    span = (left, left + shard)   # local coordinates; the shard owns ends in (left, left+shard]
    prof = _lib.CProfile()

    def step():
        if args.workload == "c5":  # non-overlapping find_iter (single GPU): occurrence stream + device selection
            arr = aut.find_iter(buf, as_numpy=True, profile=prof)
            return torch.from_numpy(arr.view(np.uint8).copy()), len(arr)
        # cold floor: the global search starts at global offset 0, i.e. local offset 0 on rank 0 and
        # "left of the halo" elsewhere -- the halo is exactly what the seam rule needs
        n, ok = aut.overlapping_device(buf, span=(0, left + shard), shard=span, out=out, profile=prof)
        assert ok, "match buffer too small"
        rec = out[: n * 24]
        if world > 1:   # the records of every shard land in rank 0's memory (one all_gather, no host round trip);
            return gatherer.gather_device(out, n), n   # decoded after the timed loop, like the N=1 device records
        return rec, n

    will_enqueue = args.workload == "c2" and os.environ.get("ACGPU_BENCH_SYNC") != "1"
    n_warm = 0
    n_warm_steps = 1 if will_enqueue else max(args.warmup, 1)   # (the pipelined form adds its own warm-up steps below)
    # synchronous warm-up calls (at least one: it reports the engine that runs and the record count; the pipelined
    # form below does its own W warm-up steps right in front of the timed region)
    for _ in range(1 if will_enqueue else max(args.warmup, 1)):
        _, n_warm = step()
    if world > 1:   # size the gather payload from what the warm-up saw (identical capacity on every rank)
        t = torch.tensor([n_warm], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        want_cap = max(4096, 2 * int(t.item())) if args.warmup else gatherer.cap
        if want_cap > gatherer.cap or want_cap * 4 < gatherer.cap:
            gatherer._alloc(want_cap)
    # Pipelined form of a step (default for the headline workload): the search is only ENQUEUED
    # (acgpu_find_overlapping_enqueue: count kernel -> event rank -> ordered records, all on the device) and, for
    # N>1, so is the gather of the records -- no host round trip inside a step, the host runs ahead of the GPU and the
    # K steps execute back to back.  Everything a synchronous step does is still done inside the timed region, which
    # ends with a full synchronisation.  ACGPU_BENCH_SYNC=1 times the synchronous calls instead.
    totals = torch.zeros(2, dtype=torch.int64, device=dev)

    def step_enqueue(i):
        aut.overlapping_enqueue(buf, out, totals, span=(0, left + shard), shard=span, slot=i % 64)
        if world > 1:
            gatherer.gather_device_async(out, totals)

    use_enqueue = will_enqueue
    if use_enqueue:   # one untimed pipelined step decides (identically on every rank) whether the form applies
        ok = 1
        try:
            step_enqueue(0)
            torch.cuda.synchronize()
            t_host = totals.cpu().numpy()
            ok = int(t_host[1] <= aut.ENQUEUE_MAX_EVENTS and t_host[0] == n_warm and
                     (world == 1 or t_host[0] <= gatherer.cap))
        except Exception as e:   # an engine other than the prefix filter was requested
            print(f"bench: enqueue-only form unavailable ({e}); timing synchronous calls", file=sys.stderr)
            ok = 0
        if world > 1:
            t = torch.tensor([ok], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        use_enqueue = bool(ok)
        # the W warm-up steps, back to back with the timed ones: the first call allocates its per-stream context
        # (tens of ms of idle GPU), after which the clocks need ~10 launches to come back up
        # exactly the W the caller asked for (the driver checks it); the default W (20) is what the clock ramp wants
        n_warm_steps = args.warmup
        for i in range(n_warm_steps):
            step_enqueue(i) if use_enqueue else step()

    scan_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if use_enqueue:
        for i in range(args.steps):
            step_enqueue(i)
    else:
        for _ in range(args.steps):
            res, n_local = step()
            scan_ms.append(prof.ms_scan)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_enqueue:
        t_host = totals.cpu().numpy()
        n_local = int(t_host[0])
        assert t_host[1] <= aut.ENQUEUE_MAX_EVENTS and n_local == n_warm, "pipelined step lost its result"
        res = out[: n_local * 24]
        scan_ms = [aut.enqueue_kernel_ms(i % 64) for i in range(max(0, args.steps - 64), args.steps)]
    per_rank = None
    if world > 1:
        res = gatherer.finalize(shard_offsets)   # rank 0: host copy + per-shard offsets of the last step's records
        # what every rank spent: its own wall time, its scan kernel (HIP events on the launch stream), and the rest of a
        # step (gather + launch gaps) -- so that a scaling run can be diagnosed from its one JSON line
        mine = torch.tensor([dt / args.steps * 1e3, float(np.mean(scan_ms)) if len(scan_ms) else 0.0], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "ms_per_step": round(float(x[0]), 4), "kernel_ms": round(float(x[1]), 4),
                     "gather_and_gaps_ms": round(float(x[0] - x[1]), 4)} for r, x in enumerate(allr)]
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    verify = None
    if world > 1 and os.environ.get("ACGPU_BENCH_VERIFY") == "1":
        # parity of the sharded run against the CPU ORACLE (test infrastructure, after the timed region): the whole global
        # haystack regenerated on the host by the oracle's own generator, the same occurrences planted, searched by the
        # oracle's chunk-parallel restatement of the reference loop; the gathered records must equal its stream
        from oracle import orc
        full = orc.gen_haystack(0, total, seed=0xAC02)
        for pos, p in planted:
            if pos >= 0 and pos + len(p) <= total:
                full[pos:pos + len(p)] = np.frombuffer(p, dtype=np.uint8)
        ref = orc.Oracle(pats, kind=orc.KIND_DFA).find_overlapping_iter(full, as_numpy=True)
        verify = bool(len(ref) == len(res) and all(np.array_equal(ref[f], res[f]) for f in ("pattern", "start", "end")))
        del full

    traffic, traffic_src = args.traffic_bytes, "--traffic-bytes"
    if traffic is None and int(prof.engine_used) == 4 and args.patterns == 1000 and args.workload == "c2":
        traffic, traffic_src, _ = committed_traffic("pf", "k_pf_count<false, false>", args.gib)
    # the same-box streaming ceiling: a plain read-only kernel over this very buffer
    try:
        empirical = float(ac.stream_read_gbps(buf[: buf.numel() // 16 * 16], iters=5))
    except Exception as exc:
        print(f"bench: streaming ceiling unavailable ({exc})", file=sys.stderr)
        empirical = None
    ms_per_step = dt / args.steps * 1e3
    value = total * args.steps / dt / 1e9
    kernel_ms = float(np.mean(scan_ms))
    achieved = shard / (kernel_ms * 1e-3) / 1e9   # algorithmic bytes: 1 B per haystack byte scanned
    if world > 1:
        n_matches = len(res)
    else:
        n_matches = n_local
    result = {
        **({"per_rank": per_rank} if per_rank else {}),
        "metric": {"c2": "GB/s haystack scanned, 1k-pattern full-DFA overlapping, 8 GiB/GPU (default engine: prefix filter "
                         "over the DFA's pattern set; the DFA transition-walk engines are reported under `engines`)",
                   "c4": "GB/s haystack scanned, 100k-pattern overlapping (parity config 4: reference kind contiguous NFA)",
                   "c5": "GB/s haystack scanned, 1k-pattern casei LeftmostFirst find_iter (parity config 5)"}[args.workload],
        "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm_steps,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": {"c2": "configs[1]: 1000 random 4-16 B patterns, random-ASCII haystack, full DFA, "
                                      "MatchKind::Standard overlapping, bit-exact ordered matches",
                                "c4": "configs[3]: 100000 patterns, AhoCorasickKind::ContiguousNFA; the device runs the prefix filter over the full DFA derived from the same noncontiguous NFA (411 MB in HBM, rows filled on the device)",
                                "c5": "configs[4]: 1000 patterns, ascii_case_insensitive + LeftmostFirst, find_iter"}[args.workload],
                   "haystack_gib_per_gpu": args.gib, "patterns": args.patterns, "engine": int(prof.engine_used),
                   "call": "enqueue-only (pipelined, no host round trip per step)" if use_enqueue else "synchronous",
                   "chunk_bytes": int(shard // max(int(prof.n_chunks), 1)) if prof.n_chunks else 0,
                   "matches": int(n_matches), "pct_hbm_peak": round(100.0 * value / (HBM_PEAK_GBS * world), 3),
                   **({"sharded_equals_oracle": verify} if verify is not None else {}),
                   **({"ranks": world, "collective_backend": ("rccl (torch.distributed nccl backend)" if backend == "nccl" else backend),
                       # one all_gather_into_tensor per step: every rank contributes 8 + 24 * cap bytes and receives world times that
                       "gather_payload_bytes_per_rank_per_step": 8 * (1 + 3 * gatherer.cap),
                       "gather_bytes_received_per_rank_per_step": 8 * (1 + 3 * gatherer.cap) * world,
                       "gather_capacity_records": gatherer.cap} if world > 1 else {})},
        "roofline": {"bound": "hbm",
                     "kernel": {4: "k_pf_count (prefix filter: two LDS Bloom tables + exact trie walk, one launch per shard)",
                                3: "k_lw_count (DFA transition walk, whole automaton in LDS)",
                                2: "k_tri_walk<CnfaTriDev> (contiguous-NFA failure-link walk, depth <= 2 skipped by an LDS trigram bitmap)",
                                1: "k_tri_walk<DfaTriDev> (DFA transition walk, depth <= 2 skipped by an LDS trigram bitmap)"}.get(int(prof.engine_used), "?"),
                     "achieved": round(achieved, 3),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "empirical_peak": round(empirical, 1) if empirical else None,
                     "frac_of_empirical": round(achieved / empirical, 5) if empirical else None,
                     "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
                     "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes_per_launch": shard},
    }

    # ---- CPU baseline: the oracle's DFA overlapping loop on ONE host core over a bounded sample
    if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
        from oracle import orc   # the checker, timed as the CPU baseline: the ONLY use of oracle/ in this file
        sample = min(args.cpu_sample_mib << 20, shard)
        host = buf[:sample].cpu().numpy()
        o = orc.Oracle(pats, kind=orc.KIND_DFA)
        o.dfa_overlapping_count(host[: 1 << 24])  # warm-up
        times = []
        for _ in range(5):
            t1 = time.perf_counter()
            cnt, hsh = o.dfa_overlapping_count(host)
            times.append(time.perf_counter() - t1)
        med = sorted(times)[2]
        # parity of the sample against the GPU result of the last timed step (same bytes)
        rec = res.cpu().numpy().view(ac.MATCH_DTYPE)
        gsel = rec[rec["end"] <= sample]
        gh = 0xCBF29CE484222325
        for p, s_, e_ in zip(gsel["pattern"].tolist(), gsel["start"].tolist(), gsel["end"].tolist()):
            for w in (p, s_, e_):
                for i in range(8):
                    gh = ((gh ^ ((w >> (8 * i)) & 0xFF)) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        result["cpu_baseline"] = {
            "value": round(sample / med / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"first {sample >> 20} MiB of the same haystack, median of 5 passes, oracle DFA loop "
                      f"(C, -O3 -march=native); host nproc={os.cpu_count()}",
            "matches_in_sample": int(cnt), "gpu_matches_in_sample": int(len(gsel)),
            "sample_parity": bool(int(cnt) == len(gsel) and gh == hsh),
        }
        # for context only: the same loop chunk-parallel over every host core (pthreads in the oracle, seam rule =
        # warm-up of max_pattern_len-1 bytes, a chunk owns the matches ending inside it) -- the reference itself is
        # single-threaded
        try:
            ncpu = os.cpu_count() or 1
            o.dfa_overlapping_count_parallel(host, ncpu)   # warm-up
            t1 = time.perf_counter()
            total_par = o.dfa_overlapping_count_parallel(host, ncpu)
            t_par = time.perf_counter() - t1
            v_par = sample / t_par / 1e9
            result["cpu_baseline"]["all_cores"] = {"value": round(v_par, 3), "unit": "GB/s", "threads": ncpu,
                                                   "speedup_vs_one_core": round(v_par / (sample / med / 1e9), 1),
                                                   "matches_in_sample": int(total_par),
                                                   "parity": bool(int(total_par) == int(cnt)),
                                                   "note": "os.cpu_count() threads; the speed-up is what the box's CPU quota allows"}
        except Exception as exc:  # context figure only
            result["cpu_baseline"]["all_cores"] = {"error": str(exc)}
    else:
        result["cpu_baseline"] = None

    # ---- what a search costs before its first byte (outside the timed region; the reference states its own build times,
    # src/ahocorasick.rs:44-50): CPU construction, upload (tables derived on the device included), device memory of the tables
    if world == 1 and not args.no_also and args.workload == "c2":
        def cost_of(label, make, first_call=None):
            try:
                t1 = time.perf_counter()
                a_ = make()
                t2 = time.perf_counter()
                torch.cuda.synchronize()
                free0, _ = torch.cuda.mem_get_info(dev)
                a_.upload(dev_index)
                if first_call:   # (tables that are uploaded by the first search: the Standard twin of a leftmost automaton)
                    first_call(a_)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                free1, _ = torch.cuda.mem_get_info(dev)
                return {"automaton": label, "build_ms": round((t2 - t1) * 1e3, 1), "upload_ms": round((t3 - t2) * 1e3, 1),
                        "device_table_bytes": int(free0 - free1), "host_memory_usage_bytes": int(a_.memory_usage())}
            except Exception as exc:
                return {"automaton": label, "error": str(exc)}
        B = ac.AhoCorasick.builder
        pats4c = ac.gen_patterns(100000, seed=0xAC04)
        result["costs"] = [
            cost_of("c2: 1000 patterns, full DFA", lambda: B().kind(ac.AhoCorasickKind.DFA).build(pats)),
            cost_of("c4: 100000 patterns, contiguous NFA (the device derives the full DFA at upload)",
                    lambda: B().kind(ac.AhoCorasickKind.ContiguousNFA).build(pats4c)),
            cost_of("c5: 1000 patterns, casei LeftmostFirst DFA (+ its Standard twin)",
                    lambda: B().kind(ac.AhoCorasickKind.DFA).match_kind(ac.MatchKind.LeftmostFirst).ascii_case_insensitive(True).build(pats),
                    lambda a_: a_.find_iter(buf[:4096], as_numpy=True)),
        ]
        del pats4c

    # ---- side measurements on the same resident haystack (N=1 only, after the timed region): the other count engines
    # of the headline workload, and BASELINE configs 4 and 5.  Synchronous calls; kernel time from the HIP events the
    # library records around the count kernel on the launch stream (acgpu_profile.ms_scan).
    if world == 1 and not args.no_also and args.workload == "c2":
        K = max(args.also_steps, 3)

        def roof(kms, tag=None, kernel=None, nbytes=None):
            nbytes = shard if nbytes is None else nbytes
            ach = nbytes / (kms * 1e-3) / 1e9
            r = {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 5), "kernel_ms": round(kms, 4), "algorithmic_bytes_per_launch": nbytes}
            if tag:
                t, src, k = committed_traffic(tag, kernel, nbytes / (1 << 30))
                r.update({"kernel": k or kernel, "traffic": t, "traffic_source": src})
            return r

        def cpu_side(make_oracle, run, sample_mib, what):
            """the oracle beside a parity config: one host core, median of 5 passes over a bounded sample"""
            if args.no_cpu_baseline:
                return None
            from oracle import orc   # (the checker, timed as a CPU baseline)
            sample = min(sample_mib << 20, shard)
            host = buf[:sample].cpu().numpy()
            o = make_oracle(orc)
            ts = []
            for _ in range(5):
                t1 = time.perf_counter()
                n_cpu = run(o, host)
                ts.append(time.perf_counter() - t1)
            return {"value": round(sample / sorted(ts)[2] / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                    "sample": f"first {sample >> 20} MiB of the same haystack, median of 5 passes, {what}", "matches_in_sample": int(n_cpu)}

        def timed(call, steps):
            p = _lib.CProfile()
            for _ in range(3):
                nres = call(p)
            torch.cuda.synchronize()
            kms, t1 = [], time.perf_counter()
            for _ in range(steps):
                nres = call(p)
                kms.append(p.ms_scan)
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t1) / steps
            return nres, float(np.mean(kms)), dt1 * 1e3, int(p.engine_used)

        engines = {}
        for name, steps in (("pf", K), ("hot", K), ("walk", 3)):
            try:
                a2 = (ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(ac.MatchKind.Standard)
                      .gpu_engine(name).build(pats))
                nres, kms, ms, eng = timed(lambda p: a2.overlapping_device(buf, out=out, profile=p)[0], steps)
                engines[name] = {"kernel": {4: "k_pf_count", 3: "k_lw_count", 1: "k_tri_walk<DfaTriDev>"}.get(eng, str(eng)),
                                 "ms_per_step": round(ms, 4), "value": round(shard / ms / 1e6, 3), "matches": int(nres),
                                 "parity_with_timed_run": bool(int(nres) == int(n_matches)),
                                 **roof(kms, *{4: ("pf", "k_pf_count<false, false>"), 3: ("hot", "k_lw_count<"), 1: ("dfa_tri", "k_tri_walk<")}.get(eng, (None, None)))}
                del a2
            except Exception as exc:
                engines[name] = {"error": str(exc)}
        result["engines"] = engines

        also = []
        try:   # config 4: 100 000 patterns, AhoCorasickKind::ContiguousNFA
            pats4 = ac.gen_patterns(100000, seed=0xAC04)
            for name, steps, label in (("auto", K, "default engine (prefix filter over the DFA derived from the same noncontiguous NFA)"),
                                       ("walk", 2, "contiguous-NFA failure-link walk, src/nfa/contiguous.rs:186-247")):
                a4 = (ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.ContiguousNFA).match_kind(ac.MatchKind.Standard)
                      .gpu_engine(name).build(pats4))
                nres, kms, ms, eng = timed(lambda p: a4.overlapping_device(buf, out=out, profile=p)[0], steps)
                also.append({"workload": f"c4 = configs[3]: 100000 patterns, ContiguousNFA, overlapping, 8 GiB; {label}",
                             "config": {"haystack_gib": args.gib, "patterns": 100000, "engine_requested": name},
                             "engine": eng, "value": round(shard / ms / 1e6, 3), "unit": "GB/s", "ms_per_step": round(ms, 4),
                             "matches": int(nres), "roofline": roof(kms, *(("c4_pfx", "k_pfx_count<false") if eng == 4 else ("c4_cnfa_tri", "k_tri_walk<")))})
                del a4
            also[-1]["cpu_baseline"] = cpu_side(lambda orc: orc.Oracle(pats4, kind=orc.KIND_CNFA),
                                                lambda o, h: len(o.find_overlapping_iter(h, as_numpy=True)), 128,
                                                "oracle contiguous-NFA overlapping loop (contiguous.rs:186-247 in automaton.rs:1491-1534)")
        except Exception as exc:
            also.append({"workload": "c4", "error": str(exc)})
        try:   # config 5: casei LeftmostFirst find_iter
            a5 = (ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).match_kind(ac.MatchKind.LeftmostFirst)
                  .ascii_case_insensitive(True).build(pats))
            nres, kms, ms, eng = timed(lambda p: a5.find_iter_device(buf, out, profile=p)[0], K)
            also.append({"workload": "c5 = configs[4]: 1000 patterns, ascii_case_insensitive + LeftmostFirst, find_iter, 8 GiB "
                                     "(occurrence stream of the Standard twin + device selection)",
                         "config": {"haystack_gib": args.gib, "patterns": args.patterns,
                                    "call": "a repeated identical call: from its second call on the pipeline is queued sized by the previous "
                                            "call's occurrence stream and synchronised once (capi_find.cpp: nonoverlapping_guessed); a first "
                                            "call, or one whose stream outgrows the guess, takes two round trips"},
                         "engine": eng, "value": round(shard / ms / 1e6, 3), "unit": "GB/s", "ms_per_step": round(ms, 4),
                         "matches": int(nres), "roofline": roof(kms, "c5_pf", "k_pf_count<false, true>"),
                         "cpu_baseline": cpu_side(lambda orc: orc.Oracle(pats, kind=orc.KIND_DFA, match_kind=1, ascii_case_insensitive=True),
                                                  lambda o, h: len(o.find_iter(h, as_numpy=True)), 256,
                                                  "oracle FindIter over the casei LeftmostFirst DFA (automaton.rs:857-936)")})
        except Exception as exc:
            also.append({"workload": "c5", "error": str(exc)})
        try:   # the reference's own benchmark inputs: natural text against a dictionary (benchmarks/haystacks, benchmarks/regexes)
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
            import corpora
            ngib = 1 << 30
            for hay_name, words_name in (("sherlock.txt", "words-5000"), ("en-huge.txt", "words-15000")):
                text = corpora.haystack(hay_name)
                nat = torch.from_numpy(np.tile(text, -(-ngib // len(text)))[:ngib].copy()).cuda()
                a6 = ac.AhoCorasick.builder().match_kind(ac.MatchKind.Standard).build(corpora.words(words_name))
                nres, kms, ms, eng = timed(lambda p: a6.overlapping_device(nat, out=out, profile=p)[0], K)
                line = {"workload": f"natural text: {hay_name} tiled to 1 GiB / {words_name} (the reference's benchmark corpora), "
                                    "overlapping, default engine",
                        "config": {"haystack_gib": 1.0, "patterns": len(corpora.words(words_name)),
                                   "note": "1 GiB steps are short: the clocks ramp less than on the 8 GiB lines (10-20 % below the rate of an 8 GiB tiling)"},
                        "engine": eng, "value": round(ngib / ms / 1e6, 3), "unit": "GB/s", "ms_per_step": round(ms, 4),
                        "matches": int(nres),
                        "roofline": roof(kms, "nat_" + hay_name.split(".")[0].replace("-", ""), "k_pfx_count<true", ngib)}
                # the pipelined (enqueue-only) form of the same search: probe + gated filters + bucket order pass, no host decision
                tot6 = torch.zeros(2, dtype=torch.int64, device=dev)
                for i in range(3):
                    a6.overlapping_enqueue(nat, out, tot6)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(K):
                    a6.overlapping_enqueue(nat, out, tot6)
                torch.cuda.synchronize()
                ems = (time.perf_counter() - t1) / K * 1e3
                th = tot6.cpu().numpy()
                line["enqueue_form"] = {"ms_per_step": round(ems, 4), "value": round(ngib / ems / 1e6, 3), "unit": "GB/s",
                                        "delivered": bool(int(th[0]) == int(nres) and int(th[1]) == 0)}
                also.append(line)
                del nat, a6
        except Exception as exc:
            also.append({"workload": "natural text", "error": str(exc)})
        try:   # the reference's small-set definitions (benchmarks/definitions/teddy.toml): completed calls of the engine
            #    they run on -- the LDS walk, one row per state, records from the events of its count walk (device/lds_emit.hip)
            n256 = 256 << 20
            defs = {b["name"]: b for fam, bs in corpora.bench_defs().items() for b in bs if fam == "teddy"}
            for name in ("teddy3-1pat-common", "teddy1-16pat-uncommon", "teddy1-1pat-common", "teddy1-1pat-uncommon"):
                b = defs[name]
                dpats, text = corpora.bench_patterns(b), corpora.bench_haystack(b)
                dh = torch.from_numpy(np.tile(text, -(-n256 // len(text)))[:n256].copy()).cuda()
                a7 = ac.AhoCorasick.builder().build(dpats)
                a7l = ac.AhoCorasick.builder().match_kind(ac.MatchKind.LeftmostFirst).build(dpats)
                need, _ = a7.overlapping_device(dh, out=None)   # (a count first: the record buffer of the other lines holds 2.8 M)
                out7 = out if need * 24 <= out.numel() else torch.empty(need * 24 + 4096, dtype=torch.uint8, device=dev)

                def completed(call):
                    n_, ok_ = call()
                    assert ok_, "record buffer too small"
                    return n_
                nres, kms, ms, eng = timed(lambda p: completed(lambda: a7.overlapping_device(dh, out=out7, profile=p)), K)
                nlf, _, mslf, _ = timed(lambda p: completed(lambda: a7l.find_iter_device(dh, out7, profile=p)), K)
                also.append({"workload": f"reference definition {name}: {len(dpats)} pattern(s), its haystack tiled to 256 MiB, completed "
                                         "find_overlapping_iter and LeftmostFirst find_iter calls (records on the device)",
                             "config": {"haystack_gib": 0.25, "patterns": len(dpats)}, "engine": eng, "unit": "GB/s",
                             "value": round(n256 / ms / 1e6, 3), "ms_per_step": round(ms, 4), "matches": int(nres),
                             "find_iter": {"value": round(n256 / mslf / 1e6, 3), "ms_per_step": round(mslf, 4), "matches": int(nlf)},
                             "roofline": roof(kms, "lw_ev_" + name.replace("-", "_"), "k_lw_count_ev<", n256)})
                del dh, a7, a7l, out7
        except Exception as exc:
            also.append({"workload": "reference small-set definitions", "error": str(exc)})
        result["also"] = also
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
