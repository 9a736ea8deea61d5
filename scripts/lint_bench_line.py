#!/usr/bin/env python3
"""Lint of a bench line kept under profiles/ (run it when the profiles are regenerated; it checks ARTIFACTS, not code, so
it is not part of the test suites): the line printed by `python bench.py` on an MI355X honours the contract bench.py is
held to -- one JSON object with the metric / value / config of BASELINE.json, a `roofline` whose fraction is achieved /
peak on algorithmic bytes, a `cpu_baseline` measured on the same box, and `also` / `engines` entries whose arithmetic is
consistent.  usage: lint_bench_line.py profiles/rNN_bench.json [per-launch csv of k_pf_count + the traced run's line]"""
import csv
import json
import sys


def last_line(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def lint(path):
    d = last_line(path)
    assert d["unit"] == "GB/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and d["dtype"] == "u8" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    shard = int(d["config"]["haystack_gib_per_gpu"] * (1 << 30))
    assert abs(d["value"] - shard / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01      # value = bytes per step / step time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["algorithmic_bytes_per_launch"] == shard
    assert abs(r["achieved"] - shard / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] <= 1.0
    if r.get("empirical_peak"):
        assert abs(r["frac_of_empirical"] - r["achieved"] / r["empirical_peak"]) < 1e-3
    assert r["traffic"] is None or 0.9 * shard < r["traffic"] < 2.0 * shard
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001                                          # a step is not faster than its kernel
    c = d.get("cpu_baseline")
    if c:
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "GB/s" and c["value"] > 0
        assert c.get("sample_parity", True) is True
    for a in d.get("also", []):
        if "error" in a:
            raise AssertionError(f"also line failed: {a}")
        assert a["unit"] == "GB/s" and a["roofline"]["kernel_ms"] <= a["ms_per_step"] * 1.001 and a["roofline"]["frac"] <= 1.0
        if "enqueue_form" in a:
            assert a["enqueue_form"]["delivered"] is True
        check_traffic(a["roofline"], a["workload"][:40])
    for name, e in d.get("engines", {}).items():
        assert "error" not in e and e["parity_with_timed_run"], (name, e)
        check_traffic(e, "engines." + name)
    check_traffic(r, "headline")
    return d


def check_traffic(r, what):
    """a roofline that cites a counter file: the file exists, names the kernel of the line (template arguments included) and
    the same launch size, and its bytes are at least the algorithmic ones"""
    import os
    src = r.get("traffic_source")
    if r.get("traffic") is None:
        print(f"  note: {what}: no counter traffic")
        return
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), src.split(" ")[0])
    assert os.path.exists(path), (what, src)
    f = json.load(open(path))
    assert abs(f["hbm_read_bytes"] - r["traffic"]) < 1.0, (what, src)
    if "_kernel" in f:
        want = r.get("kernel", "")
        assert (f["_kernel"] or "").split("<")[0].split("::")[-1] in want or want in (f["_kernel"] or ""), (what, f["_kernel"], want)
    if "_gib" in f:
        assert abs(f["_gib"] * (1 << 30) - r["algorithmic_bytes_per_launch"]) < 1.0, (what, f["_gib"])
    assert 0.9 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 4.0 * r["algorithmic_bytes_per_launch"], (what, r["traffic"])


def lint_trace(launch_csv, traced_line):
    """rocprofv3's per-launch durations of the traced run: the last `steps` launches are the timed steps; their mean agrees
    with the HIP-event kernel time the same run printed"""
    line = last_line(traced_line)
    rows = [int(r["duration_ns"]) for r in csv.DictReader(open(launch_csv))][-line["steps"]:]
    assert len(rows) == line["steps"]
    assert abs(sum(rows) / len(rows) / 1e6 - line["roofline"]["kernel_ms"]) / line["roofline"]["kernel_ms"] < 0.03


if __name__ == "__main__":
    d = lint(sys.argv[1])
    if len(sys.argv) > 3:
        lint_trace(sys.argv[2], sys.argv[3])
    print(f"ok: {sys.argv[1]}  value {d['value']} GB/s, roofline.frac {d['roofline']['frac']}")
