"""-m gpu: the bounds-checked flavour of the library (make -C aho-corasick_amd/csrc guard, -DACGPU_GUARD): every
haystack access of every kernel is checked on the device against the 16-byte-aligned hull of the searched span
(bounds-checked debug kernels; the reference has Rust's bounds checks).  The workload
(tests/guard_workload.py) covers every engine, misaligned pointers, spans, shards, find_iter, replace_all and the stream
search, each compared with the oracle; it must finish with zero violations."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARD = os.path.join(ROOT, "aho-corasick_amd", "lib", "libacgpu_guard.so")


def test_normal_build_reports_no_guard():
    import aho_corasick_amd as ac
    assert ac.load_library().acgpu_guard_violations() == -1


def test_every_engine_stays_inside_the_span_hull():
    assert os.path.exists(GUARD), f"{GUARD} is missing: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, ACGPU_LIB=GUARD)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "guard_workload.py")], env=env, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = dict(l.split() for l in r.stdout.splitlines() if l.startswith(("calls", "violations")))
    assert int(lines["calls"]) > 400
    assert int(lines["violations"]) == 0, r.stdout[-2000:]


def test_guard_notices_an_access_outside_its_hull():
    """Positive control: with the permitted hull shrunk by 16 bytes on both sides (ACGPU_GUARD_SHRINK) the same kernels
    are reported -- the counters are live, the zero above means something."""
    code = ("import numpy as np, torch, aho_corasick_amd as ac\n"
            "a = ac.AhoCorasick.builder().kind(ac.AhoCorasickKind.DFA).build([b'needle', b'hay'])\n"
            "h = torch.from_numpy(np.frombuffer(b'a needle in a haystack ' * 4000, dtype=np.uint8).copy()).cuda()\n"
            "n = len(a.find_overlapping_iter(h, as_numpy=True))\n"
            "print('matches', n); print('violations', ac.load_library().acgpu_guard_violations())\n")
    env = dict(os.environ, ACGPU_LIB=GUARD, ACGPU_GUARD_SHRINK="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = dict(l.split() for l in r.stdout.splitlines() if l.startswith(("matches", "violations")))
    assert int(lines["matches"]) == 8000 and int(lines["violations"]) > 0
