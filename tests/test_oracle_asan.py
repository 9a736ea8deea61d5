"""The oracle under AddressSanitizer + UBSan (sanitizer build of the checker).

oracle/ac_oracle.c is ~1 500 lines of hand-rolled C vectors; this runs the oracle's own test files against
`make -C oracle asan` (liborc_asan.so, -fsanitize=address,undefined) in a subprocess with the sanitizer runtimes
preloaded.  Any out-of-bounds access, use-after-free or undefined shift/overflow in the checker aborts that run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_suite_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=":".join(x for x in (asan, ubsan) if x), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ORC_SANITIZE="1")
    probe = subprocess.run([sys.executable, "-c", "from oracle import orc; orc.lib(); print(orc._LIB_PATH)"],
                           cwd=ROOT, env=env, capture_output=True, text=True)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("liborc_asan.so"), probe.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        "tests/test_oracle_golden.py", "tests/test_oracle_naive.py", "tests/test_oracle_parallel.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout
