"""The library's HOST code under AddressSanitizer + UBSan: `make -C aho-corasick_amd/csrc hostasan`
builds libacgpu_hostasan.so with every .cpp source instrumented -- the C ABI, automaton construction (host/builder.cpp)
and the table builders of every engine with their CPU models (host/lw_tables.cpp, pf_tables.cpp, cnfa_tables.cpp) -- and
this runs the not-gpu test files that drive them in a subprocess with the sanitizer runtime preloaded.  An
out-of-bounds table index, a use-after-free or an undefined shift in that code aborts the run."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "aho-corasick_amd", "lib", "libacgpu_hostasan.so")
FILES = ["tests/test_lw_tables.py", "tests/test_pf_tables.py", "tests/test_cnfa_tables.py", "tests/test_tables_parity.py",
         "tests/test_lib_abi.py", "tests/test_select_rule.py"]


def test_host_code_under_asan_ubsan():
    rts = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rts:
        pytest.skip("no clang AddressSanitizer runtime in this toolchain")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "aho-corasick_amd", "csrc"), "-s", "-j8", "hostasan"])
    deps = subprocess.run(["ldd", LIB], capture_output=True, text=True).stdout
    assert "libclang_rt.asan" in deps, deps                      # the build really is instrumented
    env = dict(os.environ, LD_PRELOAD=rts[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ACGPU_LIB=LIB)
    files = [f for f in FILES if os.path.exists(os.path.join(ROOT, f))]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] + files,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
