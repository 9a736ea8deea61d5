"""acgpu_find_overlapping_multi from a plain-C caller (tests/c/multi_test.c: include/acgpu.h + libacgpu.so only).

CPU: the program compiles with gcc -std=c11, links against libacgpu.so alone and runs its no-device mode.
GPU: N virtual shards on device 0 (what a 1-GPU box can run; the same code path as N devices up to the transport) must
reproduce the single-call stream AND the CPU oracle's stream over the regenerated input (count + order-sensitive hash);
with ACGPU_MULTI_FORCE_RCCL=1 the records of a one-shard call travel through RCCL."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "multi_test.c")
EXE = os.path.join(ROOT, "tests", "c", "_multi_test")
LIBDIR = os.path.join(ROOT, "aho-corasick_amd", "lib")


def compile_test():
    lib = os.path.join(LIBDIR, "libacgpu.so")
    if not os.path.exists(EXE) or any(os.path.getmtime(s) > os.path.getmtime(EXE) for s in (SRC, lib, os.path.join(ROOT, "include", "acgpu.h"))):
        subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE,
                        "-L" + LIBDIR, "-lacgpu", "-Wl,-rpath," + LIBDIR], check=True)


def test_c_caller_compiles_and_links():
    compile_test()
    out = subprocess.run([EXE, "--list"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("multi_test: abi ")


def oracle_stream(n_shards, mib):
    """What multi_test.c searches, rebuilt on the host: the headline pattern generator, the device haystack generator's
    formula and the program's planting loop -- and the oracle's overlapping stream over it (count, hash)."""
    from oracle import orc
    n = mib << 20
    pats = orc.gen_patterns(1000, seed=0xAC01)
    hay = orc.gen_haystack(0, n, seed=0xAC02)
    for k in range(1, 4 * n_shards + 1):
        pos = k * (n // (4 * n_shards + 1))
        if k % 4 == 0:
            pos = (k // 4) * (n // n_shards) - 1 - (k % 7)
        p = pats[(7 * k) % 1000]
        if pos + len(p) <= n:
            hay[pos:pos + len(p)] = np.frombuffer(p, dtype=np.uint8)
    o = orc.Oracle(pats, kind=orc.KIND_DFA)
    want = o.find_overlapping_iter(hay, as_numpy=True)
    return len(want), orc.hash_matches(want)


def check_against_oracle(stdout, n_shards, mib):
    m = re.search(r"(\d+) records \(single call \d+\), transport \d, hash ([0-9a-f]{16})", stdout)
    assert m, stdout
    count, h = oracle_stream(n_shards, mib)
    assert int(m.group(1)) == count and int(m.group(2), 16) == h, (stdout, count, hex(h))


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_virtual_shards_equal_single_call(shards):
    compile_test()
    r = subprocess.run([EXE, str(shards), "96"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical" in r.stdout and "transport 1" in r.stdout, r.stdout
    check_against_oracle(r.stdout, shards, 96)


@pytest.mark.gpu
def test_rccl_transport_executes():
    """One shard, RCCL forced: ncclCommInitAll over {0}, grouped ncclSend/ncclRecv to self on the device's stream."""
    compile_test()
    env = dict(os.environ, ACGPU_MULTI_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([EXE, "1", "96"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical" in r.stdout and "transport 2" in r.stdout, r.stdout
    check_against_oracle(r.stdout, 1, 96)
