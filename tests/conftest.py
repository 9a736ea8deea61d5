import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no lib/libacgpu.so (built artefacts are git-ignored): build it once, as
    # __graft_entry__.build() would (hipcc cross-compiles gfx950 without a GPU)
    lib = os.path.join(ROOT, "aho-corasick_amd", "lib", "libacgpu.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "aho-corasick_amd", "csrc"), "-j8"],
                              stdout=subprocess.DEVNULL)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Without a GPU the -m gpu tests are skipped.  With one, each of them gets a time limit (pytest-timeout, thread
    method: the process is ended even when it hangs inside a HIP call), so that a deadlocked kernel costs minutes of GPU
    time, not the round's budget."""
    has_gpu = _has_gpu()
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if item.get_closest_marker("gpu") is None:
            continue
        if not has_gpu:
            item.add_marker(skip)
        elif item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900 if "fullsize" in item.nodeid else 300, method="thread"))
