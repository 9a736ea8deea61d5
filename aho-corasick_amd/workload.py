"""Synthetic workloads of SURVEY.md Appendix C (what bench.py scans): counter-based, so any sub-range can be produced
independently on any device.  byte i = lo + splitmix64(seed ^ i) % span; patterns draw their length (4 + x % 13) and
bytes from one running counter.  The haystack itself is generated on the device (api.gen_haystack); this module is the
host side -- pure numpy, no dependency on the test oracle (whose generator the tests compare it with)."""
import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(x):
    """Vectorised splitmix64 finaliser over numpy uint64 (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def gen_patterns(n, seed=0xAC01, lo=0x20, span=95):
    """n patterns of 4..16 bytes over [lo, lo+span): the BASELINE pattern sets (seed 0xAC01: configs 2/3/5, 0xAC04: config 4)."""
    # lengths depend on the running counter, so draw generously and walk it (vectorised draw, scalar walk over n)
    draws = splitmix64(np.uint64(seed) ^ np.arange(n * 17 + 17, dtype=np.uint64))
    out, ctr = [], 0
    for _ in range(n):
        length = 4 + int(draws[ctr] % np.uint64(13))
        ctr += 1
        out.append((np.uint64(lo) + draws[ctr:ctr + length] % np.uint64(span)).astype(np.uint8).tobytes())
        ctr += length
    return out


def gen_haystack_host(offset, length, seed=0xAC02, lo=0x20, span=95):
    """Host copy of bytes [offset, offset+length) of the synthetic haystack (numpy uint8)."""
    idx = np.uint64(offset) + np.arange(length, dtype=np.uint64)
    return (np.uint64(lo) + splitmix64(np.uint64(seed) ^ idx) % np.uint64(span)).astype(np.uint8)
