"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY. May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORC_SANITIZE=1: the -fsanitize=address,undefined build of the same source (make -C oracle asan); the process must
# have libasan preloaded (tests/test_oracle_asan.py arranges that in a subprocess)
_SAN = os.environ.get("ORC_SANITIZE") == "1"
_LIB_PATH = os.path.join(_HERE, "liborc_asan.so" if _SAN else "liborc.so")

STANDARD, LEFTMOST_FIRST, LEFTMOST_LONGEST = 0, 1, 2
START_BOTH, START_UNANCHORED, START_ANCHORED = 0, 1, 2
KIND_AUTO, KIND_NNFA, KIND_CNFA, KIND_DFA = 0, 1, 2, 3

ERR_NAMES = {
    1: "StateIDOverflow", 2: "PatternIDOverflow", 3: "PatternTooLong",
    10: "InvalidInputAnchored", 11: "InvalidInputUnanchored", 12: "UnsupportedStream",
    13: "UnsupportedOverlapping", 14: "UnsupportedEmpty", 20: "InvalidSpan", 30: "NoMem",
}


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(ERR_NAMES.get(code, str(code)))
        self.code = code
        self.kind = ERR_NAMES.get(code, str(code))


class Config(C.Structure):
    _fields_ = [("match_kind", C.c_int), ("start_kind", C.c_int), ("kind", C.c_int),
                ("ascii_case_insensitive", C.c_int), ("byte_classes", C.c_int), ("prefilter", C.c_int),
                ("dense_depth_set", C.c_int), ("dense_depth", C.c_uint32)]


class Match(C.Structure):
    _fields_ = [("pattern", C.c_uint32), ("_pad", C.c_uint32), ("start", C.c_uint64), ("end", C.c_uint64)]


class Tables(C.Structure):
    _fields_ = [
        ("nnfa_states", C.c_size_t),
        ("max_match_id", C.c_uint32), ("start_unanchored_id", C.c_uint32), ("start_anchored_id", C.c_uint32),
        ("byte_classes", C.c_uint8 * 256),
        ("alphabet_len", C.c_size_t),
        ("dfa_state_len", C.c_size_t), ("dfa_stride2", C.c_size_t),
        ("dfa_trans", C.POINTER(C.c_uint32)), ("dfa_trans_len", C.c_size_t),
        ("dfa_max_match_id", C.c_uint32), ("dfa_start_unanchored_id", C.c_uint32),
        ("dfa_start_anchored_id", C.c_uint32),
        ("dfa_match_off", C.POINTER(C.c_uint32)), ("dfa_match_pid", C.POINTER(C.c_uint32)),
        ("dfa_num_match_states", C.c_size_t),
        ("cnfa_repr", C.POINTER(C.c_uint32)), ("cnfa_repr_len", C.c_size_t),
        ("cnfa_max_match_id", C.c_uint32), ("cnfa_start_unanchored_id", C.c_uint32),
        ("cnfa_start_anchored_id", C.c_uint32),
        ("pattern_lens", C.POINTER(C.c_uint32)),
    ]


def build_lib(force=False):
    src = os.path.join(_HERE, "ac_oracle.c")
    hdr = os.path.join(_HERE, "ac_oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", os.path.basename(_LIB_PATH)])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_lib()
        L = C.CDLL(_LIB_PATH)
        L.orc_build.argtypes = [C.POINTER(Config), C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t,
                                C.POINTER(C.c_void_p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_free.restype = None
        for f in ("orc_kind", "orc_match_kind", "orc_start_kind"):
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_patterns_len", "orc_min_pattern_len", "orc_max_pattern_len", "orc_memory_usage"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_size_t
        L.orc_find.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                               C.POINTER(C.c_int), C.POINTER(Match)]
        for f in ("orc_find_iter", "orc_find_overlapping_iter"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                      C.POINTER(Match), C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_find_iter_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                       C.POINTER(Match), C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_dfa_overlapping_count.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_find_overlapping_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint,
                                                    C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
        L.orc_hash_matches.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_hash_matches.restype = C.c_uint64
        L.orc_gen_haystack_parallel.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32,
                                                C.c_uint]
        L.orc_gen_haystack_parallel.restype = None
        L.orc_dfa_overlapping_count_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                                         C.c_uint, C.POINTER(C.c_uint64)]
        L.orc_get_tables.argtypes = [C.c_void_p, C.POINTER(Tables)]
        L.orc_get_tables.restype = None
        L.orc_nnfa_state.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.c_size_t]
        L.orc_nnfa_state.restype = C.c_size_t
        L.orc_nnfa_next_state.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint8]
        L.orc_nnfa_next_state.restype = C.c_uint32
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_gen_haystack.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_gen_haystack.restype = None
        L.orc_gen_patterns.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, C.c_uint64,
                                       C.c_uint32, C.c_uint32]
        L.orc_gen_patterns.restype = C.c_size_t
        _lib = L
    return _lib


def _buf_ptr(hay):
    """Return (ctypes pointer-ish, length, keepalive) for bytes / bytearray / numpy uint8."""
    try:
        import numpy as np
        if isinstance(hay, np.ndarray):
            assert hay.dtype == np.uint8 and hay.flags["C_CONTIGUOUS"]
            return C.c_void_p(hay.ctypes.data), hay.size, hay
    except ImportError:
        pass
    if isinstance(hay, str):
        hay = hay.encode()
    b = bytes(hay)
    return C.cast(C.c_char_p(b), C.c_void_p), len(b), b


class Oracle:
    """AhoCorasick look-alike backed by the C oracle."""

    def __init__(self, patterns, match_kind=STANDARD, start_kind=START_UNANCHORED, kind=KIND_AUTO,
                 ascii_case_insensitive=False, byte_classes=True, prefilter=True, dense_depth=None):
        L = lib()
        cfg = Config(match_kind, start_kind, kind, int(ascii_case_insensitive), int(byte_classes), int(prefilter),
                     0 if dense_depth is None else 1,
                     0 if dense_depth is None else min(int(dense_depth), 0xFFFFFFFF))
        pats = [p.encode() if isinstance(p, str) else bytes(p) for p in patterns]
        n = len(pats)
        arr = (C.c_char_p * max(n, 1))(*pats) if n else (C.c_char_p * 1)()
        lens = (C.c_size_t * max(n, 1))(*[len(p) for p in pats]) if n else (C.c_size_t * 1)()
        h = C.c_void_p()
        rc = L.orc_build(C.byref(cfg), arr, lens, n, C.byref(h))
        if rc:
            raise OracleError(rc)
        self._h = h
        self._L = L
        self.patterns = pats

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_free(self._h)
            self._h = None

    kind = property(lambda s: s._L.orc_kind(s._h))
    match_kind = property(lambda s: s._L.orc_match_kind(s._h))
    start_kind = property(lambda s: s._L.orc_start_kind(s._h))
    patterns_len = property(lambda s: s._L.orc_patterns_len(s._h))
    min_pattern_len = property(lambda s: s._L.orc_min_pattern_len(s._h))
    max_pattern_len = property(lambda s: s._L.orc_max_pattern_len(s._h))
    memory_usage = property(lambda s: s._L.orc_memory_usage(s._h))

    def _span(self, n, span):
        if span is None:
            return 0, n
        return span

    def find(self, hay, span=None, anchored=False, earliest=False):
        p, n, keep = _buf_ptr(hay)
        s, e = self._span(n, span)
        found = C.c_int()
        m = Match()
        rc = self._L.orc_find(self._h, p, n, s, e, int(anchored), int(earliest), C.byref(found), C.byref(m))
        if rc:
            raise OracleError(rc)
        return (m.pattern, m.start, m.end) if found.value else None

    def _collect(self, fn, hay, span, anchored, as_numpy=False):
        p, n, keep = _buf_ptr(hay)
        s, e = self._span(n, span)
        cap = 1024
        while True:
            out = (Match * cap)()
            nout = C.c_size_t()
            rc = fn(self._h, p, n, s, e, int(anchored), out, cap, C.byref(nout))
            if rc:
                raise OracleError(rc)
            if nout.value <= cap:
                break
            cap = nout.value
        if as_numpy:
            import numpy as np
            a = np.frombuffer(out, dtype=np.dtype([("pattern", "<u4"), ("_pad", "<u4"), ("start", "<u8"),
                                                   ("end", "<u8")]), count=nout.value)
            return a.copy()
        return [(out[i].pattern, out[i].start, out[i].end) for i in range(nout.value)]

    def find_iter(self, hay, span=None, anchored=False, as_numpy=False, earliest=False):
        if earliest:
            fn = lambda h, p, n, s, e, a, out, cap, nout: self._L.orc_find_iter_ex(h, p, n, s, e, a, 1, out, cap, nout)
            return self._collect(fn, hay, span, anchored, as_numpy)
        return self._collect(self._L.orc_find_iter, hay, span, anchored, as_numpy)

    def find_overlapping_iter(self, hay, span=None, anchored=False, as_numpy=False):
        return self._collect(self._L.orc_find_overlapping_iter, hay, span, anchored, as_numpy)

    def dfa_overlapping_count(self, hay, span=None):
        p, n, keep = _buf_ptr(hay)
        s, e = self._span(n, span)
        cnt, h = C.c_uint64(), C.c_uint64()
        rc = self._L.orc_dfa_overlapping_count(self._h, p, n, s, e, C.byref(cnt), C.byref(h))
        if rc:
            raise OracleError(rc)
        return cnt.value, h.value

    def dfa_overlapping_count_parallel(self, hay, threads, span=None):
        p, n, keep = _buf_ptr(hay)
        s, e = self._span(n, span)
        cnt = C.c_uint64()
        rc = self._L.orc_dfa_overlapping_count_parallel(self._h, p, n, s, e, int(threads), C.byref(cnt))
        if rc:
            raise OracleError(rc)
        return cnt.value

    def find_overlapping_parallel(self, hay, threads=None, span=None):
        """Chunk-parallel find_overlapping_iter(..).collect() (any automaton kind, non-empty patterns): returns
        (records as a numpy MATCH array in iterator order, order-sensitive FNV-1a hash of them)."""
        import numpy as np
        p, n, keep = _buf_ptr(hay)
        s, e = self._span(n, span)
        threads = threads or (os.cpu_count() or 1)
        dt = np.dtype([("pattern", "<u4"), ("_pad", "<u4"), ("start", "<u8"), ("end", "<u8")])
        cap = 1 << 16
        while True:
            out = np.empty(cap, dtype=dt)
            nout, h = C.c_size_t(), C.c_uint64()
            rc = self._L.orc_find_overlapping_parallel(self._h, p, n, s, e, int(threads), C.c_void_p(out.ctypes.data), cap,
                                                       C.byref(nout), C.byref(h))
            if rc:
                raise OracleError(rc)
            if nout.value <= cap:
                return out[:nout.value], h.value
            cap = nout.value

    def tables(self):
        t = Tables()
        self._L.orc_get_tables(self._h, C.byref(t))
        return t

    def nnfa_state(self, sid):
        fail, depth = C.c_uint32(), C.c_uint32()
        n = self._L.orc_nnfa_state(self._h, sid, C.byref(fail), C.byref(depth), None, 0)
        pids = (C.c_uint32 * max(n, 1))()
        self._L.orc_nnfa_state(self._h, sid, C.byref(fail), C.byref(depth), pids, n)
        return fail.value, depth.value, list(pids[:n])


def splitmix64(x):
    return lib().orc_splitmix64(x & 0xFFFFFFFFFFFFFFFF)


def replace_all_bytes(oracle, hay, replace_with, utf8_boundaries=False):
    """Automaton::try_replace_all_with_bytes / try_replace_all_with (src/automaton.rs:493-550) restated over the
    oracle's find_iter: the checker for acgpu_replace_all."""
    hay = bytes(hay)
    repl = [r.encode() if isinstance(r, str) else bytes(r) for r in replace_with]
    assert len(repl) == oracle.patterns_len, "replace_all requires a replacement for every pattern"

    def boundary(i):
        return i == 0 or i >= len(hay) or (hay[i] & 0xC0) != 0x80

    out, last = [], 0
    arr = oracle.find_iter(hay, as_numpy=True)
    for p, s, e in zip(arr["pattern"].tolist(), arr["start"].tolist(), arr["end"].tolist()):
        if utf8_boundaries and not (boundary(s) and boundary(e)):
            continue
        out.append(hay[last:s])
        last = e
        out.append(repl[p])
    out.append(hay[last:])
    return b"".join(out)


def gen_haystack(offset, length, seed=0xAC02, lo=0x20, span=95):
    import numpy as np
    a = np.empty(length, dtype=np.uint8)
    if length >= (64 << 20):   # large inputs of the full-size parity tests: all host cores
        lib().orc_gen_haystack_parallel(C.c_void_p(a.ctypes.data), offset, length, seed, lo, span, os.cpu_count() or 1)
    else:
        lib().orc_gen_haystack(C.c_void_p(a.ctypes.data), offset, length, seed, lo, span)
    return a


def hash_matches(arr):
    """Order-sensitive FNV-1a over (pattern, start, end) of a numpy MATCH array (24-byte records)."""
    import numpy as np
    a = np.ascontiguousarray(arr)
    assert a.dtype.itemsize == 24
    return lib().orc_hash_matches(C.c_void_p(a.ctypes.data), len(a))


def gen_patterns(n, seed=0xAC01, lo=0x20, span=95):
    L = lib()
    lens = (C.c_uint32 * max(n, 1))()
    total = L.orc_gen_patterns(None, 0, lens, n, seed, lo, span)
    buf = C.create_string_buffer(max(total, 1))
    L.orc_gen_patterns(buf, total, lens, n, seed, lo, span)
    raw = buf.raw[:total]
    out, pos = [], 0
    for i in range(n):
        out.append(raw[pos:pos + lens[i]])
        pos += lens[i]
    return out
