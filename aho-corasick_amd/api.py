"""Host-side mirror of the reference facade (src/ahocorasick.rs) on top of the C ABI.

Names, argument meaning and error behaviour follow the reference:
  AhoCorasick::{new, builder, is_match, find, find_iter, find_overlapping_iter, try_*}   :243-1357
  AhoCorasickBuilder::{match_kind, start_kind, ascii_case_insensitive, kind, prefilter,
                       dense_depth, byte_classes, build}                                 :2135-2616
  getters kind/start_kind/match_kind/min_pattern_len/max_pattern_len/patterns_len/memory_usage :1867-2027
The infallible forms of the reference panic on error (src/ahocorasick.rs:404-407); here they raise
MatchError like the try_* forms.

Haystacks may be bytes/bytearray/str/numpy uint8 arrays (host; uploaded for the call) or a torch
uint8 CUDA tensor (device resident, zero-copy).  Every search runs on the GPU.
"""
import ctypes as C
import enum

import numpy as np

from . import _lib

# numpy view of acgpu_match / Match{pattern, span} (src/util/search.rs:825-830)
MATCH_DTYPE = np.dtype([("pattern", "<u4"), ("_pad", "<u4"), ("start", "<u8"), ("end", "<u8")])


class MatchKind(enum.IntEnum):  # src/util/search.rs:1052-1074
    Standard = 0
    LeftmostFirst = 1
    LeftmostLongest = 2


class StartKind(enum.IntEnum):  # src/util/search.rs:1133-1142
    Both = 0
    Unanchored = 1
    Anchored = 2


class AhoCorasickKind(enum.IntEnum):  # src/ahocorasick.rs:2627-2634
    NoncontiguousNFA = 1
    ContiguousNFA = 2
    DFA = 3


class Anchored(enum.IntEnum):  # src/util/search.rs:784-792
    No = 0
    Yes = 1


_BUILD_ERRS = {1: "StateIDOverflow", 2: "PatternIDOverflow", 3: "PatternTooLong"}
_MATCH_ERRS = {10: "InvalidInputAnchored", 11: "InvalidInputUnanchored", 12: "UnsupportedStream",
               13: "UnsupportedOverlapping", 14: "UnsupportedEmpty"}


class BuildError(Exception):  # src/util/error.rs:16-37
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code
        self.kind = _BUILD_ERRS.get(code, "Other")


class MatchError(Exception):  # src/util/error.rs:170-204
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code
        self.kind = _MATCH_ERRS.get(code, "Other")


def _raise(code, build=False):
    L = _lib.load_library()
    msg = L.acgpu_status_str(code).decode()
    if code in (40, 41, 30):
        detail = L.acgpu_last_error().decode()
        raise RuntimeError(f"acgpu: {msg}: {detail} (no CPU fallback exists)")
    if code == 20:
        raise ValueError(msg)  # the reference panics: Input::set_span assertion, src/util/search.rs:332-342
    if build or code in _BUILD_ERRS:
        raise BuildError(code, msg)
    if code in _MATCH_ERRS:
        raise MatchError(code, msg)
    detail = L.acgpu_last_error().decode()   # (INVALID_ARGUMENT and the like: the library says which argument)
    raise RuntimeError(f"acgpu: {msg} ({code})" + (f": {detail}" if detail else ""))


class Match:
    """Match{pattern, span}, src/util/search.rs:825-930."""
    __slots__ = ("_pid", "_start", "_end")

    def __init__(self, pattern, start, end):
        assert start <= end, "invalid match span"  # Match::new, src/util/search.rs:857-861
        self._pid, self._start, self._end = int(pattern), int(start), int(end)

    @classmethod
    def must(cls, pattern, rng):
        return cls(pattern, rng[0], rng[1])

    def pattern(self):
        return self._pid

    def start(self):
        return self._start

    def end(self):
        return self._end

    def range(self):
        return range(self._start, self._end)

    def span(self):
        return (self._start, self._end)

    def is_empty(self):
        return self._start == self._end

    def len(self):
        return self._end - self._start

    __len__ = len

    def as_tuple(self):
        return (self._pid, self._start, self._end)

    def __eq__(self, o):
        if isinstance(o, Match):
            return self.as_tuple() == o.as_tuple()
        if isinstance(o, tuple):
            return self.as_tuple() == o
        return NotImplemented

    def __hash__(self):
        return hash(self.as_tuple())

    def __repr__(self):
        return f"Match(pattern={self._pid}, span={self._start}..{self._end})"


class Input:
    """Input, src/util/search.rs:83-640 (builder-style setters return self)."""

    def __init__(self, haystack):
        self._hay = haystack
        self._n = _hay_len(haystack)
        self._start, self._end = 0, self._n
        self._anchored = Anchored.No
        self._earliest = False

    def span(self, span):
        return self.range(span[0], span[1])

    def range(self, start, end=None):
        if end is None and not isinstance(start, int):
            start, end = start.start, start.stop
        # set_span assertion, src/util/search.rs:332-342
        if not (end <= self._n and start <= end + 1):
            raise ValueError(f"invalid span {start}..{end} for haystack of length {self._n}")
        self._start, self._end = start, end
        return self

    def anchored(self, mode):
        self._anchored = Anchored(int(mode))
        return self

    def earliest(self, yes):
        self._earliest = bool(yes)
        return self

    def haystack(self):
        return self._hay

    def start(self):
        return self._start

    def end(self):
        return self._end

    def get_span(self):
        return (self._start, self._end)

    def get_anchored(self):
        return self._anchored

    def get_earliest(self):
        return self._earliest

    def is_done(self):  # src/util/search.rs:627-629
        return self._start > self._end


def _is_torch(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _hay_len(h):
    if _is_torch(h):
        return int(h.numel())
    if isinstance(h, np.ndarray):
        return int(h.size)
    if isinstance(h, str):
        return len(h.encode())
    return len(h)


class _HayRef:
    """Pointer + length + keep-alive for one call."""

    def __init__(self, h):
        self.on_device = 0
        if _is_torch(h):
            import torch
            if h.dtype != torch.uint8 or not h.is_contiguous():
                raise TypeError("device haystack must be a contiguous torch.uint8 tensor")
            self.keep = h
            self.ptr = h.data_ptr()
            self.n = int(h.numel())
            self.on_device = 1 if h.is_cuda else 0
            self.device = h.device.index if h.is_cuda else None
            return
        self.device = None
        if isinstance(h, str):
            h = h.encode()
        if isinstance(h, np.ndarray):
            if h.dtype != np.uint8 or not h.flags["C_CONTIGUOUS"]:
                raise TypeError("numpy haystack must be contiguous uint8")
            self.keep = h
            self.ptr = h.ctypes.data
            self.n = int(h.size)
            return
        b = bytes(h) if not isinstance(h, bytes) else h
        self.keep = b
        self.ptr = C.cast(C.c_char_p(b), C.c_void_p).value or 0
        self.n = len(b)


def _device_call_stream(ref, stream):
    """The C ABI launches on the CURRENT HIP device (tables are replicated per device, acgpu_upload) and on the stream it
    is given: a tensor that lives on another device is refused, and a device tensor is searched on torch's current stream
    unless the caller names one -- ordered behind the kernel that produced it."""
    if not ref.on_device:
        return stream
    import torch
    cur = torch.cuda.current_device()
    if ref.device is not None and ref.device != cur:
        raise ValueError(f"haystack tensor lives on cuda:{ref.device} but the current device is cuda:{cur}; "
                         f"wrap the call in torch.cuda.device({ref.device})")
    if stream is None:
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # (the stream handle without a torch.cuda.Stream object: 3 us per call)
        stream = (raw(cur) if raw is not None else torch.cuda.current_stream().cuda_stream) or None
    return stream


def _as_input(x):
    return x if isinstance(x, Input) else Input(x)


class AhoCorasickBuilder:
    """AhoCorasickBuilder, src/ahocorasick.rs:2135-2616. Setters return self for chaining."""

    def __init__(self):
        self._match_kind = MatchKind.Standard
        self._start_kind = StartKind.Unanchored
        self._kind = None
        self._casei = False
        self._prefilter = True
        self._dense_depth = None
        self._byte_classes = True
        # GPU-side knobs (no reference counterpart)
        self._chunk_bytes = 0
        self._engine = 0
        self._gpu_dfa_fill = False
        self._deterministic_routing = False
        self._variants = {}

    def match_kind(self, kind):
        self._match_kind = MatchKind(int(kind))
        return self

    def start_kind(self, kind):
        self._start_kind = StartKind(int(kind))
        return self

    def ascii_case_insensitive(self, yes):
        self._casei = bool(yes)
        return self

    def kind(self, kind):
        self._kind = None if kind is None else AhoCorasickKind(int(kind))
        return self

    def prefilter(self, yes):
        self._prefilter = bool(yes)
        return self

    def dense_depth(self, depth):
        self._dense_depth = int(depth)
        return self

    def byte_classes(self, yes):
        self._byte_classes = bool(yes)
        return self

    def gpu_chunk_bytes(self, n):
        """Bytes of haystack per wavefront lane (multiple of 64; 0 = default)."""
        self._chunk_bytes = int(n)
        return self

    def gpu_engine(self, name):
        """acgpu_engine by name: 'auto' | 'walk' (the transition walk of the automaton's tables: DFA or contiguous NFA) |
        'hot' (DFA walk with the whole automaton in LDS) | 'pf' (prefix filter)."""
        self._engine = {"auto": 0, "walk": 1, "cnfa_walk": 2, "hot": 3, "pf": 4}[name]
        return self

    def gpu_variant(self, name, value):
        """An engine variant of the automaton being built (acgpu_set_variant; names in include/acgpu.h): the explicit,
        per-automaton way to select a form of a device engine -- for tests, the fuzzer and A/B runs.  Identical results."""
        self._variants[str(name)] = int(value)
        return self

    def gpu_deterministic_routing(self, yes):
        """No adaptive hints between the searches of this automaton: every call's engine choice follows from the
        automaton and the span alone (acgpu_config.deterministic_routing).  Results are identical either way."""
        self._deterministic_routing = bool(yes)
        return self

    def gpu_dfa_fill(self, yes):
        """Compute the DFA transition rows on the device (one launch per trie depth); same table, faster builds of
        large full DFAs.  Needs a HIP device at build time."""
        self._gpu_dfa_fill = bool(yes)
        return self

    def build(self, patterns):
        L = _lib.load_library()
        cfg = _lib.Config()
        L.acgpu_config_init(C.byref(cfg))
        cfg.match_kind = int(self._match_kind)
        cfg.start_kind = int(self._start_kind)
        cfg.kind = 0 if self._kind is None else int(self._kind)
        cfg.ascii_case_insensitive = int(self._casei)
        cfg.byte_classes = int(self._byte_classes)
        cfg.prefilter = int(self._prefilter)
        if self._dense_depth is not None:
            cfg.dense_depth_set = 1
            cfg.dense_depth = min(self._dense_depth, 0xFFFFFFFF)
        cfg.chunk_bytes = self._chunk_bytes
        cfg.engine = self._engine
        cfg.gpu_dfa_fill = int(self._gpu_dfa_fill)
        cfg.deterministic_routing = int(self._deterministic_routing)
        pats = [p.encode() if isinstance(p, str) else bytes(p) for p in patterns]
        n = len(pats)
        arr = (C.c_char_p * max(n, 1))(*pats)
        lens = (C.c_size_t * max(n, 1))(*[len(p) for p in pats])
        h = C.c_void_p()
        rc = L.acgpu_build(C.byref(cfg), arr, lens, n, C.byref(h))
        if rc:
            _raise(rc, build=True)
        a = AhoCorasick(_handle=h)
        for name, value in self._variants.items():
            a.set_variant(name, value)
        return a


class AhoCorasick:
    """AhoCorasick, src/ahocorasick.rs:177-2027 (search surface)."""

    def __init__(self, patterns=None, _handle=None):
        self._L = _lib.load_library()
        if _handle is None:
            _handle = AhoCorasickBuilder().build(patterns or [])._steal()
        self._h = _handle

    def _steal(self):
        h, self._h = self._h, None
        return h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.acgpu_free(self._h)
            self._h = None

    @classmethod
    def new(cls, patterns):
        return AhoCorasickBuilder().build(patterns)

    @staticmethod
    def builder():
        return AhoCorasickBuilder()

    def set_variant(self, name, value):
        """acgpu_set_variant: before the automaton's first upload / search."""
        rc = self._L.acgpu_set_variant(self._h, str(name).encode(), int(value))
        if rc:
            _raise(rc)
        return self

    # ---- getters
    def kind(self):
        return AhoCorasickKind(self._L.acgpu_kind_of(self._h))

    def start_kind(self):
        return StartKind(self._L.acgpu_start_kind_of(self._h))

    def match_kind(self):
        return MatchKind(self._L.acgpu_match_kind_of(self._h))

    def min_pattern_len(self):
        return self._L.acgpu_min_pattern_len(self._h)

    def max_pattern_len(self):
        return self._L.acgpu_max_pattern_len(self._h)

    def patterns_len(self):
        return self._L.acgpu_patterns_len(self._h)

    def memory_usage(self):
        return self._L.acgpu_memory_usage(self._h)

    def tables(self):
        t = _lib.CTables()
        self._L.acgpu_get_tables(self._h, C.byref(t))
        return t

    def upload(self, device=0):
        rc = self._L.acgpu_upload(self._h, int(device))
        if rc:
            _raise(rc)

    # ---- search plumbing
    def _cinput(self, inp, out_on_device=False, stream=None):
        ref = _HayRef(inp.haystack())
        stream = _device_call_stream(ref, stream)
        ci = _lib.CInput(ref.ptr, ref.n, inp.start(), inp.end(), int(inp.get_anchored()), int(inp.get_earliest()),
                         ref.on_device, int(out_on_device), stream)
        return ci, ref

    def _collect(self, fn, inp, prof=None, extra=()):
        ci, ref = self._cinput(inp)
        nout = C.c_size_t()
        # a too-small buffer costs a second full search (the C ABI reports the required size, it keeps no results):
        # size the first attempt generously for large haystacks (np.empty does not touch the pages)
        cap = max(4096, (inp.end() - inp.start()) // 2048)
        while True:
            buf = np.empty(cap, dtype=MATCH_DTYPE)
            args = [self._h, C.byref(ci), *extra, C.c_void_p(buf.ctypes.data), cap, C.byref(nout)]
            if prof is not None:
                args.append(C.byref(prof))
            rc = fn(*args)
            if rc == 21:  # ACGPU_ERR_BUFFER_TOO_SMALL: *n_out is the required capacity
                cap = nout.value
                continue
            if rc:
                _raise(rc)
            return buf[:nout.value]

    @staticmethod
    def _wrap(arr, as_numpy):
        if as_numpy:
            return arr
        return iter([Match(int(p), int(s), int(e)) for p, s, e in zip(arr["pattern"], arr["start"], arr["end"])])

    # ---- AhoCorasick::try_find / find / is_match
    def try_find(self, input):
        inp = _as_input(input)
        ci, ref = self._cinput(inp)
        found = C.c_int32()
        m = _lib.CMatch()
        rc = self._L.acgpu_find(self._h, C.byref(ci), C.byref(found), C.byref(m))
        if rc:
            _raise(rc)
        return Match(m.pattern, m.start, m.end) if found.value else None

    find = try_find

    def is_match(self, input):
        inp = _as_input(input)
        ci, ref = self._cinput(inp)
        r = C.c_int32()
        rc = self._L.acgpu_is_match(self._h, C.byref(ci), C.byref(r))
        if rc:
            _raise(rc)
        return bool(r.value)

    # ---- iterators
    def try_find_iter(self, input, as_numpy=False, profile=None):
        fn = self._L.acgpu_find_iter if profile is None else self._L.acgpu_find_iter_ex
        return self._wrap(self._collect(fn, _as_input(input), profile), as_numpy)

    find_iter = try_find_iter

    def try_find_overlapping_iter(self, input, as_numpy=False, profile=None):
        fn = self._L.acgpu_find_overlapping if profile is None else self._L.acgpu_find_overlapping_ex
        return self._wrap(self._collect(fn, _as_input(input), profile), as_numpy)

    find_overlapping_iter = try_find_overlapping_iter

    # ---- replace_all family, src/ahocorasick.rs:651-844, :1396-1560 -> src/automaton.rs:433-550
    def _replace(self, haystack, replace_with, flags, stream=None):
        repl = [r.encode() if isinstance(r, str) else bytes(r) for r in replace_with]
        if len(repl) != self.patterns_len():  # the reference asserts (src/automaton.rs:442-447)
            raise ValueError("replace_all requires a replacement for every pattern in the automaton")
        ref = _HayRef(haystack)
        on_dev = bool(ref.on_device)
        stream = _device_call_stream(ref, stream)
        ci = _lib.CInput(ref.ptr, ref.n, 0, ref.n, 0, 0, ref.on_device, int(on_dev), stream)
        n = len(repl)
        arr = (C.c_char_p * max(n, 1))(*repl)
        lens = (C.c_size_t * max(n, 1))(*[len(r) for r in repl])
        need = C.c_size_t()
        cap = ref.n + 64
        while True:
            if on_dev:
                import torch
                out = torch.empty(max(cap, 16), dtype=torch.uint8, device=ref.keep.device)
                optr = out.data_ptr()
            else:
                out = np.empty(max(cap, 16), dtype=np.uint8)
                optr = out.ctypes.data
            rc = self._L.acgpu_replace_all(self._h, C.byref(ci), arr, lens, n, flags, C.c_void_p(optr), cap,
                                           C.byref(need))
            if rc == 21:  # ACGPU_ERR_BUFFER_TOO_SMALL: *out_len is the required size
                cap = need.value
                continue
            if rc:
                _raise(rc)
            return out[:need.value]

    def try_replace_all_bytes(self, haystack, replace_with, stream=None):
        """replace_all_bytes: bytes-like / numpy in -> bytes out; torch CUDA uint8 tensor in -> CUDA tensor out."""
        out = self._replace(haystack, replace_with, 0, stream)
        return out if _is_torch(out) else out.tobytes()

    replace_all_bytes = try_replace_all_bytes

    def try_replace_all(self, haystack, replace_with):
        """replace_all on &str: matches that split a UTF-8 code point are skipped (src/automaton.rs:505-513)."""
        if not isinstance(haystack, str):
            raise TypeError("replace_all takes a str haystack; use replace_all_bytes for bytes")
        return self._replace(haystack.encode(), replace_with, 1).tobytes().decode()

    replace_all = try_replace_all

    def try_replace_all_with_bytes(self, haystack, dst, replace_with):
        """Closure form (src/automaton.rs:530-550): replace_with(match, matched_bytes, dst) -> bool appends to the
        bytearray `dst`; returning False stops after that match.  The matches come from the device find_iter, the
        closure runs on the host."""
        hay = bytes(haystack)
        last = 0
        for m in self.try_find_iter(hay):
            dst += hay[last:m.start()]
            last = m.end()
            if not replace_with(m, hay[m.start():m.end()], dst):
                break
        dst += hay[last:]

    replace_all_with_bytes = try_replace_all_with_bytes

    def try_replace_all_with(self, haystack, dst, replace_with):
        """Closure form on str (src/automaton.rs:493-522); `dst` is a list of str pieces that the closure appends to."""
        hay = haystack.encode()

        def boundary(i):
            return i == 0 or i >= len(hay) or (hay[i] & 0xC0) != 0x80

        last = 0
        for m in self.try_find_iter(hay):
            if not (boundary(m.start()) and boundary(m.end())):
                continue
            dst.append(hay[last:m.start()].decode())
            last = m.end()
            if not replace_with(m, hay[m.start():m.end()].decode(), dst):
                break
        dst.append(hay[last:].decode())

    replace_all_with = try_replace_all_with

    # ---- stream search, src/ahocorasick.rs:906-912, :1677-1683, :1751-1765, :1829-1843 -> src/automaton.rs:1036-1244
    def _stream_chunks(self, rdr, chunk_bytes):
        """Yields (chunk bytes, numpy matches with absolute offsets) per refill of `rdr` (.read(n) -> bytes)."""
        h = C.c_void_p()
        rc = self._L.acgpu_stream_begin(self._h, C.byref(h))
        if rc:
            _raise(rc)
        try:
            n = C.c_size_t()
            while True:
                chunk = rdr.read(chunk_bytes)
                if not chunk:
                    return
                chunk = bytes(chunk)
                rc = self._L.acgpu_stream_feed(h, C.cast(C.c_char_p(chunk), C.c_void_p), len(chunk), 0, None, C.byref(n))
                if rc:
                    _raise(rc)
                arr = np.empty(n.value, dtype=MATCH_DTYPE)
                rc = self._L.acgpu_stream_matches(h, C.c_void_p(arr.ctypes.data), n.value, C.byref(n))
                if rc:
                    _raise(rc)
                yield chunk, arr
        finally:
            self._L.acgpu_stream_end(h)

    def try_stream_find_iter(self, rdr, chunk_bytes=64 << 20):
        """StreamFindIter: matches of a file-like object, offsets relative to the start of the stream."""
        for _, arr in self._stream_chunks(rdr, chunk_bytes):
            for p, s, e in zip(arr["pattern"].tolist(), arr["start"].tolist(), arr["end"].tolist()):
                yield Match(p, s, e)

    stream_find_iter = try_stream_find_iter

    def try_stream_replace_all_with(self, rdr, wtr, replace_with, chunk_bytes=64 << 20):
        """src/automaton.rs try_stream_replace_all_with: non-match bytes are copied to `wtr`, every match is handed
        to replace_with(match, matched_bytes, wtr)."""
        keep = max(self.max_pattern_len() - 1, 0) if self.patterns_len() else 0
        pend, pend_abs, reported, total = b"", 0, 0, 0   # unreported tail of the stream, its offset, bytes written
        for chunk, arr in self._stream_chunks(rdr, chunk_bytes):
            pend += chunk
            total += len(chunk)
            for p, s, e in zip(arr["pattern"].tolist(), arr["start"].tolist(), arr["end"].tolist()):
                if s > reported:
                    wtr.write(pend[reported - pend_abs:s - pend_abs])
                replace_with(Match(p, s, e), pend[s - pend_abs:e - pend_abs], wtr)
                reported = e
            safe = max(reported, total - keep)   # a later match cannot start before this
            if safe > reported:
                wtr.write(pend[reported - pend_abs:safe - pend_abs])
                reported = safe
            pend, pend_abs = pend[reported - pend_abs:], reported
        if pend:
            wtr.write(pend)

    stream_replace_all_with = try_stream_replace_all_with

    def try_stream_replace_all(self, rdr, wtr, replace_with, chunk_bytes=64 << 20):
        repl = [r.encode() if isinstance(r, str) else bytes(r) for r in replace_with]
        if len(repl) != self.patterns_len():
            raise ValueError("stream_replace_all requires a replacement for every pattern in the automaton")
        self.try_stream_replace_all_with(rdr, wtr, lambda m, _b, w: w.write(repl[m.pattern()]), chunk_bytes)

    stream_replace_all = try_stream_replace_all

    def find_overlapping_shard(self, input, shard_begin, shard_end, as_numpy=True, profile=None):
        """Matches of the overlapping search whose end lies in (shard_begin, shard_end] (see acgpu.h)."""
        prof = profile if profile is not None else _lib.CProfile()
        arr = self._collect(self._L.acgpu_find_overlapping_shard, _as_input(input), prof,
                            extra=(C.c_size_t(shard_begin), C.c_size_t(shard_end)))
        return self._wrap(arr, as_numpy)

    def overlapping_device(self, hay_tensor, span=None, shard=None, out=None, profile=None, stream=None):
        """Device-to-device form: `hay_tensor` and `out` are torch CUDA tensors; returns the number of matches.

        Returns (n, ok): n = number of matches, ok = False when `out` was too small to hold them (nothing usable was
        written).  `out` must be a uint8 tensor of >= n*24 bytes, or None to only count."""
        import torch
        inp = Input(hay_tensor)
        if span is not None:
            inp.range(span[0], span[1])
        ci, ref = self._cinput(inp, out_on_device=True, stream=stream)
        sb, se = (inp.start(), inp.end()) if shard is None else shard
        nout = C.c_size_t()
        prof = profile if profile is not None else _lib.CProfile()
        cap = 0 if out is None else out.numel() // MATCH_DTYPE.itemsize
        optr = C.c_void_p(0 if out is None else out.data_ptr())
        rc = self._L.acgpu_find_overlapping_shard(self._h, C.byref(ci), sb, se, optr, cap, C.byref(nout),
                                                  C.byref(prof))
        if rc == 21:
            return nout.value, False
        if rc:
            _raise(rc)
        return nout.value, True


    def find_iter_device(self, hay_tensor, out, span=None, profile=None, stream=None):
        """find_iter with device-resident haystack AND output: `out` is a uint8 CUDA tensor of >= n*24 bytes receiving the
        non-overlapping matches in order.  Returns (n, ok); ok = False when `out` was too small (n = required records)."""
        inp = Input(hay_tensor)
        if span is not None:
            inp.range(span[0], span[1])
        ci, ref = self._cinput(inp, out_on_device=True, stream=stream)
        nout = C.c_size_t()
        prof = profile if profile is not None else _lib.CProfile()
        cap = out.numel() // MATCH_DTYPE.itemsize
        rc = self._L.acgpu_find_iter_ex(self._h, C.byref(ci), C.c_void_p(out.data_ptr()), cap, C.byref(nout), C.byref(prof))
        if rc == 21:
            return nout.value, False
        if rc:
            _raise(rc)
        return nout.value, True

    def find_overlapping_multi(self, shard_tensors, out, halo_included=True, dst_device=None):
        """acgpu_find_overlapping_multi: `shard_tensors` = the haystack cut into consecutive pieces, one uint8 CUDA tensor
        per shard (any devices, several on one device allowed), each -- except the first -- beginning with the
        max_pattern_len-1 bytes that precede it in the haystack (halo_included).  `out`: uint8 CUDA tensor on the
        destination device receiving the gathered records in haystack order.  Returns (n, per-shard counts)."""
        halo = self.max_pattern_len() - 1 if halo_included else 0
        n = len(shard_tensors)
        arr = (_lib.CShard * n)()
        off = 0
        for i, t in enumerate(shard_tensors):
            left = halo if i else 0
            if not (t.is_cuda and t.is_contiguous()):
                raise ValueError(f"shard {i}: a contiguous CUDA tensor is required")
            if t.numel() < left:
                raise ValueError(f"shard {i}: {t.numel()} bytes, shorter than its halo of {left}")
            if off < left:   # (the first shard is shorter than a halo: the second one's halo would reach in front of byte 0)
                raise ValueError(f"shard {i}: the shards in front of it hold {off} bytes, fewer than its halo of {left}")
            arr[i] = _lib.CShard(t.device.index, 0, t.data_ptr(), t.numel(), 0, t.numel(), left, t.numel(), off - left)
            off += t.numel() - left
        dst = out.device.index if dst_device is None else dst_device
        nout = C.c_size_t()
        counts = (C.c_uint64 * n)()
        rc = self._L.acgpu_find_overlapping_multi(self._h, arr, n, dst, C.c_void_p(out.data_ptr()), out.numel() // MATCH_DTYPE.itemsize,
                                                  C.byref(nout), counts)
        if rc == 21:
            raise ValueError(f"`out` holds {out.numel() // MATCH_DTYPE.itemsize} records, {nout.value} needed")
        if rc:
            _raise(rc)
        return nout.value, list(counts)

    ENQUEUE_MAX_EVENTS = 16384   # ACGPU_ENQUEUE_MAX_EVENTS

    def overlapping_enqueue(self, hay_tensor, out, totals, span=None, shard=None, slot=-1, stream=None, classic=False):
        """Enqueue-only form (acgpu_find_overlapping_enqueue): returns as soon as the kernels are queued on `stream`.
        `out`: uint8 CUDA tensor for the records, `totals`: int64/uint64 CUDA tensor of >= 2 elements receiving
        [records, occurrence events] in stream order.  The records are valid iff totals[1] <= ENQUEUE_MAX_EVENTS and
        totals[0] * 24 <= out.numel(); otherwise repeat the search with overlapping_device().  classic=True (dense
        results expected) and automata of the walk engines run chunk counters -> scan -> fill instead: no occurrence
        limit (totals[1] = 0)."""
        inp = Input(hay_tensor)
        if span is not None:
            inp.range(span[0], span[1])
        ci, ref = self._cinput(inp, out_on_device=True, stream=stream)
        sb, se = (inp.start(), inp.end()) if shard is None else shard
        cap = out.numel() // MATCH_DTYPE.itemsize
        if not (totals.is_cuda and totals.element_size() == 8 and totals.numel() >= 2):
            raise ValueError("totals: a CUDA tensor of two 64-bit elements is required")
        rc = self._L.acgpu_find_overlapping_enqueue_ex(self._h, C.byref(ci), sb, se, C.c_void_p(out.data_ptr()), cap,
                                                       C.c_void_p(totals.data_ptr()), int(slot), 1 if classic else 0)
        if rc:
            _raise(rc)

    def enqueue_kernel_ms(self, slot, stream=None):
        """Duration of the scan kernel of the enqueue-only call that used `slot` (after the stream was synchronised)."""
        ms = C.c_float()
        rc = self._L.acgpu_enqueue_kernel_ms(self._h, C.c_void_p(stream or 0), int(slot), C.byref(ms))
        if rc:
            _raise(rc)
        return float(ms.value)


def stream_read_gbps(tensor, iters=5, stream=None):
    """GB/s of a plain read-only streaming kernel over `tensor` (device uint8): the empirical ceiling of this box."""
    L = _lib.load_library()
    ms = C.c_float()
    rc = L.acgpu_stream_read(tensor.data_ptr(), tensor.numel(), int(iters), C.byref(ms), stream)
    if rc:
        _raise(rc)
    return tensor.numel() / (ms.value * 1e-3) / 1e9


def gen_haystack(tensor, offset=0, seed=0xAC02, lo=0x20, span=95, stream=None):
    """Fill a torch uint8 CUDA tensor with the synthetic haystack of SURVEY.md Appendix C."""
    L = _lib.load_library()
    rc = L.acgpu_gen_haystack(tensor.data_ptr(), offset, tensor.numel(), seed, lo, span, stream)
    if rc:
        _raise(rc)
    return tensor
