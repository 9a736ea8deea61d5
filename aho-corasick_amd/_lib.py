"""Loading (and building) lib/libacgpu.so -- the C ABI declared in include/acgpu.h."""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# ACGPU_LIB: kernel-experiment builds of the same library (scripts/pf_variants.sh) or its bounds-checked debug flavour
# (lib/libacgpu_guard.so, tests/test_gpu_guard.py); never a different backend
_LIB = os.environ.get("ACGPU_LIB") or os.path.join(_PKG, "lib", "libacgpu.so")


def library_path():
    return _LIB


def build_library(force=False, quiet=True):
    """Compile every HIP/C++ source for gfx950 into lib/libacgpu.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
    return _LIB


class Config(C.Structure):
    _fields_ = [("match_kind", C.c_int32), ("start_kind", C.c_int32), ("kind", C.c_int32),
                ("ascii_case_insensitive", C.c_int32), ("byte_classes", C.c_int32), ("prefilter", C.c_int32),
                ("dense_depth_set", C.c_int32), ("dense_depth", C.c_uint32), ("chunk_bytes", C.c_uint32),
                ("engine", C.c_int32), ("gpu_dfa_fill", C.c_int32), ("deterministic_routing", C.c_int32),
                ("reserved", C.c_uint32 * 4)]


class CMatch(C.Structure):
    _fields_ = [("pattern", C.c_uint32), ("_pad", C.c_uint32), ("start", C.c_uint64), ("end", C.c_uint64)]


class CInput(C.Structure):
    _fields_ = [("haystack", C.c_void_p), ("haystack_len", C.c_size_t), ("span_start", C.c_size_t),
                ("span_end", C.c_size_t), ("anchored", C.c_int32), ("earliest", C.c_int32),
                ("haystack_on_device", C.c_int32), ("out_on_device", C.c_int32), ("stream", C.c_void_p)]


class CProfile(C.Structure):
    _fields_ = [("ms_scan", C.c_float), ("ms_compact", C.c_float), ("ms_fill", C.c_float), ("ms_total", C.c_float),
                ("bytes_scanned", C.c_uint64), ("n_chunks", C.c_uint64), ("n_active_chunks", C.c_uint64),
                ("n_matches", C.c_uint64), ("engine_used", C.c_uint32), ("routed", C.c_uint32)]


class CTables(C.Structure):
    _fields_ = [
        ("nnfa_states", C.c_size_t),
        ("nnfa_max_match_id", C.c_uint32), ("nnfa_start_unanchored_id", C.c_uint32),
        ("nnfa_start_anchored_id", C.c_uint32),
        ("byte_classes", C.c_uint8 * 256), ("alphabet_len", C.c_size_t),
        ("nnfa_fail", C.POINTER(C.c_uint32)), ("nnfa_depth", C.POINTER(C.c_uint32)),
        ("nnfa_match_off", C.POINTER(C.c_uint32)), ("nnfa_match_pid", C.POINTER(C.c_uint32)),
        ("dfa_trans", C.POINTER(C.c_uint32)), ("dfa_trans_len", C.c_size_t), ("dfa_state_len", C.c_size_t),
        ("dfa_stride2", C.c_size_t),
        ("dfa_max_match_id", C.c_uint32), ("dfa_start_unanchored_id", C.c_uint32),
        ("dfa_start_anchored_id", C.c_uint32),
        ("dfa_match_off", C.POINTER(C.c_uint32)), ("dfa_match_pid", C.POINTER(C.c_uint32)),
        ("dfa_num_match_states", C.c_size_t),
        ("cnfa_repr", C.POINTER(C.c_uint32)), ("cnfa_repr_len", C.c_size_t),
        ("cnfa_max_match_id", C.c_uint32), ("cnfa_start_unanchored_id", C.c_uint32),
        ("cnfa_start_anchored_id", C.c_uint32),
        ("pattern_lens", C.POINTER(C.c_uint32)),
    ]


class CShard(C.Structure):
    _fields_ = [("device", C.c_int32), ("_pad", C.c_int32), ("haystack", C.c_void_p), ("haystack_len", C.c_size_t),
                ("span_start", C.c_size_t), ("span_end", C.c_size_t), ("shard_begin", C.c_size_t), ("shard_end", C.c_size_t),
                ("global_offset", C.c_uint64)]


ABI_VERSION = 3   # ACGPU_ABI_VERSION of include/acgpu.h

# every symbol include/acgpu.h declares (tests check that the library exports exactly these)
SYMBOLS = [
    "acgpu_abi_version", "acgpu_last_error", "acgpu_status_str", "acgpu_config_init", "acgpu_build", "acgpu_free", "acgpu_set_variant",
    "acgpu_kind_of", "acgpu_match_kind_of", "acgpu_start_kind_of", "acgpu_patterns_len", "acgpu_min_pattern_len",
    "acgpu_max_pattern_len", "acgpu_memory_usage", "acgpu_upload", "acgpu_find_overlapping",
    "acgpu_find_overlapping_ex", "acgpu_find_overlapping_shard", "acgpu_find_overlapping_enqueue", "acgpu_find_overlapping_enqueue_ex",
    "acgpu_enqueue_kernel_ms", "acgpu_find_iter", "acgpu_find_iter_ex",
    "acgpu_find", "acgpu_is_match", "acgpu_replace_all", "acgpu_stream_begin", "acgpu_stream_feed",
    "acgpu_stream_matches", "acgpu_stream_end", "acgpu_get_tables", "acgpu_gen_haystack", "acgpu_stream_read",
    "acgpu_find_overlapping_multi", "acgpu_multi_last_transport", "acgpu_multi_last_error",
    "acgpu_device_count", "acgpu_device_malloc", "acgpu_device_free", "acgpu_device_copy", "acgpu_guard_violations",
]

_lib = None


def load_library():
    """dlopen lib/libacgpu.so. Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise ImportError(f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    L = C.CDLL(_LIB)
    vp, sz = C.c_void_p, C.c_size_t
    L.acgpu_abi_version.restype = C.c_uint32
    if L.acgpu_abi_version() != ABI_VERSION:   # struct layouts / enum numberings of include/acgpu.h this binding mirrors
        raise ImportError(f"{_LIB} has ABI version {L.acgpu_abi_version()}, this binding was written against {ABI_VERSION}: rebuild "
                          "(python -c 'import __graft_entry__ as g; g.build()')")
    L.acgpu_last_error.restype = C.c_char_p
    L.acgpu_status_str.restype = C.c_char_p
    L.acgpu_status_str.argtypes = [C.c_int]
    L.acgpu_config_init.argtypes = [C.POINTER(Config)]
    L.acgpu_config_init.restype = None
    L.acgpu_build.argtypes = [C.POINTER(Config), C.POINTER(C.c_char_p), C.POINTER(sz), sz, C.POINTER(vp)]
    L.acgpu_free.argtypes = [vp]
    L.acgpu_set_variant.argtypes = [vp, C.c_char_p, C.c_int32]
    L.acgpu_free.restype = None
    for f in ("acgpu_kind_of", "acgpu_match_kind_of", "acgpu_start_kind_of"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_int32
    for f in ("acgpu_patterns_len", "acgpu_min_pattern_len", "acgpu_max_pattern_len", "acgpu_memory_usage"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = sz
    L.acgpu_upload.argtypes = [vp, C.c_int]
    L.acgpu_find_overlapping.argtypes = [vp, C.POINTER(CInput), vp, sz, C.POINTER(sz)]
    L.acgpu_find_overlapping_ex.argtypes = [vp, C.POINTER(CInput), vp, sz, C.POINTER(sz), C.POINTER(CProfile)]
    L.acgpu_find_overlapping_shard.argtypes = [vp, C.POINTER(CInput), sz, sz, vp, sz, C.POINTER(sz),
                                               C.POINTER(CProfile)]
    L.acgpu_find_overlapping_enqueue.argtypes = [vp, C.POINTER(CInput), sz, sz, vp, sz, vp, C.c_int32]
    L.acgpu_find_overlapping_enqueue_ex.argtypes = [vp, C.POINTER(CInput), sz, sz, vp, sz, vp, C.c_int32, C.c_uint32]
    L.acgpu_enqueue_kernel_ms.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_float)]
    L.acgpu_guard_violations.argtypes = []
    L.acgpu_guard_violations.restype = C.c_longlong
    L.acgpu_find_iter.argtypes = [vp, C.POINTER(CInput), vp, sz, C.POINTER(sz)]
    L.acgpu_find_iter_ex.argtypes = [vp, C.POINTER(CInput), vp, sz, C.POINTER(sz), C.POINTER(CProfile)]
    L.acgpu_find.argtypes = [vp, C.POINTER(CInput), C.POINTER(C.c_int32), C.POINTER(CMatch)]
    L.acgpu_is_match.argtypes = [vp, C.POINTER(CInput), C.POINTER(C.c_int32)]
    L.acgpu_replace_all.argtypes = [vp, C.POINTER(CInput), C.POINTER(C.c_char_p), C.POINTER(sz), sz, C.c_uint32, vp, sz,
                                    C.POINTER(sz)]
    L.acgpu_stream_begin.argtypes = [vp, C.POINTER(vp)]
    L.acgpu_stream_feed.argtypes = [vp, vp, sz, C.c_int32, vp, C.POINTER(sz)]
    L.acgpu_stream_matches.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.acgpu_stream_end.argtypes = [vp]
    L.acgpu_stream_end.restype = None
    L.acgpu_get_tables.argtypes = [vp, C.POINTER(CTables)]
    L.acgpu_get_tables.restype = None
    L.acgpu_gen_haystack.argtypes = [vp, C.c_uint64, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp]
    L.acgpu_stream_read.argtypes = [vp, sz, C.c_int32, C.POINTER(C.c_float), vp]
    L.acgpu_find_overlapping_multi.argtypes = [vp, C.POINTER(CShard), sz, C.c_int32, vp, sz, C.POINTER(sz), C.POINTER(C.c_uint64)]
    L.acgpu_multi_last_error.restype = C.c_char_p
    L.acgpu_device_count.argtypes = [C.POINTER(C.c_int32)]
    L.acgpu_device_malloc.argtypes = [C.c_int32, sz, C.POINTER(vp)]
    L.acgpu_device_free.argtypes = [C.c_int32, vp]
    L.acgpu_device_copy.argtypes = [C.c_int32, vp, vp, sz, C.c_int32]
    _lib = L
    return L


# every symbol include/acgpu_test.h declares
TEST_SYMBOLS = ["acgpu_test_select_host", "acgpu_test_lw_host", "acgpu_test_lw_records_host", "acgpu_test_lw_event_records_host", "acgpu_test_engine_plan", "acgpu_test_event_order_shift", "acgpu_test_pf_host", "acgpu_test_cnfa_host",
                "acgpu_test_cnfa_tri_host", "acgpu_test_dfa_tri_host"]
_hooks = None


def load_test_hooks():
    """The test hooks (include/acgpu_test.h): lib/libacgpu_testhooks.so, which links libacgpu.so -- or the library
    ACGPU_LIB names when that flavour carries them itself (the host-ASan build)."""
    global _hooks
    if _hooks is not None:
        return _hooks
    L = load_library()
    if not hasattr(L, "acgpu_test_select_host"):
        path = os.path.join(_PKG, "lib", "libacgpu_testhooks.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: make -C aho-corasick_amd/csrc testhooks")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, sz = C.c_void_p, C.c_size_t
    L.acgpu_test_select_host.argtypes = [vp, sz, C.c_int32, sz, sz, vp, sz, C.POINTER(sz)]
    for f in ("acgpu_test_lw_host", "acgpu_test_cnfa_host", "acgpu_test_cnfa_tri_host", "acgpu_test_dfa_tri_host"):
        getattr(L, f).argtypes = [vp, vp, sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acgpu_test_pf_host.argtypes = [vp, vp, sz, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acgpu_test_lw_records_host.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(C.c_int32)]
    L.acgpu_test_lw_event_records_host.argtypes = [vp, vp, sz, C.c_uint32, vp, sz, C.POINTER(sz), C.POINTER(C.c_int32)]
    _hooks = L
    return L
