"""ctypes mirror of include/sjgpu.h (the C-ABI of libsjgpu.so, HIP kernels for gfx950).

Host-side names follow the reference's plug-in interface for this path
(/root/reference/include/simdjson/implementation.h:97-128,
 /root/reference/include/simdjson/internal/dom_parser_implementation.h:80): `stage1`, `minify`,
`validate_utf8`, `set_capacity`, error codes as simdjson::error_code integers.

There is no CPU fallback: a missing library or a missing GPU raises.
"""
import ctypes
import os

import numpy as np

from . import _paths

# simdjson::error_code values on this path (include/simdjson/error.h:19-53)
SUCCESS, CAPACITY, MEMALLOC, UTF8_ERROR, EMPTY, UNESCAPED_CHARS, UNCLOSED_STRING, UNEXPECTED_ERROR = 0, 1, 2, 11, 13, 14, 15, 24
# simdjson::stage1_mode (internal/dom_parser_implementation.h:22-27)
REGULAR, STREAMING_PARTIAL, STREAMING_FINAL, JSON_SEQUENCE_PARTIAL, JSON_SEQUENCE_FINAL, COMMA_DELIMITED_PARTIAL, COMMA_DELIMITED_FINAL = range(7)
F_UNCLOSED_STRING, F_UNESCAPED_CTRL, F_UTF8_ERROR, F_IDX_OVERFLOW, F_INTERNAL, F_RANGE_CARRY = 1, 2, 4, 8, 16, 32

EXPORTS = [
    "sjgpu_device_count", "sjgpu_ctx_create", "sjgpu_ctx_destroy", "sjgpu_set_capacity", "sjgpu_capacity",
    "sjgpu_last_error", "sjgpu_stage1", "sjgpu_minify", "sjgpu_validate_utf8", "sjgpu_validate_utf8_pieces", "sjgpu_stage1_device",
    "sjgpu_minify_device", "sjgpu_validate_utf8_device", "sjgpu_result", "sjgpu_stage1_error_from_flags", "sjgpu_stage1_tokens_device", "sjgpu_depth_scan_tokens_device",
    "sjgpu_stage1_finish_host", "sjgpu_trim_partial_utf8", "sjgpu_profile_enable", "sjgpu_profile_read", "sjgpu_set_pipeline", "sjgpu_debug_trace_stage1",
    "sjgpu_clean_cut", "sjgpu_string_parity_device", "sjgpu_stage1_shard_device", "sjgpu_minify_shard_device",
    "sjgpu_stage1_range_device", "sjgpu_minify_range_device",
    "sjgpu_host_alloc", "sjgpu_host_free", "sjgpu_host_register", "sjgpu_host_unregister", "sjgpu_last_pipeline",
    "sjgpu_profile_kernel", "sjgpu_debug_trace_pipelined", "sjgpu_debug_string_path", "sjgpu_stage1_many", "sjgpu_stage1_finish_device",
    "sjgpu_depth_scan_device", "sjgpu_parse_strings_device", "sjgpu_stage2_device", "sjgpu_stage2_tokens_device", "sjgpu_parse", "sjgpu_pool_trim", "sjgpu_stream_register", "sjgpu_stream_unregister", "sjgpu_stream_unregister_len", "sjgpu_debug_stream_extent", "sjgpu_match_keys_device", "sjgpu_comm_unique_id", "sjgpu_comm_create", "sjgpu_comm_destroy", "sjgpu_comm_last_error", "sjgpu_comm_ranks", "sjgpu_comm_gather_indices", "sjgpu_mgpu_create", "sjgpu_mgpu_destroy", "sjgpu_mgpu_count", "sjgpu_mgpu_stage1", "sjgpu_mgpu_minify",
    "sjgpu_mgpu_validate_utf8",
]


class SjgpuError(RuntimeError):
    pass


class Doc(ctypes.Structure):
    """sjgpu_doc (include/sjgpu.h): one document of a sjgpu_stage1_many batch."""
    _fields_ = [("buf", ctypes.c_void_p), ("len", ctypes.c_size_t), ("idx_out", ctypes.c_void_p), ("idx_words", ctypes.c_size_t),
                ("n", ctypes.c_uint32), ("error", ctypes.c_int)]


class ScanResult(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("out_len", ctypes.c_uint64)]


_lib = None


def load_library():
    """dlopen libsjgpu.so (raises if it has not been built: the product path never degrades to CPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_paths.LIB_SJGPU):
        raise SjgpuError(f"{_paths.LIB_SJGPU} not built (run python -m simdjson_amd.build)")
    # PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1, and two HIP runtimes in
    # one process cannot both open the GPU ("No HIP GPUs are available" in whichever comes second).
    # Importing torch FIRST makes the loader resolve libsjgpu's DT_NEEDED libamdhip64.so.7 to the
    # already-loaded copy (same SONAME), so a process that uses torch for device memory / streams /
    # torch.distributed and libsjgpu for the kernels runs on ONE runtime.  Pure C/C++ users of
    # libsjgpu (the simdjson plug-in shim) simply get /opt/rocm's runtime.
    if os.environ.get("SJGPU_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(_paths.LIB_SJGPU)
    vp, sz, u32p = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)
    L.sjgpu_device_count.restype = ctypes.c_int
    L.sjgpu_ctx_create.restype = ctypes.c_int
    L.sjgpu_ctx_create.argtypes = [ctypes.c_int, sz, ctypes.POINTER(vp)]
    L.sjgpu_ctx_destroy.restype = None
    L.sjgpu_ctx_destroy.argtypes = [vp]
    L.sjgpu_set_capacity.restype = ctypes.c_int
    L.sjgpu_set_capacity.argtypes = [vp, sz]
    L.sjgpu_capacity.restype = sz
    L.sjgpu_capacity.argtypes = [vp]
    L.sjgpu_last_error.restype = ctypes.c_char_p
    L.sjgpu_last_error.argtypes = [vp]
    L.sjgpu_stage1.restype = ctypes.c_int
    L.sjgpu_stage1.argtypes = [vp, vp, sz, ctypes.c_int, vp, sz, u32p, u32p]
    L.sjgpu_minify.restype = ctypes.c_int
    L.sjgpu_minify.argtypes = [vp, vp, sz, vp, ctypes.POINTER(sz)]
    L.sjgpu_validate_utf8.restype = ctypes.c_int
    L.sjgpu_validate_utf8.argtypes = [vp, vp, sz, ctypes.POINTER(ctypes.c_int)]
    L.sjgpu_validate_utf8_pieces.restype = ctypes.c_int
    L.sjgpu_validate_utf8_pieces.argtypes = [vp, vp, sz, sz, ctypes.POINTER(ctypes.c_int)]
    L.sjgpu_stage1_device.restype = ctypes.c_int
    L.sjgpu_stage1_device.argtypes = [vp, vp, sz, vp, sz, vp]
    L.sjgpu_minify_device.restype = ctypes.c_int
    L.sjgpu_minify_device.argtypes = [vp, vp, sz, vp, vp]
    L.sjgpu_validate_utf8_device.restype = ctypes.c_int
    L.sjgpu_validate_utf8_device.argtypes = [vp, vp, sz, vp]
    L.sjgpu_stage1_tokens_device.restype = ctypes.c_int
    L.sjgpu_stage1_tokens_device.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp]
    L.sjgpu_depth_scan_tokens_device.restype = ctypes.c_int
    L.sjgpu_depth_scan_tokens_device.argtypes = [vp, vp, ctypes.c_uint32, vp, vp]
    L.sjgpu_result.restype = ctypes.c_int
    L.sjgpu_result.argtypes = [vp, vp, ctypes.POINTER(ScanResult)]
    L.sjgpu_stage1_error_from_flags.restype = ctypes.c_int
    L.sjgpu_stage1_error_from_flags.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    L.sjgpu_stage1_finish_host.restype = ctypes.c_int
    L.sjgpu_stage1_finish_host.argtypes = [vp, sz, ctypes.c_int, vp, ctypes.c_uint32, ctypes.c_uint32, u32p, u32p]
    L.sjgpu_debug_trace_stage1.restype = ctypes.c_int
    L.sjgpu_debug_trace_stage1.argtypes = [vp, vp, sz, vp, sz, vp, ctypes.c_uint32]
    L.sjgpu_debug_string_path.restype = ctypes.c_int
    L.sjgpu_debug_string_path.argtypes = [vp]
    L.sjgpu_debug_trace_pipelined.restype = ctypes.c_int
    L.sjgpu_debug_trace_pipelined.argtypes = [vp, vp, sz, vp, sz, vp, ctypes.c_uint32, u32p]
    L.sjgpu_stage1_many.restype = ctypes.c_int
    L.sjgpu_stage1_many.argtypes = [vp, ctypes.POINTER(Doc), sz]
    L.sjgpu_stage1_finish_device.restype = ctypes.c_int
    L.sjgpu_stage1_finish_device.argtypes = [vp, vp, sz, ctypes.c_int, vp, ctypes.c_uint32, ctypes.c_uint32, vp, u32p, u32p]
    L.sjgpu_depth_scan_device.restype = ctypes.c_int
    L.sjgpu_depth_scan_device.argtypes = [vp, vp, vp, ctypes.c_uint32, vp, vp]
    L.sjgpu_parse_strings_device.restype = ctypes.c_int
    L.sjgpu_parse_strings_device.argtypes = [vp, vp, sz, vp, ctypes.c_uint32, ctypes.c_int, vp, sz, vp, vp, ctypes.POINTER(ctypes.c_uint64), u32p, u32p]
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.sjgpu_stage2_device.restype = ctypes.c_int
    L.sjgpu_stage2_device.argtypes = [vp, vp, sz, vp, ctypes.c_uint32, ctypes.c_uint32, vp, sz, vp, sz, vp, u64p, u64p]
    L.sjgpu_stage2_tokens_device.restype = ctypes.c_int
    L.sjgpu_stage2_tokens_device.argtypes = [vp, vp, sz, vp, ctypes.c_uint32, vp, ctypes.c_uint32, vp, sz, vp, sz, vp, u64p, u64p]
    L.sjgpu_match_keys_device.restype = ctypes.c_int
    L.sjgpu_match_keys_device.argtypes = [vp, vp, sz, vp, ctypes.c_uint32, vp, vp, ctypes.c_uint32, vp, vp, u32p]
    L.sjgpu_parse.restype = ctypes.c_int
    L.sjgpu_parse.argtypes = [vp, vp, sz, ctypes.c_uint32, vp, sz, vp, sz, u64p, u64p]
    L.sjgpu_comm_unique_id.restype = ctypes.c_int
    L.sjgpu_comm_unique_id.argtypes = [vp, sz]
    L.sjgpu_comm_create.restype = ctypes.c_int
    L.sjgpu_comm_create.argtypes = [ctypes.c_int, ctypes.c_int, vp, sz, ctypes.c_int, ctypes.POINTER(vp)]
    L.sjgpu_comm_destroy.restype = None
    L.sjgpu_comm_destroy.argtypes = [vp]
    L.sjgpu_comm_last_error.restype = ctypes.c_char_p
    L.sjgpu_comm_last_error.argtypes = [vp]
    L.sjgpu_comm_ranks.restype = ctypes.c_int
    L.sjgpu_comm_ranks.argtypes = [vp]
    L.sjgpu_comm_gather_indices.restype = ctypes.c_int
    L.sjgpu_comm_gather_indices.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, vp, sz, u64p, u64p, vp]
    L.sjgpu_mgpu_create.restype = ctypes.c_int
    L.sjgpu_mgpu_create.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(vp)]
    L.sjgpu_mgpu_destroy.restype = None
    L.sjgpu_mgpu_destroy.argtypes = [vp]
    L.sjgpu_mgpu_count.restype = ctypes.c_int
    L.sjgpu_mgpu_count.argtypes = [vp]
    L.sjgpu_mgpu_stage1.restype = ctypes.c_int
    L.sjgpu_mgpu_stage1.argtypes = [vp, vp, sz, ctypes.c_int, vp, sz, u32p, u32p]
    L.sjgpu_mgpu_minify.restype = ctypes.c_int
    L.sjgpu_mgpu_minify.argtypes = [vp, vp, sz, vp, ctypes.POINTER(sz)]
    L.sjgpu_mgpu_validate_utf8.restype = ctypes.c_int
    L.sjgpu_mgpu_validate_utf8.argtypes = [vp, vp, sz, ctypes.POINTER(ctypes.c_int)]
    L.sjgpu_set_pipeline.restype = ctypes.c_int
    L.sjgpu_set_pipeline.argtypes = [vp, ctypes.c_int]
    L.sjgpu_profile_enable.restype = ctypes.c_int
    L.sjgpu_profile_enable.argtypes = [vp, ctypes.c_int]
    L.sjgpu_profile_read.restype = ctypes.c_int
    L.sjgpu_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), u32p]
    L.sjgpu_trim_partial_utf8.restype = sz
    L.sjgpu_trim_partial_utf8.argtypes = [vp, sz]
    L.sjgpu_clean_cut.restype = sz
    L.sjgpu_clean_cut.argtypes = [vp, sz, sz]
    L.sjgpu_string_parity_device.restype = ctypes.c_int
    L.sjgpu_string_parity_device.argtypes = [vp, vp, sz, vp]
    L.sjgpu_stage1_shard_device.restype = ctypes.c_int
    L.sjgpu_stage1_shard_device.argtypes = [vp, vp, sz, ctypes.c_int, vp, sz, vp]
    L.sjgpu_minify_shard_device.restype = ctypes.c_int
    L.sjgpu_minify_shard_device.argtypes = [vp, vp, sz, ctypes.c_int, vp, vp]
    L.sjgpu_stage1_range_device.restype = ctypes.c_int
    L.sjgpu_stage1_range_device.argtypes = [vp, vp, sz, sz, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, vp, sz, vp]
    L.sjgpu_minify_range_device.restype = ctypes.c_int
    L.sjgpu_minify_range_device.argtypes = [vp, vp, sz, sz, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, vp, vp]
    L.sjgpu_last_pipeline.restype = ctypes.c_int
    L.sjgpu_last_pipeline.argtypes = [vp]
    L.sjgpu_profile_kernel.restype = ctypes.c_char_p
    L.sjgpu_profile_kernel.argtypes = [vp]
    L.sjgpu_host_alloc.restype = vp
    L.sjgpu_host_alloc.argtypes = [sz]
    L.sjgpu_host_free.restype = None
    L.sjgpu_host_free.argtypes = [vp]
    L.sjgpu_host_register.restype = ctypes.c_int
    L.sjgpu_host_register.argtypes = [vp, sz]
    L.sjgpu_host_unregister.restype = ctypes.c_int
    L.sjgpu_host_unregister.argtypes = [vp]
    _lib = L
    return L


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(0, np.uint8)


class DomParserImplementation:
    """One GPU parser context == one simdjson dom_parser_implementation instance (stage-1 part).

    Mirrors the members the reference's callers touch: `n_structural_indexes`, `structural_indexes`,
    `capacity()` / `set_capacity()`, `stage1(buf, len, mode)`."""

    def __init__(self, capacity, device=0):
        self.L = load_library()
        h = ctypes.c_void_p()
        rc = self.L.sjgpu_ctx_create(int(device), int(capacity), ctypes.byref(h))
        if rc != 0:
            raise SjgpuError(f"sjgpu_ctx_create(device={device}, capacity={capacity}) failed with {rc} "
                             f"(-1 = no HIP device; this backend has no CPU fallback)")
        self.h = h
        self.device = device
        self.n_structural_indexes = 0
        self.next_structural_index = 0
        self.structural_indexes = np.zeros(((int(capacity) + 63) // 64) * 64 + 9, dtype=np.uint32)

    def close(self):
        if getattr(self, "h", None):
            self.L.sjgpu_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def capacity(self):
        return int(self.L.sjgpu_capacity(self.h))

    def set_capacity(self, capacity):
        rc = self.L.sjgpu_set_capacity(self.h, int(capacity))
        if rc == 0:
            self.structural_indexes = np.zeros(((int(capacity) + 63) // 64) * 64 + 9, dtype=np.uint32)
        return rc

    def last_error(self):
        return self.L.sjgpu_last_error(self.h).decode()

    # ---- host-buffer path (what dom::parser / ondemand::parser drive through the plug-in shim) ----
    def stage1(self, data, mode=REGULAR):
        a = _as_u8(data)
        n = ctypes.c_uint32(self.n_structural_indexes)
        nxt = ctypes.c_uint32(self.next_structural_index)
        rc = self.L.sjgpu_stage1(self.h, a.ctypes.data, len(a), int(mode), self.structural_indexes.ctypes.data,
                                 len(self.structural_indexes), ctypes.byref(n), ctypes.byref(nxt))
        self.n_structural_indexes = int(n.value)
        self.next_structural_index = int(nxt.value)
        if rc < 0:
            raise SjgpuError(f"sjgpu_stage1 infrastructure error {rc}: {self.last_error()}")
        return rc

    def minify(self, data):
        """-> (error_code, minified bytes).  The bytes are a view of a buffer owned by this object (like
        structural_indexes: allocated once, reused), valid until the next minify() call."""
        a = _as_u8(data)
        if getattr(self, "_minify_out", None) is None or len(self._minify_out) < max(len(a), 1):
            self._minify_out = np.zeros(max(len(a), 1), dtype=np.uint8)
        dst = self._minify_out
        n = ctypes.c_size_t(0)
        rc = self.L.sjgpu_minify(self.h, a.ctypes.data, len(a), dst.ctypes.data, ctypes.byref(n))
        if rc < 0:
            raise SjgpuError(f"sjgpu_minify infrastructure error {rc}: {self.last_error()}")
        return rc, dst[: n.value]

    def validate_utf8(self, data, piece_bytes=None):
        """piece_bytes: sjgpu_validate_utf8_pieces -- the same verdict with the input taken in pieces of that size"""
        a = _as_u8(data)
        ok = ctypes.c_int(0)
        if piece_bytes is None:
            rc = self.L.sjgpu_validate_utf8(self.h, a.ctypes.data, len(a), ctypes.byref(ok))
        else:
            rc = self.L.sjgpu_validate_utf8_pieces(self.h, a.ctypes.data, len(a), int(piece_bytes), ctypes.byref(ok))
        if rc != 0:
            raise SjgpuError(f"sjgpu_validate_utf8 error {rc}: {self.last_error()}")
        return bool(ok.value)

    # ---- device-resident path (pointers are raw device addresses, e.g. torch.Tensor.data_ptr()) ----
    def stage1_device(self, buf_ptr, length, idx_ptr, idx_words, stream=0):
        rc = self.L.sjgpu_stage1_device(self.h, buf_ptr, int(length), idx_ptr, int(idx_words), stream or None)
        if rc < 0:
            raise SjgpuError(f"sjgpu_stage1_device error {rc}: {self.last_error()}")
        return rc

    def stage1_tokens_device(self, buf_ptr, length, idx_ptr, idx_words, tok_ptr, tok_bytes, stream=0):
        """sjgpu_stage1_tokens_device: stage1_device + tok[i] = buf[idx[i]] beside the offsets (either pipeline; AUTO: the split one beyond the small-input limit)"""
        rc = self.L.sjgpu_stage1_tokens_device(self.h, buf_ptr, int(length), idx_ptr, int(idx_words), tok_ptr, int(tok_bytes), stream or None)
        if rc < 0:
            raise SjgpuError(f"sjgpu_stage1_tokens_device error {rc}: {self.last_error()}")
        return rc

    def minify_device(self, buf_ptr, length, dst_ptr, stream=0):
        rc = self.L.sjgpu_minify_device(self.h, buf_ptr, int(length), dst_ptr, stream or None)
        if rc < 0:
            raise SjgpuError(f"sjgpu_minify_device error {rc}: {self.last_error()}")
        return rc

    def validate_utf8_device(self, buf_ptr, length, stream=0):
        rc = self.L.sjgpu_validate_utf8_device(self.h, buf_ptr, int(length), stream or None)
        if rc < 0:
            raise SjgpuError(f"sjgpu_validate_utf8_device error {rc}: {self.last_error()}")
        return rc

    # ---- shards of one large document (sjgpu.h, "one large document sharded across GPUs") ----
    def string_parity_device(self, buf_ptr, length, stream=0):
        """-> 1 iff the shard holds an odd number of unescaped quotes (waits for `stream`)."""
        rc = self.L.sjgpu_string_parity_device(self.h, buf_ptr, int(length), stream or None)
        if rc != 0:
            raise SjgpuError(f"sjgpu_string_parity_device error {rc}: {self.last_error()}")
        return self.result(stream)[0] & 1

    def stage1_shard_device(self, buf_ptr, length, in_string, idx_ptr, idx_words, stream=0):
        rc = self.L.sjgpu_stage1_shard_device(self.h, buf_ptr, int(length), int(in_string), idx_ptr, int(idx_words), stream or None)
        if rc != 0:
            raise SjgpuError(f"sjgpu_stage1_shard_device error {rc}: {self.last_error()}")

    def minify_shard_device(self, buf_ptr, length, in_string, dst_ptr, stream=0):
        rc = self.L.sjgpu_minify_shard_device(self.h, buf_ptr, int(length), int(in_string), dst_ptr, stream or None)
        if rc != 0:
            raise SjgpuError(f"sjgpu_minify_shard_device error {rc}: {self.last_error()}")

    # ---- ranges of one resident buffer (sjgpu.h, "ranges of ONE resident buffer") ----
    def stage1_range_device(self, buf_ptr, begin, end, more, in_string, n_before, idx_ptr, idx_words, stream=0):
        rc = self.L.sjgpu_stage1_range_device(self.h, buf_ptr, int(begin), int(end), int(more), int(in_string), int(n_before),
                                              idx_ptr, int(idx_words), stream or None)
        if rc != 0:
            raise SjgpuError(f"sjgpu_stage1_range_device error {rc}: {self.last_error()}")

    def minify_range_device(self, buf_ptr, begin, end, more, in_string, out_before, dst_ptr, stream=0):
        rc = self.L.sjgpu_minify_range_device(self.h, buf_ptr, int(begin), int(end), int(more), int(in_string), int(out_before),
                                              dst_ptr, stream or None)
        if rc != 0:
            raise SjgpuError(f"sjgpu_minify_range_device error {rc}: {self.last_error()}")

    # ---- many small documents in one launch (sjgpu.h, "many small documents in one launch") ----
    def stage1_many(self, documents):
        """documents: list of bytes-like.  -> list of (error_code, n, idx[0..n+2]) as stage1(doc, REGULAR) would give."""
        arrays = [_as_u8(d) for d in documents]
        outs = [np.zeros(len(a) + 3, dtype=np.uint32) for a in arrays]
        docs = (Doc * len(arrays))()
        for k, (a, o) in enumerate(zip(arrays, outs)):
            docs[k] = Doc(a.ctypes.data if len(a) else 1, len(a), o.ctypes.data, len(o), 0, 0)
        rc = self.L.sjgpu_stage1_many(self.h, docs, len(arrays))
        if rc != 0:
            raise SjgpuError(f"sjgpu_stage1_many error {rc}: {self.last_error()}")
        return [(int(d.error), int(d.n), o[: d.n + 3].copy()) for d, o in zip(docs, outs)]

    def prepare_many(self, documents):
        """The sjgpu_doc array of a batch, marshalled ONCE: (docs, arrays, outs).  stage1_many_prepared then times the library, not ctypes."""
        arrays = [_as_u8(d) for d in documents]
        total = sum(len(a) + 3 for a in arrays)
        pool = np.zeros(total, dtype=np.uint32)  # one allocation for all lists
        docs = (Doc * len(arrays))()
        at = 0
        outs = []
        for k, a in enumerate(arrays):
            o = pool[at: at + len(a) + 3]
            at += len(a) + 3
            outs.append(o)
            docs[k] = Doc(a.ctypes.data if len(a) else 1, len(a), o.ctypes.data, len(o), 0, 0)
        return docs, arrays, outs

    def stage1_many_prepared(self, prepared):
        docs = prepared[0]
        rc = self.L.sjgpu_stage1_many(self.h, docs, len(docs))
        if rc != 0:
            raise SjgpuError(f"sjgpu_stage1_many error {rc}: {self.last_error()}")
        return docs

    # ---- the structural list after the scan, on the device ----
    def stage1_finish_device(self, buf_ptr, length, mode, idx_ptr, n_raw, flags, stream=0):
        """-> (error_code, n, next_start); idx (device) is left holding what stage1(mode) would have delivered."""
        n = ctypes.c_uint32(0)
        nxt = ctypes.c_uint32(0)
        rc = self.L.sjgpu_stage1_finish_device(self.h, buf_ptr, int(length), int(mode), idx_ptr, int(n_raw), int(flags), stream or None,
                                               ctypes.byref(n), ctypes.byref(nxt))
        if rc < 0:
            raise SjgpuError(f"sjgpu_stage1_finish_device error {rc}: {self.last_error()}")
        return rc, int(n.value), int(nxt.value)

    def depth_scan_device(self, buf_ptr, idx_ptr, n, depth_ptr, stream=0):
        rc = self.L.sjgpu_depth_scan_device(self.h, buf_ptr, idx_ptr, int(n), depth_ptr, stream or None)
        if rc != 0:
            raise SjgpuError(f"sjgpu_depth_scan_device error {rc}: {self.last_error()}")

    def depth_scan_tokens_device(self, tok_ptr, n, depth_ptr, stream=0):
        rc = self.L.sjgpu_depth_scan_tokens_device(self.h, tok_ptr, int(n), depth_ptr, stream or None)
        if rc != 0:
            raise SjgpuError(f"sjgpu_depth_scan_tokens_device error {rc}: {self.last_error()}")

    def parse_strings_device(self, buf_ptr, length, idx_ptr, n, out_ptr, out_bytes, offsets_ptr=0, allow_replacement=False, stream=0):
        """sjgpu_parse_strings_device -> (error_code, string buffer bytes used, strings, index of the first invalid string)"""
        used, cnt, bad = ctypes.c_uint64(0), ctypes.c_uint32(0), ctypes.c_uint32(0)
        rc = self.L.sjgpu_parse_strings_device(self.h, buf_ptr, int(length), idx_ptr, int(n), int(bool(allow_replacement)), out_ptr, int(out_bytes),
                                               offsets_ptr or None, stream or None, ctypes.byref(used), ctypes.byref(cnt), ctypes.byref(bad))
        if rc < 0:
            raise SjgpuError(f"sjgpu_parse_strings_device error {rc}: {self.last_error()}")
        return rc, int(used.value), int(cnt.value), int(bad.value)

    def string_path(self):
        """sjgpu_debug_string_path: 1 = the last string buffer came from the stream compaction, 2 = from the per-string walk"""
        return int(self.L.sjgpu_debug_string_path(self.h))

    def match_keys_device(self, buf_ptr, length, idx_ptr, n, names, match_ptr, stream=0):
        """sjgpu_match_keys_device: names = list of bytes; match_ptr -> n uint32 on the device.  Returns the number of matching keys."""
        blob = b"".join(names)
        lens = np.array([len(x) for x in names], dtype=np.uint32)
        m = ctypes.c_uint32(0)
        rc = self.L.sjgpu_match_keys_device(self.h, buf_ptr, int(length), idx_ptr, int(n), ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p), lens.ctypes.data,
                                            len(names), match_ptr, stream or None, ctypes.byref(m))
        if rc != 0:
            raise SjgpuError(f"sjgpu_match_keys_device error {rc}: {self.last_error()}")
        return int(m.value)

    def stage2_device(self, buf_ptr, length, idx_ptr, n, tape_ptr, tape_cap_words, strbuf_ptr, strbuf_bytes, max_depth=1024, stream=0, tok_ptr=0):
        """sjgpu_stage2_device (tok_ptr: sjgpu_stage2_tokens_device, the token stream of stage1_tokens_device beside the list)
        -> (simdjson error_code, tape words, string buffer bytes); raises on infrastructure errors"""
        tw, sb = ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self.L.sjgpu_stage2_tokens_device(self.h, buf_ptr, int(length), idx_ptr, int(n), tok_ptr or None, int(max_depth), tape_ptr, int(tape_cap_words), strbuf_ptr,
                                               int(strbuf_bytes), stream or None, ctypes.byref(tw), ctypes.byref(sb))
        if rc < 0:
            raise SjgpuError(f"sjgpu_stage2_device error {rc}: {self.last_error()}")
        return rc, int(tw.value), int(sb.value)

    def parse(self, data, max_depth=1024):
        """dom_parser_implementation::parse for a host buffer: (error_code, tape as uint64 array, string_buf as uint8 array)"""
        a = _as_u8(data)
        tape = np.zeros(len(a) + 8, dtype=np.uint64)
        sbuf = np.zeros(5 * (len(a) // 3) + 256, dtype=np.uint8)
        tw, sb = ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self.L.sjgpu_parse(self.h, a.ctypes.data if len(a) else None, len(a), int(max_depth), tape.ctypes.data, len(tape), sbuf.ctypes.data, len(sbuf),
                                ctypes.byref(tw), ctypes.byref(sb))
        if rc < 0:
            raise SjgpuError(f"sjgpu_parse error {rc}: {self.last_error()}")
        return rc, tape[: tw.value], sbuf[: sb.value]

    def result(self, stream=0):  # waits for `stream`
        r = ScanResult()
        rc = self.L.sjgpu_result(self.h, stream or None, ctypes.byref(r))
        if rc != 0:
            raise SjgpuError(f"sjgpu_result error {rc}: {self.last_error()}")
        return int(r.n), int(r.flags), int(r.out_len)


    def debug_trace_stage1(self, buf_ptr, length, idx_ptr, idx_words, tiles):
        """-> uint64[tiles, 8] wall-clock stamps (100 MHz) of the single-pass kernel's phases."""
        out = np.zeros((int(tiles), 8), dtype=np.uint64)
        rc = self.L.sjgpu_debug_trace_stage1(self.h, buf_ptr, int(length), idx_ptr, int(idx_words), out.ctypes.data, int(tiles))
        if rc != 0:
            raise SjgpuError(f"sjgpu_debug_trace_stage1 error {rc}: {self.last_error()}")
        return out

    def set_pipeline(self, pipeline="auto"):
        """"split" | "fused" | "auto" (also accepts True = fused / False = split)."""
        if isinstance(pipeline, bool):
            pipeline = "fused" if pipeline else "split"
        return self.L.sjgpu_set_pipeline(self.h, {"split": 0, "fused": 1, "auto": 2}[pipeline])

    def last_pipeline(self):
        """"fused" | "split": what the last enqueued scan used (AUTO decides by size and, for stage 1, output density)."""
        return "fused" if self.L.sjgpu_last_pipeline(self.h) == 1 else "split"

    def profile_kernel(self):
        """Name(s) of the scan kernel(s) the last enqueued call launched, as the launcher reported them."""
        return self.L.sjgpu_profile_kernel(self.h).decode()

    def profile_enable(self, on=True):
        rc = self.L.sjgpu_profile_enable(self.h, 1 if on else 0)
        if rc != 0:
            raise SjgpuError(f"sjgpu_profile_enable error {rc}: {self.last_error()}")

    def profile_read(self):
        """-> ([ms_sum per kernel slot], calls) accumulated since the last read (HIP events on the launch stream)."""
        ms = (ctypes.c_double * 3)()
        calls = ctypes.c_uint32(0)
        rc = self.L.sjgpu_profile_read(self.h, ms, ctypes.byref(calls))
        if rc != 0:
            raise SjgpuError(f"sjgpu_profile_read error {rc}: {self.last_error()}")
        return [float(x) for x in ms], int(calls.value)


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 bytes rank 0 makes and hands to every other rank (sjgpu_comm_unique_id = ncclGetUniqueId)"""
    L = load_library()
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    rc = L.sjgpu_comm_unique_id(buf, COMM_ID_BYTES)
    if rc != 0:
        raise SjgpuError(f"sjgpu_comm_unique_id error {rc}")
    return bytes(buf.raw)


class Comm:
    """sjgpu_comm_*: the RCCL communicator of the index concatenation, one process per GPU (collective calls)."""

    def __init__(self, rank, world, unique_id, device):
        self.L = load_library()
        self.h = ctypes.c_void_p()
        self.rank, self.world = rank, world
        rc = self.L.sjgpu_comm_create(rank, world, unique_id, len(unique_id), device, ctypes.byref(self.h))
        if rc != 0:
            raise SjgpuError(f"sjgpu_comm_create error {rc}: {self.L.sjgpu_comm_last_error(None).decode()}")

    def ranks(self):
        """ncclCommCount: the ranks RCCL itself sees in this communicator"""
        return int(self.L.sjgpu_comm_ranks(self.h))

    def close(self):
        if self.h:
            self.L.sjgpu_comm_destroy(self.h)
            self.h = ctypes.c_void_p()

    def gather_indices(self, idx_ptr, n, base, root, out_ptr, out_cap_words, stream=0):
        """-> (total structurals, [n of every rank]); the root's out array holds base + offset as u64, shards in rank order"""
        total = ctypes.c_uint64(0)
        counts = (ctypes.c_uint64 * self.world)()
        rc = self.L.sjgpu_comm_gather_indices(self.h, idx_ptr, int(n), int(base), int(root), out_ptr or None, int(out_cap_words), ctypes.byref(total), counts,
                                              stream or None)
        if rc != 0:
            raise SjgpuError(f"sjgpu_comm_gather_indices error {rc}: {self.L.sjgpu_comm_last_error(self.h).decode()}")
        return int(total.value), [int(c) for c in counts]


class MultiGpu:
    """sjgpu_mgpu (include/sjgpu.h): one host buffer, one shard per listed device, one process.  Same member names as
    DomParserImplementation for the host-buffer calls."""

    def __init__(self, devices):
        self.L = load_library()
        arr = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self.L.sjgpu_mgpu_create(arr, len(devices), ctypes.byref(h))
        if rc != 0:
            raise SjgpuError(f"sjgpu_mgpu_create({list(devices)}) failed with {rc}")
        self.h = h
        self.n_structural_indexes = 0
        self.next_structural_index = 0
        self.structural_indexes = np.zeros(0, dtype=np.uint32)

    def close(self):
        if getattr(self, "h", None):
            self.L.sjgpu_mgpu_destroy(self.h)
            self.h = None

    __del__ = close

    def stage1(self, data, mode=REGULAR):
        a = _as_u8(data)
        if len(self.structural_indexes) < len(a) + 3:
            self.structural_indexes = np.zeros(len(a) + 64, dtype=np.uint32)
        n = ctypes.c_uint32(self.n_structural_indexes)
        nxt = ctypes.c_uint32(self.next_structural_index)
        rc = self.L.sjgpu_mgpu_stage1(self.h, a.ctypes.data, len(a), int(mode), self.structural_indexes.ctypes.data, len(self.structural_indexes),
                                      ctypes.byref(n), ctypes.byref(nxt))
        self.n_structural_indexes, self.next_structural_index = int(n.value), int(nxt.value)
        if rc < 0:
            raise SjgpuError(f"sjgpu_mgpu_stage1 infrastructure error {rc}")
        return rc

    def minify(self, data):
        a = _as_u8(data)
        dst = np.zeros(max(len(a), 1), dtype=np.uint8)
        n = ctypes.c_size_t(0)
        rc = self.L.sjgpu_mgpu_minify(self.h, a.ctypes.data, len(a), dst.ctypes.data, ctypes.byref(n))
        if rc < 0:
            raise SjgpuError(f"sjgpu_mgpu_minify infrastructure error {rc}")
        return rc, dst[: n.value]

    def validate_utf8(self, data):
        a = _as_u8(data)
        ok = ctypes.c_int(0)
        rc = self.L.sjgpu_mgpu_validate_utf8(self.h, a.ctypes.data, len(a), ctypes.byref(ok))
        if rc != 0:
            raise SjgpuError(f"sjgpu_mgpu_validate_utf8 error {rc}")
        return bool(ok.value)


def stream_register(arr):
    """sjgpu_stream_register over a numpy uint8 array: stage1 calls on windows (views) of it are answered from a look-ahead span"""
    L = load_library()
    L.sjgpu_stream_register.restype = ctypes.c_int
    L.sjgpu_stream_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    return L.sjgpu_stream_register(arr.ctypes.data, arr.nbytes)


def stream_unregister(arr, named=False):
    """named: say WHICH registration over that base leaves (sjgpu_stream_unregister_len: the one made with this array's length)"""
    L = load_library()
    L.sjgpu_stream_unregister.restype = ctypes.c_int
    L.sjgpu_stream_unregister.argtypes = [ctypes.c_void_p]
    L.sjgpu_stream_unregister_len.restype = ctypes.c_int
    L.sjgpu_stream_unregister_len.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    return L.sjgpu_stream_unregister_len(arr.ctypes.data, arr.nbytes) if named else L.sjgpu_stream_unregister(arr.ctypes.data)


def stream_extent(arr):
    L = load_library()
    L.sjgpu_debug_stream_extent.restype = ctypes.c_size_t
    L.sjgpu_debug_stream_extent.argtypes = [ctypes.c_void_p]
    return int(L.sjgpu_debug_stream_extent(arr.ctypes.data))


def stage1_error_from_flags(n, flags):
    return int(load_library().sjgpu_stage1_error_from_flags(int(n), int(flags)))


def host_register(arr):
    """Page-lock a long-lived numpy array (sjgpu_host_register); call host_unregister before dropping it."""
    rc = load_library().sjgpu_host_register(arr.ctypes.data, arr.nbytes)
    if rc != 0:
        raise SjgpuError(f"sjgpu_host_register failed with {rc}")


def host_unregister(arr):
    load_library().sjgpu_host_unregister(arr.ctypes.data)


def clean_cut(buf, target):
    """First cut >= target where only the in-string bit crosses (sjgpu_clean_cut), or len(buf)."""
    a = _as_u8(buf)
    return int(load_library().sjgpu_clean_cut(a.ctypes.data, len(a), int(target)))


def device_count():
    return int(load_library().sjgpu_device_count())
