"""oracle/cpu_baseline.py -- TEST INFRASTRUCTURE (bench.py's `cpu_baseline` leg only).

Times the CPU side of the comparison on the host cores of the box, reference convention
(/root/reference/benchmark/benchmarker.h:315-346,418: pre-allocated parser, time only the call,
best of N, decimal GB of INPUT bytes):
  kind "reference": the real simdjson kernels from oracle/_ref/libsjref.so (icelake if the host has
                    AVX-512 VBMI2, else haswell, else westmere), single thread;
  kind "port":      oracle/sj_oracle.c (scalar restatement) when the reference library is absent.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_REF = os.path.join(HERE, "_ref", "libsjref.so")
LIB_ORACLE = os.path.join(HERE, "_ref", "libsjoracle.so")
WHICH = {"stage1": 0, "minify": 1, "validate_utf8": 2}


def time_cpu(buf: np.ndarray, op: str, iters: int):
    """-> dict(value GB/s, unit, cores, kind, impl, n (stage1 structural count or None), seconds)"""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    which = WHICH[op]
    if os.path.exists(LIB_REF):
        L = ctypes.CDLL(LIB_REF)
        L.sjref_available.argtypes = [ctypes.c_char_p]
        L.sjref_bench.restype = ctypes.c_double
        L.sjref_bench.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.sjref_parser_create.restype = ctypes.c_void_p
        L.sjref_parser_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.sjref_parser_stage1.restype = ctypes.c_int
        L.sjref_parser_stage1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
        L.sjref_parser_destroy.argtypes = [ctypes.c_void_p]
        impl = next((i for i in (b"icelake", b"haswell", b"westmere") if L.sjref_available(i)), None)
        if impl is not None:
            scratch = np.empty(len(buf) + 64 if which == 1 else 1, dtype=np.uint8)
            err = ctypes.c_int(0)
            best = L.sjref_bench(impl, which, buf.ctypes.data, len(buf), iters, scratch.ctypes.data, ctypes.byref(err))
            n = None
            if which == 0:
                h = L.sjref_parser_create(impl, len(buf))
                nn = ctypes.c_uint32(0)
                L.sjref_parser_stage1(h, buf.ctypes.data, len(buf), 0, None, ctypes.byref(nn), None)
                L.sjref_parser_destroy(h)
                n = int(nn.value)
            return {"value": len(buf) / 1e9 / best, "unit": "GB/s", "cores": 1, "kind": "reference",
                    "impl": impl.decode(), "n": n, "seconds": best, "err": int(err.value)}
    L = ctypes.CDLL(LIB_ORACLE)
    L.sjo_bench.restype = ctypes.c_double
    L.sjo_bench.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    scratch = np.empty(4 * (len(buf) + 3) if which == 0 else len(buf) + 64, dtype=np.uint8)
    best = L.sjo_bench(which, buf.ctypes.data, len(buf), max(1, iters // 4), scratch.ctypes.data)
    return {"value": len(buf) / 1e9 / best, "unit": "GB/s", "cores": 1, "kind": "port", "impl": "sj_oracle.c",
            "n": None, "seconds": best, "err": 0}
