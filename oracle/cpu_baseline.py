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


def time_cpu_ndjson_threads(buf: np.ndarray, threads: int, iters: int):
    """NDJSON on T independent host threads, each with its own reference parser on a newline-aligned
    slice (the CPU analogue of one shard per GPU).  -> dict(value GB/s aggregate, cores = threads used, ...) or None
    when the reference library is absent."""
    import threading
    import time
    if not os.path.exists(LIB_REF):
        return None
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    L = ctypes.CDLL(LIB_REF)
    L.sjref_available.argtypes = [ctypes.c_char_p]
    L.sjref_parser_create.restype = ctypes.c_void_p
    L.sjref_parser_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    L.sjref_parser_stage1.restype = ctypes.c_int
    L.sjref_parser_stage1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.sjref_parser_destroy.argtypes = [ctypes.c_void_p]
    impl = next((i for i in (b"icelake", b"haswell", b"westmere") if L.sjref_available(i)), None)
    if impl is None:
        return None
    n = len(buf)
    cuts = [0]
    for k in range(1, threads):  # newline-aligned, like simdjson_amd.sharded.newline_cuts
        t = max((k * n) // threads, cuts[-1])
        nl = np.flatnonzero(buf[t:min(n, t + (1 << 22))] == 0x0A)
        cuts.append(n if len(nl) == 0 else t + int(nl[0]) + 1)
    cuts.append(n)
    slices = [(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo]
    parsers = [L.sjref_parser_create(impl, hi - lo) for lo, hi in slices]
    counts = [ctypes.c_uint32(0) for _ in slices]
    errs = [0] * len(slices)
    gate = threading.Barrier(len(slices) + 1)

    def work(i):
        lo, hi = slices[i]
        ptr = buf.ctypes.data + lo
        L.sjref_parser_stage1(parsers[i], ptr, hi - lo, 0, None, ctypes.byref(counts[i]), None)  # warm: touches the index array
        gate.wait()
        for _ in range(iters):
            errs[i] |= L.sjref_parser_stage1(parsers[i], ptr, hi - lo, 0, None, ctypes.byref(counts[i]), None)
        gate.wait()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(slices))]
    for t in ts:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    for h in parsers:
        L.sjref_parser_destroy(h)
    return {"value": n * iters / 1e9 / dt, "unit": "GB/s", "cores": len(slices), "kind": "reference", "impl": impl.decode(),
            "n": int(sum(c.value for c in counts)), "seconds": dt / iters, "err": int(max(errs))}
