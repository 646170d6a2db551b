"""Host-buffer (plug-in) path: range size of the overlapped upload/scan/download pipeline vs throughput."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simdjson_amd import capi, corpus

for kind, size in (("large_random", 1 << 30), ("twitter_like", 1 << 30), ("large_random", 64 << 20), ("twitter_like", 64 << 20)):
    a, _ = getattr(corpus, kind)(size, 5)
    L = len(a)
    for chunk in (0, 8, 16, 32, -1):
        os.environ.pop("SJGPU_STREAM_FROM_MB", None); os.environ.pop("SJGPU_STREAM_CHUNK_MB", None)
        if chunk == 0:
            os.environ["SJGPU_STREAM_FROM_MB"] = "0"
        elif chunk > 0:
            os.environ["SJGPU_STREAM_FROM_MB"] = "1"
            os.environ["SJGPU_STREAM_CHUNK_MB"] = str(chunk)
        p = capi.DomParserImplementation(L)
        for _ in range(2): p.stage1(a)
        reps = 3 if L > (256 << 20) else 10
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps): p.stage1(a)
            best = min(best, (time.perf_counter() - t0) / reps)
        p.minify(a)
        tm = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); rc, out = p.minify(a); tm = min(tm, time.perf_counter() - t0)
        print(json.dumps({"kind": kind, "bytes": L, "range_MiB": {0: "serial", -1: "default"}.get(chunk, chunk), "stage1_ms": round(best * 1e3, 3),
                          "stage1_GBps": round(L / best / 1e9, 2), "minify_ms": round(tm * 1e3, 3), "minify_GBps": round(L / tm / 1e9, 2),
                          "n": p.n_structural_indexes}), flush=True)
        p.close()
