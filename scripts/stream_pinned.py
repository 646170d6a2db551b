"""Host-buffer path with page-locked buffers on both sides (document registered, index array registered):
no per-call pinning, so this is what the PCIe link and the overlap are worth by themselves."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simdjson_amd import capi, corpus

kind = sys.argv[1] if len(sys.argv) > 1 else "large_random"
a, _ = getattr(corpus, kind)(1 << 30, 5)
L = len(a)
capi.host_register(a)
for label, env in (("serial", {"SJGPU_STREAM_FROM_MB": "0"}),
                   ("ranges 8 MiB, 1+1 copy threads", {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "8", "SJGPU_COPY_THREADS": "1"}),
                   ("ranges 8 MiB, 2+2 copy threads", {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "8", "SJGPU_COPY_THREADS": "2"}),
                   ("ranges 16 MiB, 1+1 copy threads", {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "16", "SJGPU_COPY_THREADS": "1"}),
                   ("ranges 4 MiB, 1+1 copy threads", {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "4", "SJGPU_COPY_THREADS": "1"})):
    for k in ("SJGPU_STREAM_FROM_MB", "SJGPU_STREAM_CHUNK_MB", "SJGPU_COPY_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    by_ctx = []
    for rep in range(3):
        p = capi.DomParserImplementation(L)
        words = p.n_structural_indexes  # unused; the array below is what gets registered
        capi.host_register(p.structural_indexes)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); p.stage1(a); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
        by_ctx.append(ts)
        capi.host_unregister(p.structural_indexes)
        p.close()
    print(json.dumps({"kind": kind, "bytes": L, "path": label, "buffers": "registered (page-locked)", "ms_per_call_by_context": by_ctx}), flush=True)
