"""Host-buffer path with WARM vs COLD host buffers.  The runtime caches the pinning of host ranges it has copied from
before (same address, same size, same stream), so a benchmark that parses the same array again and again measures the
warm case; an application that parses a fresh buffer each time pays for pinning every page it hands over.  Cold = a
rotation over several copies of the document at different addresses, each touched by the CPU, none copied before."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simdjson_amd import capi, corpus

kind = sys.argv[1] if len(sys.argv) > 1 else "large_random"
size = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 30)
a, _ = getattr(corpus, kind)(size, 5)
L = len(a)
NCOLD = 5
for label, env in (("serial", {"SJGPU_STREAM_FROM_MB": "0"}),
                   ("ranges 8 MiB, 1+1 copy threads", {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "8", "SJGPU_COPY_THREADS": "1"}),
                   ("ranges 8 MiB, 2+2 copy threads", {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "8", "SJGPU_COPY_THREADS": "2"}),
                   ("ranges 16 MiB, 2+2 copy threads", {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "16", "SJGPU_COPY_THREADS": "2"}),
                   ("ranges 8 MiB, 3+3 copy threads", {"SJGPU_STREAM_FROM_MB": "1", "SJGPU_STREAM_CHUNK_MB": "8", "SJGPU_COPY_THREADS": "3"})):
    for k in ("SJGPU_STREAM_FROM_MB", "SJGPU_STREAM_CHUNK_MB", "SJGPU_COPY_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    row = {"kind": kind, "bytes": L, "path": label}
    # cold: fresh copies of the document, fresh parser (fresh index array), every call sees memory the GPU has never read
    colds = [a.copy() for _ in range(NCOLD)]
    p = capi.DomParserImplementation(L)
    ts = []
    for c in colds:
        t0 = time.perf_counter(); p.stage1(c); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    row["cold_ms"] = ts
    del colds
    # warm: the same array again and again
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); p.stage1(a); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    row["warm_ms"] = ts
    row["n"] = p.n_structural_indexes
    p.close()
    print(json.dumps(row), flush=True)
