"""Is the overlapped host path stable from context to context?  N fresh contexts, same configuration; with
SJGPU_DEBUG_STREAM the library reports where each call's time went (stderr)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simdjson_amd import capi, corpus

kind = sys.argv[1] if len(sys.argv) > 1 else "large_random"
threads = sys.argv[2] if len(sys.argv) > 2 else "1"
a, _ = getattr(corpus, kind)(1 << 30, 5)
L = len(a)
os.environ["SJGPU_DEBUG_STREAM"] = "1"
os.environ["SJGPU_COPY_THREADS"] = threads
for chunk in (8, 8):
    os.environ["SJGPU_STREAM_FROM_MB"] = "1"; os.environ["SJGPU_STREAM_CHUNK_MB"] = str(chunk)
    times = []
    for rep in range(4):
        p = capi.DomParserImplementation(L)
        p.stage1(a)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); p.stage1(a); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
        times.append(ts)
        sys.stderr.write(f"[repeat] {kind} range {chunk} MiB threads {threads} context {rep}: {ts}\n")
        p.close()
    print(json.dumps({"kind": kind, "range_MiB": chunk, "copy_threads": threads, "ms_per_call_by_context": times}), flush=True)
