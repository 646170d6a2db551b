"""Device-resident stage-1 time vs input size for both pipelines, plus the host-buffer (plug-in) path that
pays H2D of the document and D2H of the indices (PCIe-inclusive; reported in DESIGN.md, never as `value`)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simdjson_amd import capi, corpus

kind = sys.argv[1] if len(sys.argv) > 1 else "twitter_like"
rows = []
for size in (64 << 10, 631515, 4 << 20, 12 << 20, 32 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20, 1 << 30):
    a, _ = getattr(corpus, kind)(size, 5)
    L = len(a)
    p = capi.DomParserImplementation(L)
    buf = torch.from_numpy(a).cuda(); idx = torch.empty(L + 3, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    row = {"kind": kind, "bytes": L}
    for name, fused in (("fused", True), ("split", False)):
        p.set_pipeline(fused)
        for _ in range(3): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
        torch.cuda.synchronize()
        reps = 100 if L < (64 << 20) else 12
        dt = 1e9
        for _trial in range(3):  # best of 3 batches: single batches occasionally catch a one-off stall
            t0 = time.perf_counter()
            for _ in range(reps): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / reps)
        row[name + "_us"] = round(dt * 1e6, 1); row[name + "_GBps"] = round(L / dt / 1e9, 1)
    # host-buffer path (what the simdjson plug-in pays): pageable host memory in, pageable out
    p.set_pipeline("auto")
    def host_path(q):
        for _ in range(2): q.stage1(a)
        reps = 50 if L < (64 << 20) else 3
        best = 1e9
        for _trial in range(2):
            t0 = time.perf_counter()
            for _ in range(reps): q.stage1(a)
            best = min(best, (time.perf_counter() - t0) / reps)
        return best
    dt = host_path(p)
    row["host_path_us"] = round(dt * 1e6, 1); row["host_path_GBps"] = round(L / dt / 1e9, 2); row["n"] = p.n_structural_indexes
    p.close()
    if L >= (16 << 20):  # the same call without the range-by-range overlap (upload, scan, download one after the other)
        os.environ["SJGPU_STREAM_FROM_MB"] = "0"
        q = capi.DomParserImplementation(L)
        dt = host_path(q)
        row["host_path_serial_us"] = round(dt * 1e6, 1); row["host_path_serial_GBps"] = round(L / dt / 1e9, 2)
        q.close()
        del os.environ["SJGPU_STREAM_FROM_MB"]
    rows.append(row); print(json.dumps(row), flush=True)
