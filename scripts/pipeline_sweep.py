"""Device-resident stage-1 time vs input size for both pipelines (what AUTO's thresholds rest on): python scripts/pipeline_sweep.py [workload ...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simdjson_amd import capi, corpus

for kind in (sys.argv[1:] or ["large_random", "twitter_like", "amazon_ndjson"]):
    for size in (2 << 20, 6 << 20, 12 << 20, 24 << 20, 48 << 20, 96 << 20, 160 << 20, 256 << 20, 512 << 20):
        a, _ = getattr(corpus, kind)(size, 5)
        L = len(a)
        p = capi.DomParserImplementation(L)
        buf = torch.from_numpy(a).cuda(); idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        row = {"kind": kind, "bytes": L}
        for name in ("fused", "split"):
            p.set_pipeline(name)
            for _ in range(3): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
            torch.cuda.synchronize()
            reps = 200 if L < (16 << 20) else (50 if L < (128 << 20) else 15)
            dt = 1e9
            for _trial in range(3):
                t0 = time.perf_counter()
                for _ in range(reps): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
                torch.cuda.synchronize()
                dt = min(dt, (time.perf_counter() - t0) / reps)
            row[name + "_us"] = round(dt * 1e6, 1)
        n, _, _ = p.result(st)
        row["density"] = round(n / L, 4)
        row["faster"] = "fused" if row["fused_us"] < row["split_us"] else "split"
        p.close()
        print(json.dumps(row), flush=True)
