import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from simdjson_amd import capi, corpus
for kind, gen in (("large_random", corpus.large_random), ("deep_nesting", corpus.deep_nesting_doc)):
    for size in (768 << 20, 1 << 30):
        a, _ = gen(size, 1000)
        L = len(a)
        p = capi.DomParserImplementation(L)
        buf = torch.from_numpy(a).cuda(); idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        row = {"kind": kind, "bytes": L}
        for name in ("fused", "split", "fused", "split"):
            p.set_pipeline(name)
            for _ in range(3): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
            torch.cuda.synchronize()
            dt = 1e9
            for _trial in range(3):
                t0 = time.perf_counter()
                for _ in range(15): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
                torch.cuda.synchronize()
                dt = min(dt, (time.perf_counter() - t0) / 15)
            row.setdefault(name + "_us", []).append(round(dt * 1e6, 1))
        p.close()
        del buf, idx
        print(json.dumps(row), flush=True)
