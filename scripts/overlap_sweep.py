"""Stage 1 on one resident 1 GiB buffer per workload: the pipelines side by side, same box, same buffer.

    fused            k_fused_pipelined (single pass)
    split            summarize -> resolve -> emit, one after the other
    split/overlap=P  the split pipeline as a chain of P-MiB pieces whose scan and emission kernels overlap on two streams
                     (launch_stage1_pieces, env SJGPU_OVERLAP_MB)

Every variant's output is compared with the first variant's (count + order-sensitive digest); bench.py holds the digest
check against the reference.  Prints one JSON line per (workload, variant): ms per call from hipEvents around the whole call
on the launch stream (what bench.py's roofline uses) and from the host clock around 10 back-to-back calls."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from simdjson_amd import capi, corpus

SIZE = int(os.environ.get("SWEEP_SIZE", 1 << 30))
WORKLOADS = (sys.argv[1].split(",") if len(sys.argv) > 1 else ["large_random", "amazon_ndjson", "twitter_like"])
OVERLAPS = [int(x) for x in os.environ.get("SWEEP_OVERLAPS", "16,32,64,128,256").split(",") if x]


def digest(idx, count):
    w = idx[:count].to(torch.int64) & 0xFFFFFFFF
    return int((w * torch.arange(1, count + 1, dtype=torch.int64, device=idx.device)).sum().item()) & 0xFFFFFFFFFFFFFFFF


for wl in WORKLOADS:
    gen = {"deep_nesting": corpus.deep_nesting_doc}.get(wl) or getattr(corpus, wl)
    host, _ = gen(SIZE, 42)
    L = len(host)
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    want = None
    variants = [("fused", "fused", 0), ("split", "split", 0)] + [(f"split/overlap={p}", "split", p) for p in OVERLAPS]
    for name, pipeline, overlap in variants:
        if overlap:
            os.environ["SJGPU_OVERLAP_MB"] = str(overlap)
        else:
            os.environ.pop("SJGPU_OVERLAP_MB", None)
        p = capi.DomParserImplementation(L)
        p.set_pipeline(pipeline)
        idx.zero_()
        for _ in range(3):
            assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
        n, flags, _ = p.result(st)
        got = (n, flags, digest(idx, n + 3))
        if want is None:
            want = got
        ok = got == want
        p.profile_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 10
        ms, calls = p.profile_read()
        p.profile_enable(False)
        kernel = p.profile_kernel()
        gpu_ms = (ms[0] if "+" not in kernel else sum(ms)) / max(calls, 1)
        alg = L + 4 * (n + 3)
        print(json.dumps({"workload": wl, "variant": name, "same_output": ok, "n": n, "flags": flags, "gpu_ms": round(gpu_ms, 4), "wall_ms": round(wall * 1e3, 4),
                          "slots_ms": [round(m / max(calls, 1), 4) for m in ms], "input_GBps": round(L / gpu_ms / 1e6, 1),
                          "algorithmic_GBps": round(alg / gpu_ms / 1e6, 1), "frac_of_8TBps": round(alg / gpu_ms / 1e6 / 8000, 4), "kernel": kernel}), flush=True)
        p.close()
    del buf, idx
    os.environ.pop("SJGPU_OVERLAP_MB", None)
