"""lab: does the ALIGNMENT of the caller's buffers decide the split pipeline's pace?  (bench.py's NDJSON leg: 0.35-0.38 ms when torch carved its buffers out of
cached blocks, 0.30 from a clean allocator.)  One document, one context; the input and the output as views at byte offsets into two large allocations."""
import os, sys, json, statistics
sys.path.insert(0, os.getcwd())
import torch
from simdjson_amd import capi, corpus
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for kind, pipe in (("amazon_ndjson", "split"), ("large_random", "fused"), ("large_random", "split")):
    host = getattr(corpus, kind)(1 << 30, 2000)[0]
    L = len(host)
    big_in = torch.empty(L + (8 << 20), dtype=torch.uint8, device="cuda")
    big_out = torch.empty(4 * (L + 16) + (8 << 20), dtype=torch.uint8, device="cuda")
    p = capi.DomParserImplementation(L); p.set_pipeline(pipe)
    h = torch.from_numpy(host)
    rows = []
    for off_in, off_out in ((0, 0), (16, 0), (64, 0), (128, 0), (512, 0), (1024, 0), (2048, 0), (4096, 0), (65536, 0), (1 << 20, 0), (0, 16), (0, 128), (0, 512), (0, 4096), (0, 65536), (512, 512), (2048, 2048)):
        buf = big_in[off_in: off_in + L]
        buf.copy_(h)
        out = big_out[off_out: off_out + 4 * (L + 16)]
        for _ in range(30):
            p.stage1_device(buf.data_ptr(), L, out.data_ptr(), L + 3, st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(10):
                p.stage1_device(buf.data_ptr(), L, out.data_ptr(), L + 3, st)
            e1.record(); e1.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / 10)
        rows.append((off_in, off_out, round(statistics.median(ts), 1)))
    print(json.dumps({"workload": kind, "pipeline": pipe, "base_in": hex(big_in.data_ptr()), "base_out": hex(big_out.data_ptr()), "us_by_offsets_in_out": rows}), flush=True)
    p.close(); del big_in, big_out
