"""Per-call GPU time of 80 back-to-back device-resident stage-1 / minify calls (torch events around each call): does the time settle, and where?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from simdjson_amd import capi, corpus
a, _ = corpus.large_random(1 << 30, 1000)
L = len(a)
buf = torch.from_numpy(a).cuda()
st = torch.cuda.current_stream().cuda_stream
for op in ("stage1", "minify", "stage1"):
    p = capi.DomParserImplementation(L)
    dst = torch.empty((L + 16) * (4 if op == "stage1" else 1), dtype=torch.uint8, device="cuda")
    step = (lambda: p.stage1_device(buf.data_ptr(), L, dst.data_ptr(), L + 3, st)) if op == "stage1" else (lambda: p.minify_device(buf.data_ptr(), L, dst.data_ptr(), st))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
    ev[0].record()
    for k in range(80):
        step()
        ev[k + 1].record()
    torch.cuda.synchronize()
    print(op, [round(ev[k].elapsed_time(ev[k + 1]) * 1e3) for k in range(80)], flush=True)
    p.close(); del dst
