"""Why does the same kernel take 4-16 % longer inside bench.py than in a bare loop?  One process: the bare loop (scripts/events_ab.py's) over the bench's own
buffer, then bench.device_leg over the same host buffer, then the bare loop again."""
import os, sys, time, json, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from simdjson_amd import capi, corpus
import bench

def bare(host, op="stage1", pipeline="fused", steps=20):
    L = len(host)
    p = capi.DomParserImplementation(L)
    p.set_pipeline(pipeline)
    buf = torch.from_numpy(host).cuda()
    out = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(4): p.stage1_device(buf.data_ptr(), L, out.data_ptr(), L + 3, st)
    torch.cuda.synchronize()
    p.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): p.stage1_device(buf.data_ptr(), L, out.data_ptr(), L + 3, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms, calls = p.profile_read()
    p.close()
    return round(dt * 1e3, 4), round(ms[0] / calls, 4)

for kind, seed in (("amazon_ndjson", 2000), ("large_random", 1000)):
    host, units = getattr(corpus, kind)(1 << 30, seed)
    print(kind, "bare", bare(host), flush=True)
    args = types.SimpleNamespace(size=1 << 30, no_cpu_baseline=True)
    cx = bench.Ctx(args, torch, capi, corpus, 0)
    leg = bench.device_leg(cx, "stage1", kind, host, units, 20, 3, "fused", with_cpu=False, with_parity=False)
    print(kind, "device_leg", leg["ms_per_step"], leg["roofline"]["gpu_ms_per_step"], leg["roofline"]["kernel"], flush=True)
    print(kind, "bare", bare(host), flush=True)
