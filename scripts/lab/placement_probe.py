"""lab: the default bench line's NDJSON leg ran at 0.382 ms where the same workload alone in a process ran at 0.299 (round 5, final run 1).  Which state of the
process does that?  The split pipeline over 1 GiB of NDJSON: (A) first thing in the process; (B) the same context again after the process has allocated, used
and freed other buffers the way bench.py's earlier legs do; (C) a NEW context and NEW buffers at that point; (D) after torch.cuda.empty_cache(), new buffers."""
import os, sys, json, statistics
sys.path.insert(0, os.getcwd())
import torch
from simdjson_amd import capi, corpus
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
host = corpus.amazon_ndjson(1 << 30, 1000)[0]
L = len(host)

def timed(p, buf, idx, tag):
    for _ in range(100):
        p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        e0.record()
        for _ in range(10):
            p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
        e1.record(); e1.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / 10)
    p.profile_enable(True)
    for _ in range(10):
        p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
    torch.cuda.synchronize()
    ms, calls = p.profile_read()
    p.profile_enable(False)
    free, total = torch.cuda.mem_get_info()
    print(json.dumps({"tag": tag, "median_us": round(statistics.median(ts), 1), "slots_us": [round(1e3 * x / calls, 1) for x in ms], "buf": hex(buf.data_ptr()), "idx": hex(idx.data_ptr()),
                      "free_GiB": round(free / 2**30, 1), "torch_reserved_GiB": round(torch.cuda.memory_reserved() / 2**30, 2)}), flush=True)

def make():
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    p = capi.DomParserImplementation(L)
    p.set_pipeline("split")
    return p, buf, idx

pA, bufA, idxA = make()
timed(pA, bufA, idxA, "A: first thing in the process")
# what bench.py's legs in front of config3 do: other documents of 1 GiB with their outputs, other contexts, minify / validate, 256 MiB documents
for kind, pipe in (("large_random", "fused"), ("large_random", "split"), ("twitter_like", "split")):
    h2 = getattr(corpus, kind)(1 << 30, 1000)[0]
    b2 = torch.from_numpy(h2).cuda(); i2 = torch.empty(len(h2) + 16, dtype=torch.int32, device="cuda"); o2 = torch.empty(len(h2) + 64, dtype=torch.uint8, device="cuda")
    q = capi.DomParserImplementation(len(h2)); q.set_pipeline(pipe)
    for _ in range(20):
        q.stage1_device(b2.data_ptr(), len(h2), i2.data_ptr(), len(h2) + 3, st)
    q.set_pipeline("auto")
    for _ in range(10):
        q.minify_device(b2.data_ptr(), len(h2), o2.data_ptr(), st)
        q.validate_utf8_device(b2.data_ptr(), len(h2), st)
    torch.cuda.synchronize()
    q.close()
    del b2, i2, o2
timed(pA, bufA, idxA, "B: the same context and buffers after the other legs' work")
pC, bufC, idxC = make()
timed(pC, bufC, idxC, "C: a new context and new buffers now")
timed(pA, bufA, idxA, "B2: the first context again")
pC.close(); del bufC, idxC
torch.cuda.empty_cache()
pD, bufD, idxD = make()
timed(pD, bufD, idxD, "D: after empty_cache(), a new context and new buffers")
pA.close(); del bufA, idxA
pD2 = capi.DomParserImplementation(L); pD2.set_pipeline("split")
timed(pD2, bufD, idxD, "E: D's buffers, one more new context")
