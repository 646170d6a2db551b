"""lab: do the streaming kernels hold their pace?  (final run 1 of round 5: the bench's NDJSON leg, timed behind 128 untimed calls and behind the other legs,
ran at 0.382 ms where its first 20 calls had taken 0.317.)  One workload at a time, calls back to back for ~1.5 s, the mean of every block of 100 calls by
events; then 2 s of idling and the same again.  Prints the series."""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch
from simdjson_amd import capi, corpus
st = torch.cuda.current_stream().cuda_stream
def series(call, blocks, per=100):
    out = []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    ev[0].record()
    for b in range(blocks):
        for _ in range(per):
            call()
        ev[b + 1].record()
    torch.cuda.synchronize()
    return [round(1e3 * ev[b].elapsed_time(ev[b + 1]) / per, 1) for b in range(blocks)]
for kind, pipe, op in (("amazon_ndjson", "split", "stage1"), ("large_random", "auto", "validate_utf8"), ("large_random", "fused", "stage1"), ("amazon_ndjson", "split", "stage1")):
    a = getattr(corpus, kind)(1 << 30, 1000)[0]
    L = len(a)
    buf = torch.from_numpy(a).cuda(); idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    p = capi.DomParserImplementation(L); p.set_pipeline(pipe)
    call = (lambda: p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)) if op == "stage1" else (lambda: p.validate_utf8_device(buf.data_ptr(), L, st))
    for _ in range(3): call()
    torch.cuda.synchronize()
    time.sleep(2.0)
    s1 = series(call, 50)
    time.sleep(2.0)
    s2 = series(call, 15)
    print(json.dumps({"workload": kind, "pipeline": pipe, "op": op, "us_per_call_by_block_of_100": s1, "after_2_s_idle": s2}), flush=True)
    p.close(); del buf, idx
