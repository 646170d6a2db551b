"""The token-byte stream on both roads (round 6): sjgpu_stage1_tokens_device on the split pipeline (the scan kernel stages the structural bytes, the emission
kernel copies them) and on the single-pass kernels (the emitting wave gathers them out of the document), next to the plain calls, one process, calls interleaved.
    python scripts/tokens_roads.py [bytes ...]     us per call (HIP events around `reps` calls, median of the rounds)"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simdjson_amd import capi, corpus
sizes = [int(x) for x in sys.argv[1:]] or [1 << 30, 256 << 20, 16 << 20, 2 << 20]
st = torch.cuda.current_stream().cuda_stream
print("%-16s %11s | %9s %9s | %9s %9s   (us per call)" % ("workload", "bytes", "split", "+tokens", "fused", "+tokens"))
for size in sizes:
    for kind in ("amazon_ndjson", "twitter_like", "large_random"):
        host, _ = getattr(corpus, kind)(size, 3000)
        L = len(host)
        buf = torch.from_numpy(host).cuda()
        cap = L // 2 + 1024
        idx = torch.empty(cap + 16, dtype=torch.int32, device="cuda")
        tok = torch.empty(cap + 16, dtype=torch.uint8, device="cuda")
        ps = {}
        for road in ("split", "fused"):
            ps[road] = capi.DomParserImplementation(L)
            ps[road].set_pipeline(road)
        calls = {
            "split": lambda: ps["split"].stage1_device(buf.data_ptr(), L, idx.data_ptr(), cap, st),
            "split+tok": lambda: ps["split"].stage1_tokens_device(buf.data_ptr(), L, idx.data_ptr(), cap, tok.data_ptr(), cap + 16, st),
            "fused": lambda: ps["fused"].stage1_device(buf.data_ptr(), L, idx.data_ptr(), cap, st),
            "fused+tok": lambda: ps["fused"].stage1_tokens_device(buf.data_ptr(), L, idx.data_ptr(), cap, tok.data_ptr(), cap + 16, st),
        }
        reps = 10 if size >= (64 << 20) else 40
        times = {k: [] for k in calls}
        ref = None
        for k, f in calls.items():
            for _ in range(3):
                f()
            r = ps["split" if k.startswith("split") else "fused"].result(st)
            assert r[1] == 0 and (ref is None or r[0] == ref), (k, r, ref)
            ref = r[0]
        assert bool(torch.equal(tok[:ref], buf[idx[:ref].to(torch.int64)]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rnd in range(8):
            for k, f in (list(calls.items()) if rnd % 2 == 0 else list(calls.items())[::-1]):
                e0.record()
                for _ in range(reps):
                    f()
                e1.record()
                e1.synchronize()
                times[k].append(1e3 * e0.elapsed_time(e1) / reps)
        m = {k: statistics.median(v) for k, v in times.items()}
        print("%-16s %11d | %9.1f %9.1f | %9.1f %9.1f" % (kind, L, m["split"], m["split+tok"], m["fused"], m["fused+tok"]), flush=True)
        for p in ps.values():
            p.close()
        del buf, idx, tok
