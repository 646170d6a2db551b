"""AUTO's thresholds: the split pipeline against the single-pass kernel over sizes and texts, calls of the two taking turns (median of the rounds, HIP events
on the launch stream).  python scripts/pipeline_sweep.py [stage1|minify] [sizes in MiB ...]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.getcwd())
import torch
from simdjson_amd import capi, corpus

args = sys.argv[1:]
op = "stage1"
if args and args[0] in ("stage1", "minify"):
    op = args.pop(0)
kinds = ("large_random", "twitter_like", "amazon_ndjson", "deep_nesting_doc") if op == "stage1" else ("large_random", "twitter_like")
sizes = [int(x) for x in args] or [16, 64, 128, 192, 256, 384, 512, 768, 1024]
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for kind in kinds:
    for mib in sizes:
        a, _ = getattr(corpus, kind)(mib << 20, 1000)
        L = len(a)
        buf = torch.from_numpy(a).cuda()
        idx = torch.empty(L + 64, dtype=torch.int32 if op == "stage1" else torch.uint8, device="cuda")

        def call(p):
            if op == "stage1":
                p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
            else:
                p.minify_device(buf.data_ptr(), L, idx.data_ptr(), st)
        ps = {}
        for name in ("fused", "split"):
            ps[name] = capi.DomParserImplementation(L)
            ps[name].set_pipeline(name)
            for _ in range(3):
                call(ps[name])
        torch.cuda.synchronize()
        reps = max(4, min(20, (2 << 30) // max(L, 1)))
        t = {"fused": [], "split": []}
        for rnd in range(10):
            for name in (("fused", "split") if rnd % 2 == 0 else ("split", "fused")):
                e0.record()
                for _ in range(reps):
                    call(ps[name])
                e1.record()
                e1.synchronize()
                t[name].append(1e3 * e0.elapsed_time(e1) / reps)
        r = ps["split"].result(st)
        row = {"op": op, "kind": kind, "MiB": mib, "out_per_KiB": round(1024.0 * (r[0] if op == "stage1" else r[2]) / L, 1), "fused_us": round(statistics.median(t["fused"]), 1),
               "split_us": round(statistics.median(t["split"]), 1)}
        row["winner"] = "fused" if row["fused_us"] < row["split_us"] else "split"
        print(json.dumps(row), flush=True)
        for p in ps.values():
            p.close()
        del buf, idx
