"""What the per-call HIP events of libsjgpu's profile cost: the same 15 back-to-back device-resident stage-1 / minify calls with the profile off and on
(wall time per call; with it on also the event-pair time of slot 0), one process."""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from simdjson_amd import capi, corpus
out = {}
for op, kind, gen in (("stage1", "large_random", corpus.large_random), ("minify", "large_random", corpus.large_random), ("stage1", "amazon_ndjson", corpus.amazon_ndjson)):
    a, _ = gen(1 << 30, 1000)
    L = len(a)
    p = capi.DomParserImplementation(L)
    if op == "stage1": p.set_pipeline("fused")
    buf = torch.from_numpy(a).cuda()
    dst = torch.empty((L + 16) * (4 if op == "stage1" else 1), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    step = (lambda: p.stage1_device(buf.data_ptr(), L, dst.data_ptr(), L + 3, st)) if op == "stage1" else (lambda: p.minify_device(buf.data_ptr(), L, dst.data_ptr(), st))
    for _ in range(3): step()
    torch.cuda.synchronize()
    row = {}
    for prof in (False, True, False, True):
        p.profile_enable(prof)
        best = 1e9
        for _trial in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(15): step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 15)
            if prof:
                ms, calls = p.profile_read()
                row.setdefault("event_ms_slot0", []).append(round(ms[0] / max(calls, 1), 4))
        row.setdefault("wall_us_profile_" + ("on" if prof else "off"), []).append(round(best * 1e6, 1))
    p.profile_enable(False)
    out[f"{op}:{kind}"] = row
    p.close(); del buf, dst
print(json.dumps(out))
