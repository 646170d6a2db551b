import os, sys, time, json, subprocess
# each grid in its own process (the variable is read when a context is made; contexts are pooled)
if len(sys.argv) > 1:
    sys.path.insert(0, os.getcwd())
    import torch
    from simdjson_amd import capi, corpus
    out = {"grid": os.environ.get("SJGPU_MAX_WORKGROUPS")}
    for size in (512 << 20, 1 << 30):
        a, _ = corpus.large_random(size, 1000)
        L = len(a)
        p = capi.DomParserImplementation(L)
        p.set_pipeline("fused")
        buf = torch.from_numpy(a).cuda(); idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
        torch.cuda.synchronize()
        dt = 1e9
        for _trial in range(4):
            t0 = time.perf_counter()
            for _ in range(15): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / 15)
        out[str(size >> 20) + "MiB_us"] = round(dt * 1e6, 1)
        p.close(); del buf, idx
    print(json.dumps(out), flush=True)
else:
    for g in ("2048", "1024", "1280", "1536", "4096", "8192", "2048"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, SJGPU_MAX_WORKGROUPS=g))
