"""A/B: the pipelined single-pass kernel with four waves per workgroup (64 KiB tiles, default) and with eight (SJGPU_PIPE_WAVES=8: 128 KiB tiles, two
workgroups per CU); each variant in its own process; a digest of the list says that both write the same."""
import os, sys, time, json, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.getcwd())
    import torch
    from simdjson_amd import capi, corpus
    out = {"pipe_waves": os.environ.get("SJGPU_PIPE_WAVES", "4")}
    for kind, gen, size in (("large_random", corpus.large_random, 256 << 20), ("large_random", corpus.large_random, 1 << 30), ("amazon_ndjson", corpus.amazon_ndjson, 1 << 30),
                            ("twitter_like", corpus.twitter_like, 1 << 30)):
        a, _ = gen(size, 1000)
        L = len(a)
        p = capi.DomParserImplementation(L)
        p.set_pipeline("fused")
        buf = torch.from_numpy(a).cuda(); idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
        torch.cuda.synchronize()
        n, flags, _ = p.result(st)
        dt = 1e9
        for _trial in range(4):
            t0 = time.perf_counter()
            for _ in range(15): p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / 15)
        out[f"{kind}_{size >> 20}MiB_us"] = round(dt * 1e6, 1)
        v = idx[:n].to(torch.int64)
        out[f"{kind}_{size >> 20}MiB_digest"] = [int(n), int(flags), int((v * torch.arange(1, n + 1, device="cuda", dtype=torch.int64)).sum().item() & ((1 << 62) - 1))]
        out["kernel"] = p.profile_kernel()
        p.close(); del buf, idx, v
    print(json.dumps(out), flush=True)
else:
    for w in ("4", "8", "4", "8"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, SJGPU_PIPE_WAVES=w))
