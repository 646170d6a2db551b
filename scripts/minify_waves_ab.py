"""A/B: k_minify_onchip with 4 waves per workgroup (32 KiB tiles, default), 8 (64 KiB) and 16 (128 KiB, one workgroup per CU): SJGPU_MINIFY_WAVES; each
variant in its own process; a digest of the output says that all write the same."""
import os, sys, time, json, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.getcwd())
    import torch
    from simdjson_amd import capi, corpus
    out = {"minify_waves": os.environ.get("SJGPU_MINIFY_WAVES", "4")}
    for kind, gen, size in (("large_random", corpus.large_random, 256 << 20), ("large_random", corpus.large_random, 1 << 30), ("twitter_like", corpus.twitter_like, 1 << 30)):
        a, _ = gen(size, 1000)
        L = len(a)
        p = capi.DomParserImplementation(L)
        buf = torch.from_numpy(a).cuda(); dst = torch.empty(L + 64, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): p.minify_device(buf.data_ptr(), L, dst.data_ptr(), st)
        torch.cuda.synchronize()
        r = p.result(st)
        dt = 1e9
        for _trial in range(4):
            t0 = time.perf_counter()
            for _ in range(15): p.minify_device(buf.data_ptr(), L, dst.data_ptr(), st)
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / 15)
        out[f"{kind}_{size >> 20}MiB_us"] = round(dt * 1e6, 1)
        n = int(r[2]) if len(r) > 2 else 0
        v = dst[:n].to(torch.int64)
        out[f"{kind}_{size >> 20}MiB_digest"] = [n, int((v * (torch.arange(n, device="cuda", dtype=torch.int64) % 251 + 1)).sum().item())]
        out["kernel"] = p.profile_kernel()
        p.close(); del buf, dst, v
    print(json.dumps(out), flush=True)
else:
    for w in ("4", "8", "16", "4", "8", "16"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, SJGPU_MINIFY_WAVES=w))
