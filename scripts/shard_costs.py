"""What the one-document sharding costs per rank: the read-only string-parity pre-pass and the shard scan, 1 GiB / G."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simdjson_amd import capi, corpus, sharded

for kind in ("large_random", "twitter_like"):
    a, _ = getattr(corpus, kind)(1 << 30, 5)
    st = torch.cuda.current_stream().cuda_stream
    for parts in (1, 8):
        cuts = sharded.clean_cuts(a, parts)
        lo, hi = cuts[0], cuts[1]
        L = hi - lo
        buf = torch.from_numpy(a[lo:hi].copy()).cuda()
        idx = torch.empty(L + 3, dtype=torch.int32, device="cuda")
        p = capi.DomParserImplementation(L)
        def timed(fn, reps=20):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
        t_par = timed(lambda: p.string_parity_device(buf.data_ptr(), L, st))
        t_scan = timed(lambda: (p.stage1_shard_device(buf.data_ptr(), L, 0, idx.data_ptr(), L + 3, st), p.result(st)))
        t_cut = time.perf_counter(); sharded.clean_cuts(a, 8); t_cut = time.perf_counter() - t_cut
        print(json.dumps({"kind": kind, "shards": parts, "shard_bytes": L, "parity_pass_us": round(t_par * 1e6, 1),
                          "parity_pass_GBps": round(L / t_par / 1e9, 1), "shard_scan_us": round(t_scan * 1e6, 1),
                          "shard_scan_GBps": round(L / t_scan / 1e9, 1), "clean_cuts_host_us": round(t_cut * 1e6, 1)}), flush=True)
        p.close()
