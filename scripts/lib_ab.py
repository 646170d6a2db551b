"""A/B of two builds of libsjgpu.so on ONE box (box-to-box variation is +-5 %: figures of different sessions do not compare): every variant in its own process
(SJGPU_LIB names the library, extra NAME=VALUE pairs go into the environment), alternating, the best of four trials of fifteen calls each per workload,
a digest of what was written so that a faster kernel that writes something else is found out here.
    python scripts/lib_ab.py base=build/ab/libsjgpu_base.so new=simdjson_amd/lib/libsjgpu.so [new1=simdjson_amd/lib/libsjgpu.so,SJGPU_EMIT_WAVES=1 ...] [--rounds 2] [--size BYTES]"""
import json
import os
import subprocess
import sys
import time


def child(size):
    sys.path.insert(0, os.getcwd())
    import torch
    from simdjson_amd import capi, corpus
    out = {"variant": os.environ.get("LIB_AB_NAME")}
    st = torch.cuda.current_stream().cuda_stream

    def best(call, trials=4, reps=15):
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        dt = 1e9
        for _ in range(trials):
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / reps)
        return round(dt * 1e6, 1)

    jobs = (("large_random", "fused"), ("large_random", "split"), ("amazon_ndjson", "split"), ("amazon_ndjson", "fused"), ("twitter_like", "split"),
            ("escape_heavy", "split"), ("deep_nesting_doc", "fused"))
    if os.environ.get("LIB_AB_QUICK"):
        jobs = (("large_random", "fused"), ("amazon_ndjson", "split"), ("amazon_ndjson", "fused"), ("escape_heavy", "split"))
    made = {}
    for kind, pipe in jobs:
        if kind not in made:
            made.clear()
            made[kind] = getattr(corpus, kind)(size, 1000)[0]
        a = made[kind]
        L = len(a)
        p = capi.DomParserImplementation(L)
        p.set_pipeline(pipe)
        buf = torch.from_numpy(a).cuda()
        idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        key = f"stage1:{kind}:{pipe}"
        out[key + ":us"] = best(lambda: p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st))
        n, flags, _ = p.result(st)
        p.profile_enable(True)  # HIP events around the kernels of a call: [scan / summarize, resolve, emit] in us per call
        for _ in range(12):
            p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st)
        torch.cuda.synchronize()
        ms, calls = p.profile_read()
        out[key + ":slots_us"] = [round(1e3 * x / max(calls, 1), 1) for x in ms]
        p.profile_enable(False)
        v = idx[:n].to(torch.int64)
        out[key + ":digest"] = [int(n), int(flags), int((v * torch.arange(1, n + 1, device="cuda", dtype=torch.int64)).sum().item() & ((1 << 62) - 1))]
        del v
        if kind == "large_random" and pipe == "fused":
            dst = torch.empty(L + 64, dtype=torch.uint8, device="cuda")
            p.set_pipeline("auto")
            out["minify:large_random:us"] = best(lambda: p.minify_device(buf.data_ptr(), L, dst.data_ptr(), st))
            _, mflags, out_len = p.result(st)
            out["minify:large_random:digest"] = [int(out_len), int(mflags), int(dst[:out_len].to(torch.int64).sum().item())]
            out["validate_utf8:large_random:us"] = best(lambda: p.validate_utf8_device(buf.data_ptr(), L, st))
            out["validate_utf8:large_random:flags"] = int(p.result(st)[1])
            del dst
        if kind == "twitter_like" and not os.environ.get("LIB_AB_QUICK"):  # the string pass and the tape read the same planes
            m = 256 << 20
            a2 = corpus.twitter_like(m, 1000)[0]
            L2 = len(a2)
            b2 = torch.from_numpy(a2).cuda()
            q = capi.DomParserImplementation(L2)
            i2 = torch.empty(L2 + 16, dtype=torch.int32, device="cuda")
            q.stage1_device(b2.data_ptr(), L2, i2.data_ptr(), L2 + 3, st)
            n2 = q.result(st)[0]
            tape = torch.empty(L2 + 8, dtype=torch.int64, device="cuda")
            scap = 5 * (L2 // 3) + 256
            sb = torch.empty(scap, dtype=torch.uint8, device="cuda")
            got = {}

            def stage2():
                got["v"] = q.stage2_device(b2.data_ptr(), L2, i2.data_ptr(), n2, tape.data_ptr(), L2 + 8, sb.data_ptr(), scap, 1024, st)
            out["stage2:twitter_like_256MiB:us"] = best(stage2, 3, 8)
            err, tw, sbn = got["v"]
            out["stage2:twitter_like_256MiB:digest"] = [int(err), int(tw), int(sbn), int(tape[:tw].sum().item() & ((1 << 62) - 1)), int(sb[:sbn].to(torch.int64).sum().item())]
            q.close()
            del b2, i2, tape, sb
        p.close()
        del buf, idx
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(int(args[1]))
        sys.exit(0)
    rounds, size, variants = 2, 1 << 30, []
    i = 0
    while i < len(args):
        if args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        elif args[i] == "--size":
            size = int(args[i + 1]); i += 2
        else:
            name, spec = args[i].split("=", 1)
            parts = spec.split(",")
            variants.append((name, parts[0], dict(kv.split("=", 1) for kv in parts[1:])))
            i += 1
    for _ in range(rounds):
        for name, lib, env in variants:
            e = dict(os.environ, SJGPU_LIB=os.path.abspath(lib), LIB_AB_NAME=name, **env)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(size)], env=e, timeout=900)
