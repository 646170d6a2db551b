"""A/B of builds of libsjgpu.so on the TAPE (sjgpu_stage2_device) inside one process, calls interleaved like scripts/lib_ab.py: every variant builds the tape of
the SAME resident document from the SAME structural list, round after round in turn; per variant the median and the best of the per-round means (HIP events
around `reps` calls, host round trip of the call's result included -- the call synchronises), and digests of the tape and the string buffer.
    python scripts/tape_ab.py base=build/ab/libsjgpu_base.so new=simdjson_amd/lib/libsjgpu.so [...] [--rounds 8] [--reps 5] [--size BYTES] [--kinds a,b]"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from lib_ab import capi_for


def main():
    args = sys.argv[1:]
    rounds, reps, size, kinds, variants = 8, 5, 256 << 20, ["twitter_like", "large_random"], []
    i = 0
    while i < len(args):
        if args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        elif args[i] == "--reps":
            reps = int(args[i + 1]); i += 2
        elif args[i] == "--size":
            size = int(args[i + 1]); i += 2
        elif args[i] == "--kinds":
            kinds = args[i + 1].split(","); i += 2
        else:
            name, path = args[i].split("=", 1)
            variants.append((name, path)); i += 1
    import torch
    from simdjson_amd import corpus
    mods = {name: capi_for(lib, name) for name, lib in variants}
    names = [n for n, _ in variants]
    st = torch.cuda.current_stream().cuda_stream
    table = {}
    for kind in kinds:
        host, _ = getattr(corpus, kind)(size, 3000)
        L = len(host)
        buf = torch.from_numpy(host).cuda()
        idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        tape = torch.empty(L + 8, dtype=torch.int64, device="cuda")
        scap = 5 * (L // 3) + 256
        sbuf = torch.empty(scap, dtype=torch.uint8, device="cuda")
        ps = {n: mods[n].DomParserImplementation(L) for n in names}
        assert ps[names[0]].stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, st) == 0
        n_tok, flags, _ = ps[names[0]].result(st)

        def call(p):
            return p.stage2_device(buf.data_ptr(), L, idx.data_ptr(), n_tok, tape.data_ptr(), L + 8, sbuf.data_ptr(), scap, 1024, st)
        digests, times = {}, {n: [] for n in names}
        for n in names:
            tape.zero_(); sbuf.zero_()
            for _ in range(2):
                err, tw, sb = call(ps[n])
            w = torch.arange(1, tw + 1, device="cuda", dtype=torch.int64)
            digests[n] = [err, tw, sb, int((tape[:tw] * w).sum().item()), int(sbuf[:sb].to(torch.int64).sum().item())]
            del w
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rnd in range(rounds):
            order = names if rnd % 2 == 0 else names[::-1]
            for n in order:
                e0.record()
                for _ in range(reps):
                    call(ps[n])
                e1.record()
                e1.synchronize()
                times[n].append(1e3 * e0.elapsed_time(e1) / reps)
        table[kind] = {"median_us": {n: round(statistics.median(times[n]), 1) for n in names}, "best_us": {n: round(min(times[n]), 1) for n in names},
                       "digests_equal": len({json.dumps(d) for d in digests.values()}) == 1, "digest": digests[names[0]], "bytes": L, "tokens": n_tok}
        print(json.dumps({kind: table[kind]}), flush=True)
        for p in ps.values():
            p.close()
        del buf, idx, tape, sbuf
        torch.cuda.empty_cache()
    print("%-28s" % "tape: median us per call", *["%11s" % n[:11] for n in names])
    for kind, t in table.items():
        print("%-28s" % kind, *["%11.1f" % t["median_us"][n] for n in names], "" if t["digests_equal"] else "  DIGESTS DIFFER")
    print("%-28s" % "best round", *["%11s" % n[:11] for n in names])
    for kind, t in table.items():
        print("%-28s" % kind, *["%11.1f" % t["best_us"][n] for n in names])


if __name__ == "__main__":
    main()
