"""A/B of builds of libsjgpu.so INSIDE ONE PROCESS, calls interleaved (box-to-box variation is +-5 %, and on one box the variant that runs first after an idle
phase gets other clocks than the one that runs last: processes that take turns -- the first version of this script -- showed 2-3 % between two copies of the
same kernel).  Every library is loaded under its own copy of simdjson_amd.capi (ctypes opens each file separately), all variants scan the SAME device buffer
into the SAME list, round after round in turn; per variant the median and the best of the per-round means (HIP events of the launch stream around `reps`
calls), the kernels' own event slots [scan or summarize, resolve, emit], and a digest of what was written, so that a faster kernel that writes something else
is found out here.
    python scripts/lib_ab.py base=build/ab/libsjgpu_base.so new=simdjson_amd/lib/libsjgpu.so[,ENV=VALUE...] [...] [--rounds 12] [--reps 10] [--size BYTES] [--quick]"""
import importlib.util
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def capi_for(path, name):
    """a private copy of simdjson_amd.capi bound to the library at `path`"""
    import simdjson_amd
    from simdjson_amd import _paths
    spec = importlib.util.spec_from_file_location("simdjson_amd.capi_" + name, os.path.join(os.path.dirname(simdjson_amd.__file__), "capi.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    keep = _paths.LIB_SJGPU
    _paths.LIB_SJGPU = os.path.abspath(path)
    try:
        mod.load_library()
    finally:
        _paths.LIB_SJGPU = keep
    return mod


def main():
    args = sys.argv[1:]
    rounds, reps, size, quick, variants = 12, 10, 1 << 30, False, []
    i = 0
    while i < len(args):
        if args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        elif args[i] == "--reps":
            reps = int(args[i + 1]); i += 2
        elif args[i] == "--size":
            size = int(args[i + 1]); i += 2
        elif args[i] == "--quick":
            quick = True; i += 1
        else:
            name, spec = args[i].split("=", 1)  # name=path[,ENV=VALUE...]: the environment a library's A/B switches read at their FIRST call
            parts = spec.split(",")
            variants.append((name, parts[0], dict(kv.split("=", 1) for kv in parts[1:]))); i += 1
    import torch
    from simdjson_amd import corpus
    mods = {name: capi_for(lib, name) for name, lib, _ in variants}
    names = [n for n, _, _ in variants]
    envs = {n: e for n, _, e in variants}
    st = torch.cuda.current_stream().cuda_stream
    jobs = [("large_random", "fused", "stage1"), ("large_random", "auto", "minify"), ("large_random", "auto", "validate_utf8"), ("large_random", "split", "stage1"),
            ("amazon_ndjson", "split", "stage1"), ("amazon_ndjson", "fused", "stage1"), ("twitter_like", "split", "stage1"), ("twitter_like", "fused", "stage1"),
            ("escape_heavy", "split", "stage1"), ("escape_heavy", "fused", "stage1"), ("deep_nesting_doc", "fused", "stage1")]
    if os.environ.get("LIB_AB_JOBS"):  # kind:pipeline:op,... -- a session's own selection
        jobs = [tuple(j.split(":")) for j in os.environ["LIB_AB_JOBS"].split(",")]
    if quick:
        jobs = [j for j in jobs if j[0] in ("large_random", "amazon_ndjson", "escape_heavy") and not (j[0] == "large_random" and j[1] == "split")]
    table, made = {}, {}
    for kind, pipe, op in jobs:
        if kind not in made:
            made.clear()
            a = getattr(corpus, kind)(size, 1000)[0]
            made[kind] = (torch.from_numpy(a).cuda(), len(a))
        buf, L = made[kind]
        out = torch.empty(L + 64, dtype=torch.int32 if op == "stage1" else torch.uint8, device="cuda")
        ps = {}
        for n in names:
            p = mods[n].DomParserImplementation(L)
            p.set_pipeline(pipe)
            ps[n] = p

        def call(p):
            if op == "stage1":
                p.stage1_device(buf.data_ptr(), L, out.data_ptr(), L + 3, st)
            elif op == "minify":
                p.minify_device(buf.data_ptr(), L, out.data_ptr(), st)
            else:
                p.validate_utf8_device(buf.data_ptr(), L, st)
        key = f"{op}:{kind}:{pipe}"
        digests, times, slots = {}, {n: [] for n in names}, {}
        for n in names:  # warm-up, and what the variant writes (the switches of a library are static: read once, at its first call)
            out.zero_()
            keep = {k: os.environ.get(k) for k in envs[n]}
            os.environ.update(envs[n])
            for _ in range(3):
                call(ps[n])
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            r = ps[n].result(st)
            if op == "stage1":
                v = out[:r[0]].to(torch.int64)
                digests[n] = [r[0], r[1], int((v * torch.arange(1, r[0] + 1, device="cuda", dtype=torch.int64)).sum().item() & ((1 << 62) - 1))]
                del v
            elif op == "minify":
                digests[n] = [r[2], r[1], int(out[:r[2]].to(torch.int64).sum().item())]
            else:
                digests[n] = [r[1]]
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
        for n in names:
            ps[n].profile_enable(True)
            for _ in range(reps):
                call(ps[n])
            torch.cuda.synchronize()
            ms, calls = ps[n].profile_read()
            slots[n] = [round(1e3 * x / max(calls, 1), 1) for x in ms]
            ps[n].profile_enable(False)
            ps[n].close()
        table[key] = {"median_us": {n: round(statistics.median(times[n]), 1) for n in names}, "best_us": {n: round(min(times[n]), 1) for n in names},
                      "slots_us": slots, "digests_equal": len({json.dumps(d) for d in digests.values()}) == 1, "digest": digests[names[0]]}
        print(json.dumps({key: table[key]}), flush=True)
        del out
    print("%-34s" % "median us per call", *["%11s" % n[:11] for n in names])
    for key, t in table.items():
        print("%-34s" % key, *["%11.1f" % t["median_us"][n] for n in names], "" if t["digests_equal"] else "  DIGESTS DIFFER")
    print("%-34s" % "best round", *["%11s" % n[:11] for n in names])
    for key, t in table.items():
        print("%-34s" % key, *["%11.1f" % t["best_us"][n] for n in names])
    print("%-34s" % "event slots of the split calls", *["%11s" % n[:11] for n in names])
    for key, t in table.items():
        if ":split" in key:
            for s in range(3):
                print("%-34s" % (key + " slot %d" % s), *["%11.1f" % t["slots_us"][n][s] for n in names])


if __name__ == "__main__":
    main()
