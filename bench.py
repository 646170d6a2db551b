#!/usr/bin/env python3
"""bench.py -- stage-1 structural indexing throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--op stage1|minify|validate_utf8]
                    [--workload large_random|amazon_ndjson|twitter_like|deep_nesting|escape_heavy] [--size BYTES]

A "step" = one pass of the hot path (sjgpu_*_device through the C-ABI) over one synthetic buffer that
is already resident in HBM.  N = 1: BASELINE.json configs[1] -- 1 GiB large_random-style JSON.
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank scans its OWN 1 GiB shard
(independent documents, SURVEY 8(e): no data-path collective), weak scaling; value = bytes all ranks
scanned / max-over-ranks time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 measured copy
KERNELS = {("stage1", "split"): ["k_stage1_summarize", "k_resolve_segments", "k_stage1_emit"],
           ("minify", "split"): ["k_minify_summarize", "k_resolve_segments", "k_minify_emit"],
           ("stage1", "fused"): ["k_fused<0>"], ("minify", "fused"): ["k_fused<1>"],
           ("validate_utf8", "split"): ["k_validate_utf8"], ("validate_utf8", "fused"): ["k_validate_utf8"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--op", default="stage1", choices=["stage1", "minify", "validate_utf8"])
    ap.add_argument("--workload", default="large_random", choices=["large_random", "amazon_ndjson", "twitter_like", "deep_nesting", "escape_heavy"])
    ap.add_argument("--size", type=int, default=1 << 30, help="bytes per GPU")
    ap.add_argument("--pipeline", default=os.environ.get("SJGPU_PIPELINE", "auto"), choices=["auto", "fused", "split"])
    ap.add_argument("--ndjson-leg", type=int, default=-1, help="1: also time config 4 (amazon NDJSON shard per GPU, "
                    "with and without the RCCL index concatenation); default: on when --gpus > 1")
    ap.add_argument("--cpu-iters", type=int, default=12)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--share-device", action="store_true", help="dry run: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from simdjson_amd import build, capi, corpus

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    if rank == 0:
        build.build_corpus()
        build.build_sjgpu()
    if world > 1:
        dist.barrier()

    # ---- workload: one synthetic buffer per rank, resident in HBM before anything is timed ----
    gen = {"deep_nesting": corpus.deep_nesting_doc}.get(args.workload) or getattr(corpus, args.workload)
    host, units = gen(args.size, 1000 + rank)
    L = len(host)
    parser = capi.DomParserImplementation(L, device=local_rank)
    parser.set_pipeline(args.pipeline)
    auto = args.pipeline == "auto"
    args.requested_pipeline = args.pipeline
    buf = torch.from_numpy(host).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    if args.op == "stage1":
        out = torch.empty(L + 3, dtype=torch.int32, device="cuda")
        step = lambda: parser.stage1_device(buf.data_ptr(), L, out.data_ptr(), L + 3, stream)
    elif args.op == "minify":
        out = torch.empty(L + 16, dtype=torch.uint8, device="cuda")
        step = lambda: parser.minify_device(buf.data_ptr(), L, out.data_ptr(), stream)
    else:
        out = None
        step = lambda: parser.validate_utf8_device(buf.data_ptr(), L, stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        assert step() == 0
    n, flags, out_len = parser.result(stream)
    err = capi.stage1_error_from_flags(n, flags) if args.op == "stage1" else (15 if flags & 1 else 0)
    if args.op == "validate_utf8":
        err = 11 if flags & capi.F_UTF8_ERROR else 0
    if flags & capi.F_INTERNAL:
        raise SystemExit(f"rank {rank}: single-pass pipeline reported SJGPU_F_INTERNAL")
    if err != 0:
        raise SystemExit(f"rank {rank}: {args.op} returned error_code {err} on the synthetic buffer")

    if auto:  # what AUTO settled on after the warm-up scans (size and, for stage 1, the density it saw)
        assert step() == 0
        parser.result(stream)
        args.pipeline = "fused" if args.op == "validate_utf8" else parser.last_pipeline()
    parser.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ms_sum, calls = parser.profile_read()
    parser.profile_enable(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(L)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bytes = float(tot.item())
    else:
        total_bytes = float(L)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_bytes * args.steps / dt / 1e9
        # roofline: algorithmic bytes per launch over the GPU time of the kernels of one
        # step, measured with HIP events on the launch stream inside the timed region
        if args.op == "stage1":
            alg = L + 4 * (n + 3)
        elif args.op == "minify":
            alg = L + out_len
        else:
            alg = L
        knames = KERNELS[(args.op, args.pipeline)]
        if args.pipeline == "fused" and args.op != "validate_utf8":  # which single-pass kernel this size gets (sjgpu_fused.hip)
            knames = [("k_fused_pipelined" if L > (8 << 20) else "k_fused_16KiB_tiles") + ("<0>" if args.op == "stage1" else "<1>")]
        kms = [m / max(calls, 1) for m in ms_sum][: len(knames)]
        gpu_ms = sum(kms)
        achieved = alg / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(f"{args.op}:{args.workload}:{args.size}:{args.pipeline}")
        line = {
            "metric": "stage1 GB/s (structural indexing)" if args.op == "stage1" else f"{args.op} GB/s",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload} {L} B per GPU (seed 1000+rank), op={args.op}, regular mode, "
                                   f"device-resident input and output, {args.pipeline} pipeline", "bytes_per_gpu": L, "units_per_gpu": units,
                       "structurals": n if args.op == "stage1" else None, "out_bytes": out_len if args.op == "minify" else None,
                       "parallelism": f"{world} independent shard(s), one rank per GPU, no data-path collective",
                       "library": os.path.basename(capi._paths.LIB_SJGPU)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg, "gpu_ms_per_step": round(gpu_ms, 4),
                         "kernels_ms": dict(zip(knames, [round(x, 4) for x in kms])),
                         "timing": "hipEvent pairs around each kernel on the launch stream, averaged over the timed steps"},
        }
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import cpu_baseline  # test infrastructure: the CPU side of the comparison only
            cb = cpu_baseline.time_cpu(host, args.op, args.cpu_iters)
            if args.op == "stage1" and cb["n"] is not None and cb["n"] != n:
                raise SystemExit(f"PARITY FAILURE: GPU n={n} vs CPU reference n={cb['n']}")
            line["cpu_baseline"] = {"value": round(cb["value"], 3), "unit": "GB/s", "cores": cb["cores"], "kind": cb["kind"],
                                    "sample": f"the same {L}-byte buffer, {cb['impl']} kernel, 1 thread, best of {args.cpu_iters}"}
    # ---- optional second leg: BASELINE.json config 4 (parse_many-style NDJSON shards, RCCL concatenation) ----
    ndjson = None
    if (args.ndjson_leg == 1 or (args.ndjson_leg < 0 and world > 1)) and args.op == "stage1":
        try:
            ndjson = ndjson_leg(args, torch, dist, corpus, capi, rank, world, local_rank, fence)
        except Exception as e:  # the primary line must survive a failure here
            ndjson = {"error": repr(e)[:300]}
    # ---- optional third leg (N > 1 only): ONE document cut into one shard per GPU (general inputs) ----
    docshards = None
    if world > 1 and args.op == "stage1" and args.ndjson_leg != 0:
        try:
            docshards = document_leg(args, torch, dist, capi, rank, world, buf, L, out, fence)
        except Exception as e:
            docshards = {"error": repr(e)[:300]}
    if rank == 0:
        if ndjson is not None:
            line["config4_ndjson"] = ndjson
        if docshards is not None:
            line["one_document_shards"] = docshards
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    parser.close()


def document_leg(args, torch, dist, capi, rank, world, buf, L, idx, fence):
    """Every rank's resident buffer taken as its shard of ONE document of world x --size bytes (the large_random
    buffers end in a newline, so the cuts between them are clean cuts): string-parity pre-pass, all_gather of one
    integer per rank, shard scan with the carried in-string bit (include/sjgpu.h, "one large document sharded
    across GPUs").  Timed like the primary leg: barrier + synchronize, MAX over ranks."""
    import time as _t
    dev = buf.device
    parser = capi.DomParserImplementation(L, device=dev.index or 0)
    stream = torch.cuda.current_stream().cuda_stream
    bits = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(world)]
    mine = torch.zeros(1, dtype=torch.int32, device=dev)

    def step():
        mine[0] = parser.string_parity_device(buf.data_ptr(), L, stream)
        dist.all_gather(bits, mine)
        carry = 0
        for r in range(rank):
            carry ^= int(bits[r]) & 1
        parser.stage1_shard_device(buf.data_ptr(), L, carry, idx.data_ptr(), L + 3, stream)
        return parser.result(stream)

    for _ in range(2):
        n, flags, _ = step()
    steps = max(4, args.steps // 2)
    fence()
    t0 = _t.perf_counter()
    for _ in range(steps):
        n, flags, _ = step()
    fence()
    dt = _t.perf_counter() - t0
    t = torch.tensor([dt, float(L), float(n)], dtype=torch.float64, device=dev)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(t)
    parser.close()
    return {"workload": f"one document of {int(t[1])} B in {world} shards (clean cuts), parity pre-pass + all_gather of {world} integers + shard scan",
            "value_GBps": round(float(t[1]) * steps / float(tmax[0]) / 1e9, 2), "total_structurals": int(t[2]),
            "flags_rank0": flags, "steps": steps}


def ndjson_leg(args, torch, dist, corpus, capi, rank, world, local_rank, fence):
    """amazon_cellphones-style NDJSON, one newline-aligned shard of --size bytes per GPU (weak scaling):
    per-rank stage 1 with zero carry-in; then the same plus the all_gather that concatenates the index
    arrays into global 64-bit positions (simdjson_amd.sharded)."""
    import time as _t
    from simdjson_amd import sharded
    host, lines = corpus.amazon_ndjson(args.size, 2000 + rank)  # each rank's slice of the stream (ends in '\n')
    L = len(host)
    scanner = sharded.GpuShardScanner(L, local_rank)
    scanner.parser.set_pipeline(args.requested_pipeline)  # AUTO learns the density from the warm-up scans
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 3, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    step = lambda: scanner.parser.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream)
    for _ in range(2):
        step()
    n, flags, _ = scanner.parser.result(stream)
    steps = max(4, args.steps // 2)
    fence()
    t0 = _t.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt_scan = _t.perf_counter() - t0
    out = {"workload": f"amazon_ndjson {L} B per GPU, newline-aligned shards, zero carry-in", "structurals_rank0": n,
           "lines_rank0": lines, "flags_rank0": flags}
    if world > 1:
        base = torch.tensor([L], dtype=torch.int64, device="cuda")
        sizes = [torch.empty_like(base) for _ in range(world)]
        dist.all_gather(sizes, base)
        my_base = sum(int(x) for x in sizes[:rank])
        local = sharded.ShardScan(my_base, L, n, flags, idx)
        sharded.gather_global_indices(local)  # warm the communicator
        fence()
        t0 = _t.perf_counter()
        for _ in range(steps):
            step()
            n2, f2, _ = scanner.parser.result(stream)
            pos, counts, _ = sharded.gather_global_indices(sharded.ShardScan(my_base, L, n2, f2, idx))
        fence()
        dt_cat = _t.perf_counter() - t0
        t = torch.tensor([dt_scan, dt_cat], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_scan, dt_cat = float(t[0]), float(t[1])
        tot = torch.tensor([float(L)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot)
        total = float(tot)
        out["with_index_concat_GBps"] = round(total * steps / dt_cat / 1e9, 2)
        out["total_structurals"] = int(sum(counts))
        out["sorted_global_positions"] = bool((pos[1:] > pos[:-1]).all())
    else:
        total = float(L)
    out["value_GBps"] = round(total * steps / dt_scan / 1e9, 2)
    out["steps"] = steps
    scanner.parser.close()
    if world == 1 and not args.no_cpu_baseline:
        # the same NDJSON on T independent host threads, each with its own reference parser
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cpu_baseline
        threads = os.cpu_count() or 1
        cb = cpu_baseline.time_cpu_ndjson_threads(host, threads, max(2, args.cpu_iters // 4))
        if cb is not None:
            out["cpu_baseline_threads"] = {"value": round(cb["value"], 2), "unit": "GB/s", "cores": cb["cores"], "kind": cb["kind"],
                                           "sample": f"the same {L}-byte NDJSON buffer cut at newlines into {cb['cores']} slices, "
                                                     f"{cb['impl']} kernel, one parser per thread, {threads} hardware threads on the box",
                                           "structurals": cb["n"]}
    return out


if __name__ == "__main__":
    main()
