#!/usr/bin/env python3
"""bench.py -- stage-1 structural indexing throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--op stage1|minify|validate_utf8]
                    [--workload large_random|amazon_ndjson|twitter_like|deep_nesting|escape_heavy] [--size BYTES]
                    [--legs all|none|name,name]

A "step" = one pass of the hot path (sjgpu_*_device through the C-ABI) over one synthetic buffer that is already
resident in HBM.  The TOP-LEVEL fields of the one JSON line are BASELINE.json configs[1]: stage 1 on 1 GiB of
large_random-style JSON on one MI355X.  N = 1 additionally carries, under "legs", every other configuration of
BASELINE.json measured the same way (each with its own `roofline` and `cpu_baseline`):

    config0_twitter_json      the reference's real twitter.json: exact index check + device-resident and host-buffer timing
    config2_minify            1 GiB large_random, minify          config2_validate_utf8   the same, validate_utf8
    config3_amazon_ndjson     1 GiB amazon_cellphones-style NDJSON, stage 1 (+ the reference on all host threads)
    config4_deep_nesting      1 GiB of brackets (one offset per byte)   config4_escape_heavy   backslash runs up to 64 KiB + 1
    plugin_host_path          sjgpu_stage1 with HOST buffers (PCIe both ways): 1 GiB document, 1 MB
                              parse_many-sized batches next to the reference's icelake kernel, sjgpu_stage1_many

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank scans its OWN 1 GiB shard (independent
documents, SURVEY 8(e): no data-path collective), weak scaling; value = bytes all ranks scanned / max-over-ranks time;
"config3_ndjson_sharded" / "one_document_shards" carry the NDJSON shards (BASELINE configs[3]) with the index concatenation and the one-document path.

Output (round 6): rank 0 prints ONE JSON line, the LAST line of stdout, at most COMPACT_LINE_LIMIT (4096) characters: the contract's fields with numbers and
short identifiers only (compact_line below) -- the reference's own convention is three figures per stage and run (benchmark/benchmarker.h:408-419).  Everything
else a leg measures (per-slot kernel times, digests, samples, sweeps) goes to the side-car `bench_detail.json` beside this script (and into gpurun_out/ when
that directory exists); what the fields MEAN is DESIGN.md section 5, not the line.
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
ALL_LEGS = ["config0_twitter_json", "config2_minify", "config2_validate_utf8", "config3_amazon_ndjson", "config4_deep_nesting",
            "config4_escape_heavy", "plugin_host_path", "next_f2_finish_device", "next_f3_depth_scan", "next_f3_parse_strings", "next_f3_tape"]

COMPACT_LINE_LIMIT = 4096  # characters of the ONE line rank 0 prints (round 5's 23 KB line outgrew the driver's capture: BENCH_r05.json parsed = null)
DETAIL_FILE = "bench_detail.json"


def _short(x, n=96):
    """identifiers stay, sentences are cut: nothing in the printed line explains itself (DESIGN.md section 5 does)"""
    return x if not isinstance(x, str) or len(x) <= n else x[: n - 1] + "~"


def _roofline_compact(r):
    if not isinstance(r, dict):
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "gpu_ms_per_step", "kernel", "traffic_static_from_profiles")
    return {k: _short(r[k], 64) for k in keep if k in r}


def _cpu_compact(c):
    if not isinstance(c, dict):
        return None
    return {"value": c.get("value"), "unit": _short(c.get("unit"), 24), "cores": c.get("cores"), "kind": c.get("kind"), "sample": _short(c.get("sample", ""), 72)}


def _leg_compact(name, leg):
    """one leg of the detail -> at most a dozen numbers"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return {"error": _short(leg["error"], 80)}
    out = {}
    if "ms_per_step" in leg:  # device_leg: a BASELINE config timed like the headline
        r = leg.get("roofline") or {}
        out = {"value": leg.get("value"), "ms_per_step": leg.get("ms_per_step"), "gpu_ms": r.get("gpu_ms_per_step"), "frac": r.get("frac"),
               "pipeline": leg.get("pipeline"), "parity_ok": (leg.get("parity") or {}).get("ok"), "cpu": (leg.get("cpu_baseline") or {}).get("value")}
        if "cpu_baseline_threads" in leg:
            out["cpu_threads"] = [leg["cpu_baseline_threads"].get("value"), leg["cpu_baseline_threads"].get("cores")]
        return out
    if name == "config0_twitter_json":
        return {"n": leg.get("n"), "exact": leg.get("exact_vs_reference"), "device_us": (leg.get("device_resident") or {}).get("gpu_us_per_call"),
                "host_us": (leg.get("host_buffers") or {}).get("us_per_call"), "cpu": (leg.get("cpu_baseline") or {}).get("value")}
    if name == "plugin_host_path":
        doc, win, small = leg.get("document_1GiB_pageable") or {}, leg.get("parse_many_window_1MB") or {}, leg.get("small_documents") or {}
        return {"doc_1GiB_GBps": doc.get("value"), "window_us": win.get("mi355x_us_per_window"), "window_ref_us": win.get("reference_us_per_window"),
                "small_doc_us": small.get("one_launch_us_per_document")}
    if name == "next_f3_tape":
        for kind, v in leg.items():
            if isinstance(v, dict):
                w = v.get("with_token_stream") or {}
                out[kind] = {"ms": v.get("gpu_ms_per_call"), "frac": (v.get("roofline") or {}).get("frac"), "traffic": (v.get("roofline") or {}).get("traffic"),
                             "ms_from_tokens": w.get("stage2_ms_per_call"), "cpu": (v.get("cpu_baseline") or {}).get("value")}
        return out
    r = leg.get("roofline") or {}
    out = {"ms": leg.get("gpu_ms_per_call", leg.get("ms_per_call")), "value": leg.get("value"), "unit": _short(leg.get("unit"), 24), "frac": r.get("frac")}
    w = leg.get("with_token_stream")
    if isinstance(w, dict):
        out["ms_from_tokens"] = w.get("depth_scan_ms_per_call")
    return out


def compact_line(d):
    """The ONE printed line from the full record: the contract's fields, numbers and short identifiers only, <= COMPACT_LINE_LIMIT characters."""
    c = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = d.get("config") or {}
    c["config"] = {"workload": _short(cfg.get("workload"), 160), "bytes_per_gpu": cfg.get("bytes_per_gpu"), "structurals": cfg.get("structurals"),
                   "library": cfg.get("library")}
    c["roofline"] = _roofline_compact(d.get("roofline"))
    c["cpu_baseline"] = _cpu_compact(d.get("cpu_baseline"))
    if "cpu_baseline_threads" in d:
        c["cpu_baseline_threads"] = _cpu_compact(d["cpu_baseline_threads"])
    p = d.get("parity") or {}
    c["parity"] = {k: p[k] for k in ("checked", "ok", "ranks_checked", "ranks_ok", "all_ranks_ok") if k in p}
    for k in ("first_reps_ms_per_step", "value_first_reps", "clock_warmup_calls", "n1_same_workload_GBps", "scaling_efficiency", "n_ranks_seen_by_rccl"):
        if k in d:
            c[k] = d[k]
    if "index_concat" in d:
        c["index_concat"] = _short(d["index_concat"], 40)
    if "legs" in d:
        c["legs"] = {k: _leg_compact(k, v) for k, v in d["legs"].items()}
        c["legs_failed"] = d.get("legs_failed", [])
    nd = d.get("config3_ndjson_sharded")
    if isinstance(nd, dict):
        c["config3_ndjson_sharded"] = ({"error": _short(nd["error"], 80)} if "error" in nd else
                                       {k: nd.get(k) for k in ("value_GBps", "with_index_concat_GBps", "n1_same_workload_GBps", "scaling_efficiency", "total_structurals",
                                                               "sorted_global_positions", "steps") if k in nd})
        if "cpu_baseline" in nd:
            c["config3_ndjson_sharded"]["cpu"] = nd["cpu_baseline"].get("value")
        if "cpu_baseline_threads" in nd:
            c["config3_ndjson_sharded"]["cpu_threads"] = [nd["cpu_baseline_threads"].get("value"), nd["cpu_baseline_threads"].get("cores")]
    ds = d.get("one_document_shards")
    if isinstance(ds, dict):
        c["one_document_shards"] = {"error": _short(ds["error"], 80)} if "error" in ds else {k: ds.get(k) for k in ("value_GBps", "total_structurals", "steps")}
    c["detail"] = DETAIL_FILE
    text = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(text) > COMPACT_LINE_LIMIT:  # never let the line outgrow the capture again: drop what is least needed, say so
        for victim in ("one_document_shards", "legs"):
            if victim in c:
                c[victim] = {"dropped": "see " + DETAIL_FILE}
                text = json.dumps(c, allow_nan=False, separators=(",", ":"))
                if len(text) <= COMPACT_LINE_LIMIT:
                    break
    assert len(text) <= COMPACT_LINE_LIMIT, len(text)
    return text


def write_detail(d):
    """the full record beside the script (and under gpurun_out/ when that directory exists, so that it comes home from the GPU box)"""
    for where in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(where):
            try:
                with open(os.path.join(where, DETAIL_FILE), "w") as f:
                    json.dump(d, f, indent=1)
            except OSError:
                pass


def position_digest_host(words):
    """Order-sensitive digest of an index list: sum of (i + 1) * idx[i] modulo 2^64 (wraps like the int64 torch computes)."""
    w = np.asarray(words, dtype=np.uint64)
    return int((w * np.arange(1, len(w) + 1, dtype=np.uint64)).sum(dtype=np.uint64))


def position_digest_device(torch, idx, count):
    w = idx[:count].to(torch.int64) & 0xFFFFFFFF
    return int((w * torch.arange(1, count + 1, dtype=torch.int64, device=idx.device)).sum().item()) & 0xFFFFFFFFFFFFFFFF


def byte_digest_host(b):
    w = np.asarray(b, dtype=np.uint64)
    return int((w * (np.arange(1, len(w) + 1, dtype=np.uint64) | np.uint64(1))).sum(dtype=np.uint64))


def byte_digest_device(torch, dst, count):
    w = dst[:count].to(torch.int64)
    return int((w * (torch.arange(1, count + 1, dtype=torch.int64, device=dst.device) | 1)).sum().item()) & 0xFFFFFFFFFFFFFFFF


class Ctx:
    """What every leg needs: torch, the C-ABI mirror, the corpus generators, the reference (CPU side of the comparison)."""

    def __init__(self, args, torch, capi, corpus, local_rank):
        self.args, self.torch, self.capi, self.corpus, self.local_rank = args, torch, capi, corpus, local_rank
        self.traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            self.traffic = json.load(open(tpath))
        self._ref = None

    def cpu(self):
        if self._ref is None:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import cpu_baseline  # test infrastructure: the CPU side of the comparison only
            self._ref = cpu_baseline
        return self._ref


CLOCK_WARMUP_S = 0.04


def clock_warmup(torch, step, budget_s=CLOCK_WARMUP_S):
    """UNTIMED continuous work in front of a timed region, on top of the --warmup steps.  After an idle phase (the reference timed on the host cores, a
    buffer being generated) the first 30-40 calls of a 0.5 ms kernel run 5-15 % slow -- 487 us, rising to 552 by the tenth call, back at 482 from the
    fortieth on; a GPU that has just worked is steady from its first call (scripts/lab/drift_probe.py, profiles/r04_drift_probe.txt).  With 3 + 20 calls the
    rounds 1-4a timed that transient.  Returns the number of calls made; the line says so (clock_warmup_calls)."""
    torch.cuda.synchronize()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s and n < 4096:
        for _ in range(8):
            step()
        n += 8
        torch.cuda.synchronize()
    return n


def event_ms_per_call(torch, run, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


FIRST_REPS_NOTE = ("gpu_ms_per_call: after clock_warmup (40 ms of the same calls, untimed) -- the sustained figure; first_reps_ms_per_call: the same number of calls "
                   "timed straight after one warm-up call, the way rounds 1-4a timed this leg (a GPU fresh from an idle phase runs short sequences at boost clocks)")


def make_workload(corpus, workload, size, seed):
    gen = {"deep_nesting": corpus.deep_nesting_doc}.get(workload) or getattr(corpus, workload)
    return gen(size, seed)


def device_leg(cx, op, workload, host, units, steps, warmup, pipeline, fence=None, cpu_iters=None, with_cpu=True, with_parity=None, parity_raises=True,
               rank=0, world=1):
    """One op over one resident buffer: K timed steps bracketed by synchronize (+ barrier), HIP-event kernel time from
    libsjgpu, exact parity of the timed buffer's output against the reference (count AND an order-sensitive digest).
    Three timed regions, in this order: (1) `steps` steps straight behind the --warmup steps -- what a cold 5 + 20 run gives, the way rounds 1-4a
    timed (first_reps_ms_per_step / value_first_reps); (2) N > 1 only: rank 0 ALONE over its own buffer while the other ranks wait in a barrier --
    the same-workload single-rank figure the N-rank value is divided by (n1_same_workload); (3) behind clock_warmup: the sustained figure = `value`."""
    torch, capi = cx.torch, cx.capi
    L = len(host)
    # every leg starts from an empty caching allocator: the split pipeline's pace depends on WHERE its buffers lie (profiles/r05_stream_ab.txt, session R: up to
    # 7 % between two contexts of one library; this line's NDJSON leg 0.303 against 0.349 ms with and without the temporaries of the parity checks of the
    # legs in front of it in torch's cache) -- a leg must not inherit the blocks an earlier leg's checks left behind
    torch.cuda.empty_cache()
    parser = capi.DomParserImplementation(L, device=cx.local_rank)
    parser.set_pipeline(pipeline)
    buf = torch.from_numpy(host).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    out = None
    if op == "stage1":
        out = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        step = lambda: parser.stage1_device(buf.data_ptr(), L, out.data_ptr(), L + 3, stream)
    elif op == "minify":
        out = torch.empty(L + 64, dtype=torch.uint8, device="cuda")
        step = lambda: parser.minify_device(buf.data_ptr(), L, out.data_ptr(), stream)
    else:
        step = lambda: parser.validate_utf8_device(buf.data_ptr(), L, stream)
    sync = fence or torch.cuda.synchronize
    for _ in range(max(warmup, 1)):
        assert step() == 0
    n, flags, out_len = parser.result(stream)
    if flags & capi.F_INTERNAL:
        raise SystemExit(f"{op}/{workload}: single-pass pipeline reported SJGPU_F_INTERNAL")
    err = capi.stage1_error_from_flags(n, flags) if op == "stage1" else ((15 if flags & 1 else 0) if op == "minify" else (11 if flags & capi.F_UTF8_ERROR else 0))
    if err != 0:
        raise SystemExit(f"{op}/{workload}: error_code {err} on the synthetic buffer")
    assert step() == 0  # AUTO has settled (size and, for stage 1, the density the warm-up scans saw)
    parser.result(stream)
    used = "-" if op == "validate_utf8" else parser.last_pipeline()
    kernel = parser.profile_kernel()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt_first = time.perf_counter() - t0
    solo = None
    if world > 1:  # every rank passes both fences; between them only rank 0 works
        sync()
        if rank == 0:
            clock_warmup(torch, step)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            dt_solo = time.perf_counter() - t0
            solo = {"value": round(L * steps / dt_solo / 1e9, 2), "unit": "GB/s", "ms_per_step": round(dt_solo / steps * 1e3, 4), "steps": steps,
                    "how": "rank 0 alone over its own resident buffer (same workload, same size, same kernels) while every other rank waits in a barrier: "
                           "what an N = 1 run of THIS workload gives, measured inside this run"}
        sync()
    warm_calls = clock_warmup(torch, step)
    parser.profile_enable(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    ms_sum, calls = parser.profile_read()
    parser.profile_enable(False)
    # slot 0 = clears + (first) scan kernel; slots 1, 2 hold the resolve and emit kernels of the split pipeline.
    # A single-pass call leaves them empty, and an empty event pair still measures ~5 us of event bookkeeping: not counted.
    single_kernel = "+" not in kernel
    gpu_ms = (ms_sum[0] if single_kernel else sum(ms_sum)) / max(calls, 1)
    alg = L + (4 * (n + 3) if op == "stage1" else (out_len if op == "minify" else 0))
    achieved = alg / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else 0.0
    tkey = f"{op}:{workload}:{cx.args.size}:{used if used != '-' else 'fused'}"
    leg = {
        "value": round(L * steps / dt / 1e9, 2), "unit": "GB/s", "ms_per_step": round(dt / steps * 1e3, 4), "bytes": L, "units": units,
        "pipeline": used, "structurals": n if op == "stage1" else None, "out_bytes": out_len if op == "minify" else None, "clock_warmup_calls": warm_calls,
        "first_reps_ms_per_step": round(dt_first / steps * 1e3, 4), "value_first_reps": round(L * steps / dt_first / 1e9, 2), "n1_same_workload": solo,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes_per_launch": alg, "gpu_ms_per_step": round(gpu_ms, 4), "kernel": kernel,
                     "kernel_ms_slots": [round(m / max(calls, 1), 4) for m in ms_sum],
                     "traffic": cx.traffic.get(tkey), "traffic_static_from_profiles": cx.traffic.get(tkey), "traffic_measured_in_this_run": False,
                     "traffic_source": "STATIC: read from profiles/traffic.json, not measured by this run -- rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE of this "
                     "command in separate counter passes (scripts/gpu_pmc.sh, profiles/r06_pmc_summary.txt); null = not collected for this workload",
                     "timing": "hipEvent pairs around what one call enqueues on the launch stream (slot 0: clears and the scan kernel; slots 1-2: resolve and emit "
                               "kernels of the split pipeline, empty and not counted for a single-pass call), mean over the timed steps"},
    }
    if with_cpu:
        cb = cx.cpu().time_cpu(host, op, cpu_iters or cx.args.cpu_iters)
        leg["cpu_baseline"] = {"value": round(cb["value"], 3), "unit": "GB/s", "cores": cb["cores"], "kind": cb["kind"],
                               "sample": f"the same {L}-byte buffer, {cb['impl']} kernel, 1 thread, best of {cpu_iters or cx.args.cpu_iters}"}
    if with_cpu if with_parity is None else with_parity:
        try:
            leg["parity"] = parity_check(cx, op, host, parser, n, out_len, flags, out)
        except SystemExit as e:  # N > 1: a rank must not leave while the others wait in a collective -- main() reduces the verdicts and fails then
            if parity_raises:
                raise
            leg["parity"] = {"checked": True, "ok": False, "why": str(e)}
    parser.close()
    del buf, out
    return leg


def parity_check(cx, op, host, parser, n, out_len, flags, out):
    """The output of the TIMED buffer against the real reference: count, error code and an order-sensitive digest of
    every word / byte (the exact full compare lives in tests/test_gpu_parity.py::test_full_size_*)."""
    import ctypes
    torch = cx.torch
    L = len(host)
    cpu = cx.cpu()
    if not os.path.exists(cpu.LIB_REF):
        return {"checked": False, "why": "oracle/_ref/libsjref.so absent"}
    R = ctypes.CDLL(cpu.LIB_REF)
    R.sjref_available.argtypes = [ctypes.c_char_p]
    impl = next((i for i in (b"icelake", b"haswell", b"westmere") if R.sjref_available(i)), None)
    if op == "stage1":
        R.sjref_stage1.restype = ctypes.c_int
        R.sjref_stage1.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        ridx = np.zeros(L + 80, dtype=np.uint32)
        rn = ctypes.c_uint32(0)
        rerr = R.sjref_stage1(impl, host.ctypes.data, L, 0, 0, ridx.ctypes.data, ctypes.byref(rn))
        want = position_digest_host(ridx[: rn.value + 3])
        got = position_digest_device(torch, out, n + 3)
        ok = rerr == 0 and rn.value == n and want == got
        if not ok:
            raise SystemExit(f"PARITY FAILURE stage1: n {n} vs {rn.value}, digest {got} vs {want}, reference error {rerr}")
        return {"checked": True, "ok": True, "n": n, "digest_sum_i_times_idx_mod_2_64": got, "reference": impl.decode()}
    if op == "minify":
        R.sjref_minify.restype = ctypes.c_int
        R.sjref_minify.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        rout = np.zeros(L + 64, dtype=np.uint8)
        rl = ctypes.c_size_t(0)
        rerr = R.sjref_minify(impl, host.ctypes.data, L, rout.ctypes.data, ctypes.byref(rl))
        want = byte_digest_host(rout[: rl.value])
        got = byte_digest_device(torch, out, out_len)
        if not (rerr == 0 and rl.value == out_len and want == got):
            raise SystemExit(f"PARITY FAILURE minify: len {out_len} vs {rl.value}, digest {got} vs {want}")
        return {"checked": True, "ok": True, "out_bytes": out_len, "digest": got, "reference": impl.decode()}
    R.sjref_validate_utf8.restype = ctypes.c_int
    R.sjref_validate_utf8.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    want = R.sjref_validate_utf8(impl, host.ctypes.data, L)
    got = 0 if flags & cx.capi.F_UTF8_ERROR else 1
    if want != got:
        raise SystemExit("PARITY FAILURE validate_utf8")
    return {"checked": True, "ok": True, "verdict": got, "reference": impl.decode()}


def leg_twitter_json(cx):
    """BASELINE.json configs[0]: the reference's twitter.json (fixture), plumbing + bit-exact index check."""
    import ctypes
    torch, capi = cx.torch, cx.capi
    path = os.path.join(ROOT, "tests", "golden", "jsonexamples", "twitter.json")
    host = np.fromfile(path, dtype=np.uint8)
    L = len(host)
    cpu = cx.cpu()
    R = ctypes.CDLL(cpu.LIB_REF)
    R.sjref_available.argtypes = [ctypes.c_char_p]
    impl = next((i for i in (b"icelake", b"haswell", b"westmere") if R.sjref_available(i)), None)
    R.sjref_stage1.restype = ctypes.c_int
    R.sjref_stage1.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    ridx = np.zeros(L + 80, dtype=np.uint32)
    rn = ctypes.c_uint32(0)
    assert R.sjref_stage1(impl, host.ctypes.data, L, 0, 0, ridx.ctypes.data, ctypes.byref(rn)) == 0
    p = capi.DomParserImplementation(1 << 20, device=cx.local_rank)
    err = p.stage1(host, capi.REGULAR)  # host buffers: what dom::parser drives through the plug-in
    n = p.n_structural_indexes
    exact = err == 0 and n == rn.value and np.array_equal(p.structural_indexes[: n + 3], ridx[: n + 3])
    if not exact:
        raise SystemExit(f"PARITY FAILURE twitter.json: err {err}, n {n} vs {rn.value}")
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        p.stage1(host, capi.REGULAR)
    host_us = (time.perf_counter() - t0) / reps * 1e6
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream)
    p.result(stream)
    kernel = p.profile_kernel()
    p.profile_enable(True)
    for _ in range(reps):
        p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream)
    ms, calls = p.profile_read()
    p.profile_enable(False)
    gpu_ms = (ms[0] if "+" not in kernel else sum(ms)) / max(calls, 1)  # empty event slots of a single-pass call are not kernel time
    dn, dflags, _ = p.result(stream)
    assert dn == n and position_digest_device(torch, idx, n + 3) == position_digest_host(ridx[: n + 3])
    cb = cpu.time_cpu(host, "stage1", 200)
    alg = L + 4 * (n + 3)
    p.close()
    return {"workload": "tests/golden/jsonexamples/twitter.json (the reference's file, 631 515 B)", "n": n, "exact_vs_reference": True,
            "known_answer": {"n": 55263, "matches": n == 55263},
            "device_resident": {"gpu_us_per_call": round(gpu_ms * 1e3, 2), "value": round(L / gpu_ms / 1e6, 2), "unit": "GB/s", "kernel": kernel,
                                "roofline": {"bound": "launch latency (10 tiles of 64 KiB)", "achieved": round(alg / gpu_ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": round(alg / gpu_ms / 1e6 / HBM_PEAK_GBS, 5)}},
            "host_buffers": {"us_per_call": round(host_us, 1), "value": round(L / host_us / 1e3, 2), "unit": "GB/s",
                             "note": "sjgpu_stage1: upload, scan, download, finish -- PCIe both ways, what dom::parser::parse sees"},
            "cpu_baseline": {"value": round(cb["value"], 3), "unit": "GB/s", "cores": 1, "kind": cb["kind"],
                             "sample": f"twitter.json, {cb['impl']} kernel, 1 thread, best of 200 ({cb['seconds'] * 1e6:.1f} us per call)"}}


def leg_parse_strings(cx):
    """SURVEY 8(f3), the first stage-2 piece on the device: every string of a 256 MiB twitter-like document unescaped into the
    reference's string-buffer format (sjgpu_parse_strings_device), next to the reference kernel's parse_string over the same list."""
    import ctypes
    torch, capi, corpus = cx.torch, cx.capi, cx.corpus
    host, _ = make_workload(corpus, "twitter_like", 256 << 20, 3000)
    L = len(host)
    p = capi.DomParserImplementation(L, device=cx.local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
    n, flags, _ = p.result(stream)
    assert flags == 0
    cap = 5 * (L + 1) // 3 + 64
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    off = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    run = lambda: p.parse_strings_device(buf.data_ptr(), L, idx.data_ptr(), n, out.data_ptr(), cap, off.data_ptr(), False, stream)
    err, used, strings, bad = run()
    if err != 0:
        raise SystemExit(f"parse_strings: error {err} at structural {bad} on the synthetic document")
    reps = 10
    first_ms = event_ms_per_call(torch, run, reps)
    clock_warmup(torch, run)
    gpu_ms = event_ms_per_call(torch, run, reps)
    leg = {"workload": f"twitter_like {L} B, {n} structurals, {strings} strings -> {used} B of [u32 length][bytes][0] records (document::string_buf of the reference)",
           "gpu_ms_per_call": round(gpu_ms, 3), "first_reps_ms_per_call": round(first_ms, 3), "timing": FIRST_REPS_NOTE, "value": round(L / gpu_ms / 1e6, 1), "unit": "GB/s of document",
           "string_bytes_GBps": round(used / gpu_ms / 1e6, 1),
           "kernel": ("k_strs_count + k_strs_resolve + k_strs_tokens (one bit per token, a count per tile) + k_scan_partials + k_strs_write + k_strs_finalize (stream compaction of the document, sjgpu_string_stream.hip)"
                      if p.string_path() == 1 else "k_strings<false> + scan + k_strings<true> (per-string walk)"),
           "roofline": {"bound": "hbm", "achieved": round((L + 4 * (n + 1) + used + 4 * (n + 1)) / gpu_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round((L + 4 * (n + 1) + used + 4 * (n + 1)) / gpu_ms / 1e6 / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": L + 4 * (n + 1) + used + 4 * (n + 1),
                        "algorithmic_bytes": "document + list in, records + offsets out; all kernels of one call together"},
           "note": "includes the 24-byte result read-back of every call; the list and the buffer stay on the device"}
    cpu = cx.cpu()
    if cpu.LIB_REF and os.path.exists(cpu.LIB_REF):
        R = ctypes.CDLL(cpu.LIB_REF)
        R.sjref_available.argtypes = [ctypes.c_char_p]
        impl = next((i for i in (b"icelake", b"haswell", b"westmere") if R.sjref_available(i)), None)
        if impl:
            R.sjref_bench_parse_strings.restype = ctypes.c_double
            R.sjref_bench_parse_strings.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
            padded = np.concatenate([host, np.full(128, 0x20, np.uint8)])
            hidx = idx[:n].cpu().numpy().view(np.uint32)
            rout = np.empty(cap + 128, dtype=np.uint8)
            rused, rstr = ctypes.c_uint64(0), ctypes.c_uint32(0)
            sec = R.sjref_bench_parse_strings(impl, padded.ctypes.data, L, hidx.ctypes.data, n, rout.ctypes.data, 3, ctypes.byref(rused), ctypes.byref(rstr))
            same = sec > 0 and rused.value == used and rstr.value == strings and bool((out[:used].cpu().numpy() == rout[:used]).all())
            if not same:
                raise SystemExit(f"PARITY FAILURE parse_strings: reference {rused.value} B / {rstr.value} strings vs {used} / {strings}")
            leg["parity"] = "byte for byte the buffer the reference kernel's parse_string leaves for the same list"
            leg["cpu_baseline"] = {"value": round(L / sec / 1e9, 3), "unit": "GB/s of document", "cores": 1, "kind": "reference",
                                   "sample": f"{impl.decode()} kernel, parse_string over the same {strings} strings, 1 thread, best of 3 ({sec * 1e3:.1f} ms)"}
    p.close()
    return leg


def leg_list_passes(cx, which):
    """the passes over the structural LIST on the device -- finish() of a streaming mode (document boundary search,
    sjgpu_stage1_finish_device) and the nesting-depth scan (sjgpu_depth_scan_device) -- timed on the list of 1 GiB of amazon NDJSON, next to
    the host walk that needs the list in host memory first (sjgpu_stage1_finish_host over a downloaded copy)."""
    import ctypes
    torch, capi, corpus = cx.torch, cx.capi, cx.corpus
    host, _ = make_workload(corpus, "amazon_ndjson", cx.args.size, 2000)
    L = len(host)
    p = capi.DomParserImplementation(L, device=cx.local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
    n, flags, _ = p.result(stream)
    keep = idx[: n + 3].clone()
    reps = 10
    if which == "depth":
        depth = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        p.depth_scan_device(buf.data_ptr(), idx.data_ptr(), n, depth.data_ptr(), stream)
        run_depth = lambda: p.depth_scan_device(buf.data_ptr(), idx.data_ptr(), n, depth.data_ptr(), stream)
        first_ms = event_ms_per_call(torch, run_depth, reps)
        clock_warmup(torch, run_depth)
        gpu_ms = event_ms_per_call(torch, run_depth, reps)
        alg = 4 * n + n + 4 * (n + 1)  # list in, one byte of the document per structural, depths out
        final_depth = int(depth[n].item())
        leg = {"workload": f"amazon_ndjson {L} B: {n} structurals -> int32 depth in front of every structural (final depth {final_depth})",
               "gpu_ms_per_call": round(gpu_ms, 4), "first_reps_ms_per_call": round(first_ms, 4), "timing": FIRST_REPS_NOTE, "value": round(n / gpu_ms / 1e6, 2),
               "unit": "G structurals/s", "kernel": "k_depth_codes + k_scan_partials + k_depth_write",
               "roofline": {"bound": "hbm", "achieved": round(alg / gpu_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / gpu_ms / 1e6 / HBM_PEAK_GBS, 4),
                            "algorithmic_bytes_per_launch": alg, "algorithmic_bytes": "4 B of list + 1 B of document in, 4 B of depth out, per structural"}}
        hidx = keep[:n].cpu().numpy().view(np.uint32)
        t0 = time.perf_counter()
        c = host[hidx]
        d = np.cumsum((c == ord("{")).astype(np.int32) + (c == ord("[")) - (c == ord("}")) - (c == ord("]")))
        leg["cpu_baseline"] = {"value": round(n / (time.perf_counter() - t0) / 1e9, 3), "unit": "G structurals/s", "cores": 1, "kind": "port",
                               "sample": "numpy gather + cumsum over the same list on one host core (the reference carries the depth inside its serial stage-2 walk)"}
        assert int(d[-1]) == final_depth
        # round 5: the same pass fed from the token-byte stream stage 1 can write beside the offsets (sjgpu_stage1_tokens_device) -- one coalesced byte per
        # entry instead of a gather through the document's 128-byte lines -- and what the stream costs stage 1 (split pipeline, same buffer)
        tok = torch.empty(L // 2 + 16, dtype=torch.uint8, device="cuda")
        depth2 = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        p.set_pipeline("split")
        run_plain = lambda: p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream)
        run_tok = lambda: p.stage1_tokens_device(buf.data_ptr(), L, idx.data_ptr(), L // 2, tok.data_ptr(), L // 2 + 16, stream)  # (a byte per list word)
        run_tok()
        nt, ft, _ = p.result(stream)
        assert (nt, ft) == (n, flags) and bool(torch.equal(idx[: n + 3], keep))
        run_depth_tok = lambda: p.depth_scan_tokens_device(tok.data_ptr(), n, depth2.data_ptr(), stream)
        run_depth_tok()
        if not bool(torch.equal(depth, depth2)):
            raise SystemExit("PARITY FAILURE: the depth scan from the token stream differs from the depth scan that gathers")
        clock_warmup(torch, run_plain)
        ms_plain = event_ms_per_call(torch, run_plain, reps)
        clock_warmup(torch, run_tok)
        ms_tok = event_ms_per_call(torch, run_tok, reps)
        clock_warmup(torch, run_depth_tok)
        ms_dt = event_ms_per_call(torch, run_depth_tok, reps)
        alg_t = n + 4 * (n + 1)  # one token byte in, 4 B of depth out, per structural
        leg["with_token_stream"] = {
            "depth_scan_ms_per_call": round(ms_dt, 4), "value": round(n / ms_dt / 1e6, 2), "unit": "G structurals/s",
            "roofline": {"bound": "hbm", "achieved": round(alg_t / ms_dt / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_t / ms_dt / 1e6 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes_per_launch": alg_t, "algorithmic_bytes": "1 B of token stream in, 4 B of depth out, per structural",
                         "frac_by_the_gathering_pass_accounting": round(alg / ms_dt / 1e6 / HBM_PEAK_GBS, 4)},
            "stage1_split_ms_without_tokens": round(ms_plain, 4), "stage1_split_ms_with_tokens": round(ms_tok, 4),
            "stage1_cost_of_the_stream": round(ms_tok / ms_plain - 1.0, 4),
            "stage1_plus_depth_scan_ms": {"gathering": round(ms_plain + gpu_ms, 4), "token_stream": round(ms_tok + ms_dt, 4)},
            "parity": "same list and flags as sjgpu_stage1_device; depth array identical to the gathering pass's",
            "note": "sjgpu_stage1_tokens_device + sjgpu_depth_scan_tokens_device: the stream is opt-in (it costs the scan kernel a byte compaction per chunk); "
                    "what a pipeline of stage 1 + one list pass pays in total is the last entry"}
        del tok, depth2
    else:
        mode = capi.STREAMING_FINAL
        err, n_kept, nxt = p.stage1_finish_device(buf.data_ptr(), L, mode, idx.data_ptr(), n, flags, stream)
        times = []
        for _ in range(reps):
            idx[: n + 3].copy_(keep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            err, n_kept, nxt = p.stage1_finish_device(buf.data_ptr(), L, mode, idx.data_ptr(), n, flags, stream)  # waits for its 64-byte state
            times.append(time.perf_counter() - t0)
        gpu_ms = min(times) * 1e3
        alg = 4 * n + n
        leg = {"workload": f"amazon_ndjson {L} B: finish(streaming_final) over {n} structurals on the device -> error {err}, {n_kept} kept",
               "ms_per_call": round(gpu_ms, 4), "value": round(n / gpu_ms / 1e6, 2), "unit": "G structurals/s",
               "kernel": "k_finish_init + k_last_boundary x 2 + k_tail_balance (+ one 64-byte read-back: host clock around the call)",
               "roofline": {"bound": "latency", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                            "whole_list_equivalent_GBps": round(alg / gpu_ms / 1e6, 1), "whole_list_bytes": alg,
                            "note": "round 4: the boundary search walks the list from its END and stops at the first hit, like the reference's "
                                    "(find_next_document_index.h:39-98): the bytes touched are O(last document), the call is four launches and a wait; rounds 2-3 "
                                    "mapped over the whole list (4 B of list + 1 B of document per structural: 0.075 of peak, 0.546 ms) -- whole_list_equivalent_GBps "
                                    "is that accounting, kept for comparison, and is not a bandwidth"}}
        L_ = capi.load_library()
        hidx = np.zeros(n + 16, dtype=np.uint32)
        t0 = time.perf_counter()
        hidx[: n + 3] = keep.cpu().numpy().view(np.uint32)  # the host walk needs the list at home first
        t_copy = time.perf_counter() - t0
        n_io, nx = ctypes.c_uint32(0), ctypes.c_uint32(0)
        t0 = time.perf_counter()
        herr = L_.sjgpu_stage1_finish_host(host.ctypes.data, L, mode, hidx.ctypes.data, n, flags, ctypes.byref(n_io), ctypes.byref(nx))
        t_walk = time.perf_counter() - t0
        assert (herr, n_io.value) == (err, n_kept), (herr, n_io.value, err, n_kept)
        leg["host_finish"] = {"download_ms": round(t_copy * 1e3, 2), "walk_ms": round(t_walk * 1e3, 4),
                              "note": "sjgpu_stage1_finish_host (the reference's backward walk) is O(last document) once the list is in host memory; bringing it there is the cost"}
    p.close()
    return leg


def leg_tape(cx):
    """SURVEY 8(f3), stage 2 on the device: the reference's DOM tape (sjgpu_stage2_device: strings + tape) for resident documents, next to the
    reference kernel's stage2() on one host core; tape and string buffer compared word for word with the reference's dom parse."""
    import ctypes
    torch, capi, corpus = cx.torch, cx.capi, cx.corpus
    cpu = cx.cpu()
    R = impl = None
    if cpu.LIB_REF and os.path.exists(cpu.LIB_REF):
        R = ctypes.CDLL(cpu.LIB_REF)
        R.sjref_available.argtypes = [ctypes.c_char_p]
        impl = next((i for i in (b"icelake", b"haswell", b"westmere") if R.sjref_available(i)), None)
    out = {}
    for kind in ("twitter_like", "large_random"):
        host, _ = make_workload(corpus, kind, 256 << 20, 3000)
        L = len(host)
        p = capi.DomParserImplementation(L, device=cx.local_rank)
        stream = torch.cuda.current_stream().cuda_stream
        buf = torch.from_numpy(host).cuda()
        idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
        assert p.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream) == 0
        n, flags, _ = p.result(stream)
        assert flags == 0
        tape = torch.empty(L + 8, dtype=torch.int64, device="cuda")
        scap = 5 * (L // 3) + 256
        sbuf = torch.empty(scap, dtype=torch.uint8, device="cuda")
        run = lambda: p.stage2_device(buf.data_ptr(), L, idx.data_ptr(), n, tape.data_ptr(), L + 8, sbuf.data_ptr(), scap, 1024, stream)
        err, tw, sb = run()
        if err != 0:
            raise SystemExit(f"tape/{kind}: error {err} on the synthetic document")
        reps = 8
        first_ms = event_ms_per_call(torch, run, reps)
        clock_warmup(torch, run)
        gpu_ms = event_ms_per_call(torch, run, reps)
        alg = L + 4 * (n + 1) + 8 * tw + sb
        leg = {"workload": f"{kind} {L} B, {n} structurals -> {tw} tape words + {sb} B of string records (dom::document of the reference)",
               "gpu_ms_per_call": round(gpu_ms, 3), "first_reps_ms_per_call": round(first_ms, 3), "timing": FIRST_REPS_NOTE, "value": round(L / gpu_ms / 1e6, 1),
               "unit": "GB/s of document", "kernel": "k_tok_stage (token bytes, atoms and numbers from one staged pass over the document) / scan_sums + string buffer (" + ("stream compaction: k_strs_count / resolve / write" if p.string_path() == 1 else "per-string walk")
                         + ") + k_tok_apply (tape positions, string, atom and number words) + k_radix_hist / scatter (container ordinals) + k_tape_match / rules + 1 scan (sjgpu_tape.hip, "
                           "sjgpu_string_stream.hip)",
               "roofline": {"bound": "hbm", "achieved": round(alg / gpu_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / gpu_ms / 1e6 / HBM_PEAK_GBS, 4),
                            "algorithmic_bytes_per_launch": alg,
                            "algorithmic_bytes": "document + 4 (n + 1) list in, 8 tape words + string records out; all kernels of one sjgpu_stage2_device call together",
                            "traffic": cx.traffic.get(f"tape:{kind}:{256 << 20}"),
                            "traffic_source": "profiles/traffic.json (profiles/r06_pmc_summary.txt): FETCH_SIZE x 2 + WRITE_SIZE summed over the kernels of one call"},
               "note": "includes the 48-byte result read-back of every call; document, list, tape and string buffer stay on the device"}
        # round 5: stage 1 + stage 2 with the token stream between them (sjgpu_stage1_tokens_device -> sjgpu_stage2_tokens_device) against the two plain calls
        tok = torch.empty(L // 2 + 16, dtype=torch.uint8, device="cuda")
        idx_t = torch.empty(L // 2 + 16, dtype=torch.int32, device="cuda")
        if n + 3 <= L // 2:
            p.set_pipeline("split")
            s1_plain = lambda: p.stage1_device(buf.data_ptr(), L, idx_t.data_ptr(), L // 2, stream)
            s1_tok = lambda: p.stage1_tokens_device(buf.data_ptr(), L, idx_t.data_ptr(), L // 2, tok.data_ptr(), L // 2 + 16, stream)
            s1_tok()
            assert p.result(stream)[0] == n
            tape_t = torch.empty(L + 8, dtype=torch.int64, device="cuda")
            run_t = lambda: p.stage2_device(buf.data_ptr(), L, idx_t.data_ptr(), n, tape_t.data_ptr(), L + 8, sbuf.data_ptr(), scap, 1024, stream, tok_ptr=tok.data_ptr())
            et, twt, sbt = run_t()
            if (et, twt, sbt) != (0, tw, sb) or not bool(torch.equal(tape_t[:tw], tape[:tw])):
                raise SystemExit(f"PARITY FAILURE tape/{kind}: stage 2 from the token stream differs from stage 2 from the document")
            clock_warmup(torch, s1_plain)
            ms_s1 = event_ms_per_call(torch, s1_plain, reps)
            clock_warmup(torch, s1_tok)
            ms_s1t = event_ms_per_call(torch, s1_tok, reps)
            clock_warmup(torch, run_t)
            ms_t = event_ms_per_call(torch, run_t, reps)
            leg["with_token_stream"] = {"stage2_ms_per_call": round(ms_t, 3), "stage1_split_ms_without_tokens": round(ms_s1, 4), "stage1_split_ms_with_tokens": round(ms_s1t, 4),
                                        "stage1_plus_stage2_ms": {"plain": round(ms_s1 + gpu_ms, 3), "token_stream": round(ms_s1t + ms_t, 3)},
                                        "parity": "the same tape, word for word, as the call that gathers its token bytes out of the document",
                                        "note": "since round 6 the tape's token front (k_tok_stage) takes the token bytes from the staged document: the stream is accepted and not read"}
            del tape_t
        del tok, idx_t
        if impl:
            R.sjref_dom_parse.restype = ctypes.c_int
            R.sjref_dom_parse.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64),
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
            R.sjref_bench_stage2.restype = ctypes.c_double
            R.sjref_bench_stage2.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
            padded = np.concatenate([host, np.zeros(128, np.uint8)])
            rt = np.zeros(L + 72, dtype=np.uint64)
            rs = np.zeros(scap + 64, dtype=np.uint8)
            rtw, rsb = ctypes.c_uint64(0), ctypes.c_uint64(0)
            rerr = R.sjref_dom_parse(impl, padded.ctypes.data, L, 1024, rt.ctypes.data, len(rt), ctypes.byref(rtw), rs.ctypes.data, len(rs), ctypes.byref(rsb))
            same = rerr == 0 and rtw.value == tw and rsb.value == sb and bool((tape[:tw].cpu().numpy().view(np.uint64) == rt[:tw]).all()) and \
                bool((sbuf[:sb].cpu().numpy() == rs[:sb]).all())
            if not same:
                raise SystemExit(f"PARITY FAILURE tape/{kind}: reference {rerr} / {rtw.value} words / {rsb.value} B vs {tw} / {sb}")
            leg["parity"] = "doc.tape and doc.string_buf of the reference's dom::parser::parse, word for word"
            cerr = ctypes.c_int(0)
            sec = R.sjref_bench_stage2(impl, padded.ctypes.data, L, 2, ctypes.byref(cerr))
            if sec > 0 and cerr.value == 0:
                leg["cpu_baseline"] = {"value": round(L / sec / 1e9, 3), "unit": "GB/s of document", "cores": 1, "kind": "reference",
                                       "sample": f"{impl.decode()} kernel, dom_parser_implementation::stage2 of the same document after one stage1, 1 thread, best of 2 ({sec * 1e3:.1f} ms)"}
        p.close()
        del buf, idx, tape, sbuf
        out[kind] = leg
    return out


def leg_dom_parse(cx, R, impl):
    """dom::parser::parse end to end, host buffers in, tape + string buffer out (VERDICT r03 missing #4): the reference on one core, the
    reference benchmarker's method (benchmark/benchmarker.h:315-346: parser and document allocated once, best of several parses of the same
    buffer), beside the two roads the plug-in's parse() has -- (A) stage 1 on the GPU + the reference's stage 2 on its list (what documents
    below SJGPU_STAGE2_FROM_KB take; timed as sjgpu_stage1 on the host buffer + the reference's stage2() alone: the shim lends the list, nothing
    is copied in between) and (B) sjgpu_parse, stage 2 on the device too (from SJGPU_STAGE2_FROM_KB = 1024 on).  The sweep is what the
    threshold rests on."""
    import ctypes
    capi = cx.capi
    R.sjref_bench_parse.restype = ctypes.c_double
    R.sjref_bench_parse.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    R.sjref_bench_stage2.restype = ctypes.c_double
    R.sjref_bench_stage2.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L = capi.load_library()
    rows = {}
    for label, size in (("0.6MiB", 630_000), ("1MiB", 1 << 20), ("2MiB", 2 << 20), ("4MiB", 4 << 20), ("64MiB", 64 << 20), ("256MiB", 256 << 20)):
        doc, _ = cx.corpus.twitter_like(size, 91)
        n = len(doc)
        padded = np.concatenate([doc, np.full(64, 0x20, np.uint8)])  # SIMDJSON_PADDING readable bytes behind the document
        iters = 3 if n > (32 << 20) else 10
        err = ctypes.c_int(0)
        t_ref = R.sjref_bench_parse(impl, padded.ctypes.data, n, iters, ctypes.byref(err))
        t_ref2 = R.sjref_bench_stage2(impl, padded.ctypes.data, n, iters, ctypes.byref(err))
        if t_ref <= 0 or t_ref2 <= 0 or err.value != 0:
            rows[label] = {"error": f"reference parse failed ({t_ref}, {t_ref2}, error {err.value})"}
            continue
        p = capi.DomParserImplementation(n, device=cx.local_rank)
        tape = np.zeros(n + 8, dtype=np.uint64)
        sbuf = np.zeros(5 * (n // 3) + 256, dtype=np.uint8)
        tw, sb = ctypes.c_uint64(0), ctypes.c_uint64(0)
        best_s1 = best_parse = 1e9
        for it in range(iters + 1):
            t0 = time.perf_counter()
            e1 = p.stage1(doc, capi.REGULAR)
            dt1 = time.perf_counter() - t0
            t0 = time.perf_counter()
            rc = L.sjgpu_parse(p.h, doc.ctypes.data, n, 1024, tape.ctypes.data, len(tape), sbuf.ctypes.data, len(sbuf), ctypes.byref(tw), ctypes.byref(sb))
            dt2 = time.perf_counter() - t0
            assert e1 == 0 and rc == 0, (label, e1, rc)
            if it:
                best_s1, best_parse = min(best_s1, dt1), min(best_parse, dt2)
        p.close()
        road_a = best_s1 + t_ref2
        rows[label] = {"bytes": n, "reference_parse_ms": round(t_ref * 1e3, 3), "reference_stage2_ms": round(t_ref2 * 1e3, 3),
                       "mi355x_stage1_host_ms": round(best_s1 * 1e3, 3), "road_a_gpu_stage1_plus_reference_stage2_ms": round(road_a * 1e3, 3),
                       "road_b_sjgpu_parse_ms": round(best_parse * 1e3, 3),
                       "reference_GBps": round(n / t_ref / 1e9, 3), "road_a_GBps": round(n / road_a / 1e9, 3), "road_b_GBps": round(n / best_parse / 1e9, 3),
                       "faster_road": "b" if best_parse < road_a else "a", "tape_words": int(tw.value)}
        del tape, sbuf
    return {"documents": "twitter-like, seed 91", "reference_kernel": impl.decode(), "threshold_SJGPU_STAGE2_FROM_KB": 1024, "sizes": rows,
            "note": "wall time per parse, host (pageable) buffers on both sides, best of the repetitions; road B includes the upload of the document and the download of "
                    "tape and string buffer (PCIe), road A the upload and the download of the structural list"}


def leg_plugin_host_path(cx, host_large):
    """the end-to-end plug-in number, PCIe included -- never `value`."""
    import ctypes
    capi = cx.capi
    cpu = cx.cpu()
    out = {}
    L = len(host_large)
    p = capi.DomParserImplementation(L, device=cx.local_rank)
    p.stage1(host_large, capi.REGULAR)  # first call: workspace + the runtime pins the pages it meets
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        err = p.stage1(host_large, capi.REGULAR)
        times.append(time.perf_counter() - t0)
    assert err == 0
    out["document_1GiB_pageable"] = {"ms_per_call": round(min(times) * 1e3, 2), "value": round(L / min(times) / 1e9, 2), "unit": "GB/s",
                                     "note": "large_random through sjgpu_stage1 with ordinary host memory: overlapped upload / scan / download in 16 MiB ranges"}
    p.close()
    # parse_many-sized batches: 1 MB windows of NDJSON, streaming_partial, as document_stream issues them
    nd, _ = cx.corpus.amazon_ndjson(64 << 20, 77)
    B = 1_000_000  # dom::DEFAULT_BATCH_SIZE
    windows = [nd[k: k + B] for k in range(0, len(nd) - B, B)][:48]
    q = capi.DomParserImplementation(B, device=cx.local_rank)
    for w in windows[:4]:
        q.stage1(w, capi.STREAMING_PARTIAL)
    t0 = time.perf_counter()
    for w in windows:
        q.stage1(w, capi.STREAMING_PARTIAL)
    gpu_us_unregistered = (time.perf_counter() - t0) / len(windows) * 1e6
    # the same calls with the stream registered, as document_stream::start() of the in-tree build (and mi355x::register_stream out of tree) does:
    # windows are cut out of spans scanned once
    capi.stream_register(nd)
    try:
        for w in windows[:2]:
            q.stage1(w, capi.STREAMING_PARTIAL)
        gpu_us = 1e9
        for _ in range(3):
            q.stage1(windows[0][:1000], capi.STREAMING_PARTIAL)  # leave the span: every repetition pays for its look-ahead scans again
            t0 = time.perf_counter()
            for w in windows:
                q.stage1(w, capi.STREAMING_PARTIAL)
            gpu_us = min(gpu_us, (time.perf_counter() - t0) / len(windows) * 1e6)
    finally:
        capi.stream_unregister(nd)
    q.close()
    R = ctypes.CDLL(cpu.LIB_REF)
    R.sjref_available.argtypes = [ctypes.c_char_p]
    impl = next((i for i in (b"icelake", b"haswell", b"westmere") if R.sjref_available(i)), None)
    R.sjref_parser_create.restype = ctypes.c_void_p
    R.sjref_parser_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    R.sjref_parser_stage1.restype = ctypes.c_int
    R.sjref_parser_stage1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    R.sjref_parser_destroy.argtypes = [ctypes.c_void_p]
    h = R.sjref_parser_create(impl, B)
    nn = ctypes.c_uint32(0)
    for w in windows[:4]:
        R.sjref_parser_stage1(h, w.ctypes.data, len(w), 1, None, ctypes.byref(nn), None)
    t0 = time.perf_counter()
    for w in windows:
        R.sjref_parser_stage1(h, w.ctypes.data, len(w), 1, None, ctypes.byref(nn), None)
    cpu_us = (time.perf_counter() - t0) / len(windows) * 1e6
    R.sjref_parser_destroy(h)
    out["parse_many_window_1MB"] = {"mi355x_us_per_window": round(gpu_us, 1), "mi355x_us_per_window_unregistered": round(gpu_us_unregistered, 1),
                                    "reference_us_per_window": round(cpu_us, 1), "reference_kernel": impl.decode(),
                                    "mi355x_GBps": round(B / gpu_us / 1e3, 2), "reference_GBps": round(B / cpu_us / 1e3, 2),
                                    "note": "sjgpu_stage1(streaming_partial) per 1 000 000-byte window of amazon NDJSON (dom::DEFAULT_BATCH_SIZE), host buffers, "
                                            "next to the reference's stage1 on one host thread.  Registered (sjgpu_stream_register, what document_stream::start() of the "
                                            "in-tree build does): the 48 windows are cut out of two 32 MiB spans scanned once each, look-ahead scans included in the time; "
                                            "unregistered: one upload, launch and download per window"}
    out["dom_parse"] = leg_dom_parse(cx, R, impl)
    # many small documents per launch
    lines = [bytes(l) for l in bytes(nd[: 4 << 20]).split(b"\n") if l][:8192]
    r = capi.DomParserImplementation(1 << 20, device=cx.local_rank)
    r.stage1_many(lines[:64])
    res = r.stage1_many(lines)
    prepared = r.prepare_many(lines)  # the sjgpu_doc array built once: what follows times libsjgpu, not ctypes marshalling
    r.stage1_many_prepared(prepared)
    dt_many = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        done = r.stage1_many_prepared(prepared)
        dt_many = min(dt_many, time.perf_counter() - t0)
    assert all(int(d.error) == 0 and int(d.n) == n0 for d, (_, n0, _) in zip(done, res))
    t0 = time.perf_counter()
    for l in lines[:512]:
        r.stage1(l, capi.REGULAR)
    dt_single = (time.perf_counter() - t0) / 512
    r.close()
    assert all(e == 0 for e, _, _ in res)
    nbytes = sum(len(l) for l in lines)
    out["small_documents"] = {"documents": len(lines), "mean_bytes": round(nbytes / len(lines), 1),
                              "one_launch_us_per_document": round(dt_many / len(lines) * 1e6, 3), "one_launch_GBps": round(nbytes / dt_many / 1e9, 3),
                              "one_call_each_us_per_document": round(dt_single * 1e6, 2),
                              "note": "sjgpu_stage1_many (one workgroup per document, ONE launch; the C call alone, its sjgpu_doc array marshalled beforehand: gather into the "
                                      "page-locked block, launch, wait, scatter of the lists) vs one sjgpu_stage1 call per document"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--op", default="stage1", choices=["stage1", "minify", "validate_utf8"])
    ap.add_argument("--workload", default=None, choices=["large_random", "amazon_ndjson", "twitter_like", "deep_nesting", "escape_heavy"])
    ap.add_argument("--size", type=int, default=1 << 30, help="bytes per GPU")
    ap.add_argument("--pipeline", default=os.environ.get("SJGPU_PIPELINE", "auto"), choices=["auto", "fused", "split"])
    ap.add_argument("--legs", default="all", help="N = 1: which of the other BASELINE configs to measure into the same line: all | none | comma list of "
                    + ",".join(ALL_LEGS))
    ap.add_argument("--ndjson-leg", type=int, default=-1, help="N > 1: 0 switches the NDJSON / one-document legs off")
    ap.add_argument("--cpu-iters", type=int, default=12)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--share-device", action="store_true", help="dry run: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from simdjson_amd import build, capi, corpus

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU of this node, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]])
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
    cx = Ctx(args, torch, capi, corpus, local_rank)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the headline: one synthetic buffer per rank, resident in HBM before anything is timed ----
    # EVERY N takes the same workload by default -- BASELINE.json configs[1], large_random, one independent 1 GiB document per rank (weak scaling, no
    # data-path collective) -- so that a sweep `bench.py --gpus 1,2,4,8` is ONE workload end to end and value(N) / (N * value(1)) means something.
    # (Rounds 3-4 switched the N > 1 headline to amazon NDJSON: a curve assembled from those lines jumped workloads between its first and second
    # point.)  configs[3] -- the NDJSON shards with the RCCL index concatenation -- travels in every line: `legs.config3_amazon_ndjson` at N = 1,
    # `config3_ndjson_sharded` at N > 1, each N > 1 figure with the same-workload single-rank figure measured inside the same run; and
    # `--workload amazon_ndjson` makes NDJSON the top-level workload at ANY N, N = 1 included (roofline, cpu_baseline, cpu_baseline_threads).
    if args.workload is None:
        args.workload = "large_random"
    host, units = make_workload(corpus, args.workload, args.size, 1000 + rank)
    L = len(host)
    # the reference beside it: rank 0 times it (one thread on its own buffer; NDJSON adds all hardware threads);
    # EVERY rank checks its own output against the reference, the verdicts are reduced into the line
    with_cpu = rank == 0 and not args.no_cpu_baseline
    with_parity = not args.no_cpu_baseline
    leg = device_leg(cx, args.op, args.workload, host, units, args.steps, args.warmup, args.pipeline, fence=fence, with_cpu=with_cpu, with_parity=with_parity,
                     parity_raises=world == 1, rank=rank, world=world)
    parity_failed_somewhere = False
    if world > 1:
        mine_ok = 1.0 if leg.get("parity", {}).get("ok", False) else 0.0
        pv = torch.tensor([mine_ok, 1.0 if "parity" in leg else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(pv)
        if "parity" in leg or rank == 0:
            leg.setdefault("parity", {"checked": False})
            leg["parity"].update({"ranks_checked": int(pv[1].item()), "ranks_ok": int(pv[0].item()), "all_ranks_ok": int(pv[0].item()) == world,
                                  "note": "every rank compares the output of its own timed buffer with the reference (count + order-sensitive digest); this entry is rank 0's, "
                                          "the counters are the all-reduced verdicts"})
        parity_failed_somewhere = with_parity and int(pv[0].item()) != world
    if with_cpu and args.op == "stage1" and args.workload == "amazon_ndjson":
        threads = os.cpu_count() or 1
        cb = cx.cpu().time_cpu_ndjson_threads(host, threads, 3)
        if cb is not None:
            leg["cpu_baseline_threads"] = {"value": round(cb["value"], 2), "unit": "GB/s", "cores": cb["cores"], "kind": cb["kind"],
                                           "sample": f"rank 0's NDJSON buffer cut at newlines into {cb['cores']} slices, {cb['impl']} kernel, one parser per thread, "
                                                     f"{threads} hardware threads on the box", "structurals": cb["n"]}
    dt = leg["ms_per_step"] * 1e-3 * args.steps
    dt_first = leg["first_reps_ms_per_step"] * 1e-3 * args.steps
    if world > 1:
        t = torch.tensor([dt, dt_first], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_first = float(t[0].item()), float(t[1].item())
        tot = torch.tensor([float(L)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bytes = float(tot.item())
    else:
        total_bytes = float(L)
    line = None
    if rank == 0:
        value = total_bytes * args.steps / dt / 1e9
        line = {
            "metric": "stage1 GB/s (structural indexing)" if args.op == "stage1" else f"{args.op} GB/s",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload} {L} B per GPU (seed 1000+rank), op={args.op}, regular mode, device-resident input and output, "
                                   f"{leg['pipeline']} pipeline", "bytes_per_gpu": L, "units_per_gpu": units, "structurals": leg["structurals"],
                       "out_bytes": leg["out_bytes"], "parallelism": f"{world} independent shard(s), one rank per GPU, no data-path collective",
                       "library": os.path.basename(capi._paths.LIB_SJGPU)},
            "roofline": leg["roofline"],
            "clock_warmup_calls": leg.get("clock_warmup_calls"),  # untimed calls in front of the timed region, on top of --warmup: see clock_warmup()
            # the same K steps timed straight behind the --warmup steps, before clock_warmup (MAX over ranks): what a cold `--warmup W --steps K` run gives
            "first_reps_ms_per_step": round(dt_first / args.steps * 1e3, 4), "value_first_reps": round(total_bytes * args.steps / dt_first / 1e9, 2),
            "timing": "value / ms_per_step: K steps behind W warm-up steps + clock_warmup (40 ms of the same calls, untimed: the first 30-40 calls after an idle "
                      "phase run 5-15 % slow, profiles/r04_drift_probe.txt); value_first_reps / first_reps_ms_per_step: the same K steps behind the W warm-up steps alone",
            "same_workload_at_every_n": f"{args.workload}: `bench.py --gpus N` takes this workload at N = 1 and at N > 1 unless --workload names another one, so "
                                        "value(N) / (N x value(1)) compares like with like; configs[3] (sharded NDJSON + RCCL index concatenation) is "
                                        + ("`legs.config3_amazon_ndjson`" if world == 1 else "`config3_ndjson_sharded`") + " of this line",
        }
        if world > 1 and leg.get("n1_same_workload"):
            n1 = leg["n1_same_workload"]
            line["n1_same_workload_GBps"] = n1["value"]
            line["n1_same_workload"] = n1
            line["scaling_efficiency"] = round(value / (world * n1["value"]), 4)
            line["scaling_efficiency_is"] = "value / (n_gpus x n1_same_workload_GBps), both measured in THIS run on THIS workload"
        for k in ("cpu_baseline", "cpu_baseline_threads", "parity"):
            if k in leg:
                line[k] = leg[k]

    # ---- N = 1: the other BASELINE configs, same method, into the same line ----
    if world == 1 and args.op == "stage1" and args.workload == "large_random" and args.legs != "none":
        wanted = ALL_LEGS if args.legs == "all" else [x for x in args.legs.split(",") if x]
        legs = {}
        sub_steps, sub_warm = max(5, args.steps // 2), 2

        def guarded(name, fn):
            if name not in wanted:
                return
            try:
                legs[name] = fn()
            except SystemExit:
                raise  # parity failures must fail the bench
            except Exception as e:  # an infrastructure hiccup in one leg must not cost the headline line
                legs[name] = {"error": repr(e)[:300]}

        guarded("config0_twitter_json", lambda: leg_twitter_json(cx))
        guarded("config2_minify", lambda: device_leg(cx, "minify", "large_random", host, units, sub_steps, sub_warm, args.pipeline, with_cpu=with_cpu))
        guarded("config2_validate_utf8", lambda: device_leg(cx, "validate_utf8", "large_random", host, units, sub_steps, sub_warm, args.pipeline, with_cpu=with_cpu))
        guarded("plugin_host_path", lambda: leg_plugin_host_path(cx, host))
        del host
        guarded("next_f2_finish_device", lambda: leg_list_passes(cx, "finish"))
        guarded("next_f3_depth_scan", lambda: leg_list_passes(cx, "depth"))
        guarded("next_f3_parse_strings", lambda: leg_parse_strings(cx))
        guarded("next_f3_tape", lambda: leg_tape(cx))

        def ndjson():
            h, u = make_workload(corpus, "amazon_ndjson", args.size, 2000)
            out = device_leg(cx, "stage1", "amazon_ndjson", h, u, sub_steps, sub_warm, args.pipeline, with_cpu=with_cpu)
            if with_cpu:  # the same NDJSON on T independent host threads, one reference parser each
                threads = os.cpu_count() or 1
                cb = cx.cpu().time_cpu_ndjson_threads(h, threads, 3)
                if cb is not None:
                    out["cpu_baseline_threads"] = {"value": round(cb["value"], 2), "unit": "GB/s", "cores": cb["cores"], "kind": cb["kind"],
                                                   "sample": f"the same NDJSON buffer cut at newlines into {cb['cores']} slices, {cb['impl']} kernel, one parser per thread, "
                                                             f"{threads} hardware threads on the box", "structurals": cb["n"]}
            return out

        guarded("config3_amazon_ndjson", ndjson)
        for wl in ("deep_nesting", "escape_heavy"):
            def adversarial(wl=wl):
                h, u = make_workload(corpus, wl, args.size, 1000)
                return device_leg(cx, "stage1", wl, h, u, sub_steps, sub_warm, args.pipeline, with_cpu=with_cpu, cpu_iters=4)
            guarded(f"config4_{wl}", adversarial)
        line["legs"] = legs
        line["legs_failed"] = sorted(k for k, v in legs.items() if isinstance(v, dict) and "error" in v)  # a swallowed exception must be visible

    # ---- N > 1: BASELINE.json configs[3] (parse_many-style NDJSON shards, RCCL concatenation) and the one-document path ----
    ndjson = docshards = None
    if world > 1 and args.op == "stage1" and args.ndjson_leg != 0:
        try:
            ndjson = ndjson_leg(args, torch, dist, corpus, capi, rank, world, local_rank, fence)
        except Exception as e:  # the primary line must survive a failure here
            ndjson = {"error": repr(e)[:300]}
        try:
            docshards = document_leg(args, torch, dist, capi, corpus, rank, world, local_rank, fence)
        except Exception as e:
            docshards = {"error": repr(e)[:300]}
    if rank == 0:
        if ndjson is not None:
            line["config3_ndjson_sharded"] = ndjson
            # what the first multi-GPU run has to show at a glance: did RCCL see all ranks, and which road did the index concatenation take
            line["n_ranks_seen_by_rccl"] = ndjson.get("n_ranks_seen_by_rccl")
            line["index_concat"] = ndjson.get("index_concat")
        if docshards is not None:
            line["one_document_shards"] = docshards
        write_detail(line)
        sys.stdout.flush()
        print(compact_line(line), flush=True)  # the ONE line, and the last one: <= 4 KB, numbers and identifiers (the full record: bench_detail.json)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed_somewhere:
        raise SystemExit("PARITY FAILURE on at least one rank (see the line's parity object)")


def document_leg(args, torch, dist, capi, corpus, rank, world, local_rank, fence):
    """Every rank's resident buffer taken as its shard of ONE document of world x --size bytes (the large_random
    buffers end in a newline, so the cuts between them are clean cuts): string-parity pre-pass, all_gather of one
    integer per rank, shard scan with the carried in-string bit (include/sjgpu.h, "one large document sharded
    across GPUs").  Timed like the primary leg: barrier + synchronize, MAX over ranks."""
    import time as _t
    host, _ = corpus.large_random(args.size, 1000 + rank)
    L = len(host)
    torch.cuda.empty_cache()  # (device_leg: a leg starts from an empty caching allocator)
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
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
    per-rank stage 1 with zero carry-in; then the same plus the gather that concatenates the index
    arrays into global 64-bit positions (simdjson_amd.sharded)."""
    import time as _t
    from simdjson_amd import sharded
    host, lines = corpus.amazon_ndjson(args.size, 2000 + rank)  # each rank's slice of the stream (ends in '\n')
    L = len(host)
    torch.cuda.empty_cache()  # (device_leg: a leg starts from an empty caching allocator)
    scanner = sharded.GpuShardScanner(L, local_rank)
    scanner.parser.set_pipeline(args.pipeline)  # AUTO learns the density from the warm-up scans
    buf = torch.from_numpy(host).cuda()
    idx = torch.empty(L + 16, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    step = lambda: scanner.parser.stage1_device(buf.data_ptr(), L, idx.data_ptr(), L + 3, stream)
    for _ in range(2):
        step()
    n, flags, _ = scanner.parser.result(stream)
    steps = max(4, args.steps // 2)
    # the same-workload single-rank figure, inside this run: rank 0 alone over its shard while the others wait in the second fence
    fence()
    n1 = None
    if rank == 0:
        clock_warmup(torch, step)
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        n1 = L * steps / (_t.perf_counter() - t0) / 1e9
    fence()
    clock_warmup(torch, step)
    fence()
    t0 = _t.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt_scan = _t.perf_counter() - t0
    out = {"workload": f"amazon_ndjson {L} B per GPU, newline-aligned shards, zero carry-in (BASELINE.json configs[3])", "structurals_rank0": n,
           "lines_rank0": lines, "flags_rank0": flags}
    base = torch.tensor([L], dtype=torch.int64, device="cuda")
    sizes = [torch.empty_like(base) for _ in range(world)]
    dist.all_gather(sizes, base)
    my_base = sum(int(x) for x in sizes[:rank])
    # the exchange below the C-ABI (sjgpu_comm_*: RCCL from C++); torch.distributed only carries the 128-byte id to the ranks
    comm = None
    exchange = "sjgpu_comm_gather_indices (libsjgpu: ncclAllGather of (n, base) + exact-count ncclSend / ncclRecv to rank 0, widened to 64-bit global positions there)"
    box = [None]
    if rank == 0:
        try:
            box[0] = capi.comm_unique_id()
        except Exception as e:  # the ranks still meet in the broadcast below and all take the twin
            box[0] = None
    dist.broadcast_object_list(box, src=0)
    try:
        if box[0] is None:
            raise RuntimeError("rank 0 could not create a communicator id")
        gathered = torch.empty(world * (L // 8 + 1024) if rank == 0 else 8, dtype=torch.int64, device="cuda")
        comm = capi.Comm(rank, world, box[0], local_rank)
        out["n_ranks_seen_by_rccl"] = comm.ranks()
    except Exception as e:  # never seen N > 1 hardware: keep the leg alive on the torch.distributed twin and say so
        comm = None
        exchange = f"sharded.gather_to_root over torch.distributed (sjgpu_comm unavailable: {repr(e)[:120]})"
    # every rank takes the same road: one rank without a communicator sends all of them to the twin (no rank may wait in a
    # collective the others never enter)
    have = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(have, op=dist.ReduceOp.MIN)
    if int(have) == 0 and comm is not None:
        comm.close()
        comm = None
        exchange = "sharded.gather_to_root over torch.distributed (sjgpu_comm unavailable on another rank)"
        out["n_ranks_seen_by_rccl"] = None

    def concat(n_now, f_now):
        if comm is not None:
            total, counts = comm.gather_indices(idx.data_ptr(), n_now, my_base, 0, gathered.data_ptr(), gathered.numel(), stream)
            return (gathered[:total] if rank == 0 else None), counts
        return sharded.gather_to_root(sharded.ShardScan(my_base, L, n_now, f_now, idx))

    concat(n, flags)  # warm the communicator
    fence()
    t0 = _t.perf_counter()
    for _ in range(steps):
        step()
        n2, f2, _ = scanner.parser.result(stream)
        pos, counts = concat(n2, f2)
    fence()
    dt_cat = _t.perf_counter() - t0
    t = torch.tensor([dt_scan, dt_cat], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_scan, dt_cat = float(t[0]), float(t[1])
    tot = torch.tensor([float(L)], dtype=torch.float64, device="cuda")
    dist.all_reduce(tot)
    total = float(tot)
    out["with_index_concat_GBps"] = round(total * steps / dt_cat / 1e9, 2)
    out["index_concat"] = exchange
    if rank == 0:
        out["total_structurals"] = int(sum(counts))
        out["sorted_global_positions"] = bool((pos[1:] > pos[:-1]).all()) if len(pos) > 1 else True
    out["value_GBps"] = round(total * steps / dt_scan / 1e9, 2)
    out["steps"] = steps
    if rank == 0 and not args.no_cpu_baseline:  # the reference over rank 0's shard, one thread and all hardware threads of the box
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cpu_baseline as cpu  # test infrastructure: the CPU side of the comparison only
        cb1 = cpu.time_cpu(host, "stage1", 3)
        out["cpu_baseline"] = {"value": round(cb1["value"], 3), "unit": "GB/s", "cores": cb1["cores"], "kind": cb1["kind"],
                               "sample": f"rank 0's {L}-byte shard, {cb1['impl']} kernel, 1 thread, best of 3"}
        threads = os.cpu_count() or 1
        cbt = cpu.time_cpu_ndjson_threads(host, threads, 3)
        if cbt is not None:
            out["cpu_baseline_threads"] = {"value": round(cbt["value"], 2), "unit": "GB/s", "cores": cbt["cores"], "kind": cbt["kind"],
                                           "sample": f"rank 0's shard cut at newlines into {cbt['cores']} slices, {cbt['impl']} kernel, one parser per thread, "
                                                     f"{threads} hardware threads on the box", "structurals": cbt["n"]}
    if n1 is not None:
        out["n1_same_workload_GBps"] = round(n1, 2)
        out["scaling_efficiency"] = round(total * steps / dt_scan / 1e9 / (world * n1), 4)
        out["scaling_efficiency_is"] = ("value_GBps / (n_gpus x n1_same_workload_GBps): rank 0 alone over its own shard (the others waiting in a barrier) against all ranks "
                                        "together, same run, same shards, same kernels; with_index_concat_GBps adds the gather to rank 0 and is link-bound")
    if comm is not None:
        comm.close()
    scanner.parser.close()
    return out


if __name__ == "__main__":
    main()
