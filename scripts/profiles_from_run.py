#!/usr/bin/env python3
"""Turns the artefacts of scripts/gpu_r5_final.sh (gpurun_out/) into the files committed under profiles/ (TAG = r05; round 4's: git history):
  TAG_bench_final.json        the default bench line of the final tree
  TAG_bench_n2_dry.json       the N = 2 dry run (bench.py launching itself; gloo, both ranks on one device)
  TAG_bench_ndjson_n1.json    the same command with --workload amazon_ndjson at N = 1 (the single-rank point of configs[3]'s curve)
  TAG_final_kernel_stats.txt  rocprofv3 --kernel-trace of the same bench command: per-kernel table + the headline leg's dispatches
Usage: python scripts/profiles_from_run.py [gpurun_out]"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r05"
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")

def last_line(path):  # the bench line is the last line that starts with a brace
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


line = last_line(os.path.join(out, TAG + "_bench_default.json"))
json.dump(line, open(os.path.join(prof, TAG + "_bench_final.json"), "w"))
n2_text = [l for l in open(os.path.join(out, TAG + "_bench_n2_dry.json")).read().splitlines() if l.startswith("{")]
if n2_text:
    json.dump(json.loads(n2_text[-1]), open(os.path.join(prof, TAG + "_bench_n2_dry.json"), "w"))

prof_line = last_line(os.path.join(out, TAG + "_bench_profiled.json"))
db = os.path.join(out, "prof_" + TAG + "_bench_final", "b_results.db")
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
per = {}
for name, s, e in rows:
    if "sjgpu" not in name:
        continue
    short = name.replace("sjgpu::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    per.setdefault(short, []).append((e - s) / 1000.0)
head = per.get("k_fused_pipelined<0, false, 4u, 8u>", []) or per.get("k_fused_pipelined<0, false>", [])  # (eight waves per workgroup since round 4)
# the headline leg's dispatches: --warmup calls, one more (AUTO settles), the clock warm-up's (bench.py: clock_warmup), then the timed steps
# (round 5: the `steps` first repetitions -- value_first_reps -- lie between the warm-up calls and the clock warm-up)
skip = max(prof_line["warmup"], 1) + 1 + prof_line["steps"] + int(prof_line.get("clock_warmup_calls") or 0)
first = head[skip: skip + prof_line["steps"]]
r = prof_line["roofline"]
with open(os.path.join(prof, TAG + "_final_kernel_stats.txt"), "w") as f:
    f.write("# " + TAG + " (final): rocprofv3 --kernel-trace -- python bench.py (the default command: N = 1, 20 steps + 3 warm-up, all legs; scripts/gpu_r5_final.sh); sjgpu kernels only, from the\n"
            "# rocpd database rocprofv3 writes (view `kernels`; scripts/profiles_from_run.py).  The headline kernel is k_fused_pipelined<0, false, 4u, 8u> (eight waves, 128 KiB tiles): the TIMED dispatches of the\n"
            "# headline leg (large_random, 1 GiB: behind 3 + 1 warm-up calls, the 20 first repetitions and the clock warm-up's) are listed first; the others belong to the deep_nesting leg, the 256 MiB documents of the stage-2 legs and\n"
            "# the parity calls, so the mean over all calls mixes workloads.  The comparable figures:\n")
    if first:
        f.write(f"#   headline leg, the {len(first)} timed dispatches: mean {sum(first) / len(first):.1f} us, min {min(first):.1f}, max {max(first):.1f}  (kernel alone)\n")
    f.write(f"#   bench.py line of the SAME run: value {prof_line['value']} GB/s, ms_per_step {prof_line['ms_per_step']}, roofline gpu_ms_per_step {r['gpu_ms_per_step']} (HIP events around\n"
            f"#   what one call enqueues: the scan kernel alone since round 5), achieved {r['achieved']} GB/s, frac {r['frac']}, kernel {r['kernel']}\n")
    if first:
        alg = r["algorithmic_bytes_per_launch"]
        k = sum(first) / len(first)
        f.write(f"#   -> algorithmic {alg} B / kernel time = {alg / k / 1e3:.0f} GB/s = {alg / k / 1e3 / 8000:.3f} of 8 TB/s for the kernel alone\n")
    f.write(f"{'kernel':44s}{'calls':>7s}{'avg us':>11s}{'min us':>11s}{'max us':>11s}{'total ms':>11s}\n")
    for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"{name[:43]:44s}{len(d):7d}{sum(d) / len(d):11.1f}{min(d):11.1f}{max(d):11.1f}{sum(d) / 1000:11.2f}\n")
nd = [l for l in open(os.path.join(out, TAG + "_bench_ndjson_n1.json")).read().splitlines() if l.startswith("{")] if os.path.exists(os.path.join(out, TAG + "_bench_ndjson_n1.json")) else []
if nd:
    json.dump(json.loads(nd[-1]), open(os.path.join(prof, TAG + "_bench_ndjson_n1.json"), "w"))
print(f"profiles/{TAG}_bench_final.json, {TAG}_bench_n2_dry.json, {TAG}_bench_ndjson_n1.json, {TAG}_final_kernel_stats.txt written")
