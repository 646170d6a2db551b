#!/usr/bin/env python3
"""Turns the artefacts of scripts/gpu_r6_final.sh <SRC> (gpurun_out/SRC_*) into the files committed under profiles/ (named r06_*):
  r06_bench_final.json        the line bench.py PRINTED for the driver's command (compact: <= 4 KB)        r06_bench_detail.json   the full record of the same run
  r06_bench_n2_dry.json       the N = 2 dry run's printed line (gloo, both ranks on one device)             r06_bench_n2_dry_detail.json
  r06_bench_ndjson_n1.json    --workload amazon_ndjson at N = 1, printed line                               r06_bench_ndjson_n1_detail.json
  r06_final_kernel_stats.txt  rocprofv3 --kernel-trace of the same bench command: per-kernel table + the headline leg's dispatches
Usage: python scripts/profiles_from_run.py [gpurun_out] [SRC tag, default r06]"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r06"
SRC = sys.argv[2] if len(sys.argv) > 2 else TAG
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")

def last_text(path):  # the bench line is the LAST line of stdout
    return [l for l in open(path).read().splitlines() if l.strip()][-1]


def last_line(path):
    return json.loads(last_text(path))


def printed_and_detail(src_printed, src_detail, name):
    text = last_text(os.path.join(out, src_printed))
    assert len(text) <= 4096 and text.startswith("{"), (src_printed, len(text))
    open(os.path.join(prof, name + ".json"), "w").write(text + "\n")
    json.dump(json.load(open(os.path.join(out, src_detail))), open(os.path.join(prof, name + "_detail.json"), "w"), indent=1)


printed_and_detail(SRC + "_bench_default.json", SRC + "_bench_detail.json", TAG + "_bench_final")
os.replace(os.path.join(prof, TAG + "_bench_final_detail.json"), os.path.join(prof, TAG + "_bench_detail.json"))
if os.path.exists(os.path.join(out, SRC + "_bench_n2_dry.json")):
    printed_and_detail(SRC + "_bench_n2_dry.json", SRC + "_bench_n2_dry_detail.json", TAG + "_bench_n2_dry")

prof_line = json.load(open(os.path.join(out, SRC + "_bench_profiled_detail.json")))
db = os.path.join(out, "prof_" + SRC + "_bench_final", "b_results.db")
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
per = {}
for name, s, e in rows:
    if "sjgpu" not in name:
        continue
    short = name.replace("sjgpu::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    per.setdefault(short, []).append((e - s) / 1000.0)
head = per.get("k_fused_pipelined<0, false, 4u, 8u, false>", []) or per.get("k_fused_pipelined<0, false, 4u, 8u>", [])  # (eight waves per workgroup since round 4)
# the headline leg's dispatches: --warmup calls, one more (AUTO settles), the clock warm-up's (bench.py: clock_warmup), then the timed steps
# (round 5: the `steps` first repetitions -- value_first_reps -- lie between the warm-up calls and the clock warm-up)
skip = max(prof_line["warmup"], 1) + 1 + prof_line["steps"] + int(prof_line.get("clock_warmup_calls") or 0)
first = head[skip: skip + prof_line["steps"]]
r = prof_line["roofline"]
with open(os.path.join(prof, TAG + "_final_kernel_stats.txt"), "w") as f:
    f.write("# " + TAG + " (final): rocprofv3 --kernel-trace -- python bench.py (the driver's command: --gpus 1 --steps 20 --warmup 5, all legs; scripts/gpu_r6_final.sh); sjgpu kernels only, from the\n"
            "# rocpd database rocprofv3 writes (view `kernels`; scripts/profiles_from_run.py).  The headline kernel is k_fused_pipelined<0, false, 4u, 8u> (eight waves, 128 KiB tiles): the TIMED dispatches of the\n"
            "# headline leg (large_random, 1 GiB: behind 5 + 1 warm-up calls, the 20 first repetitions and the clock warm-up's) are listed first; the others belong to the deep_nesting leg, the 256 MiB documents of the stage-2 legs and\n"
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
if os.path.exists(os.path.join(out, SRC + "_bench_ndjson_n1.json")):
    printed_and_detail(SRC + "_bench_ndjson_n1.json", SRC + "_bench_ndjson_n1_detail.json", TAG + "_bench_ndjson_n1")
print(f"profiles/{TAG}_bench_final.json, {TAG}_bench_n2_dry.json, {TAG}_bench_ndjson_n1.json, {TAG}_final_kernel_stats.txt written")
