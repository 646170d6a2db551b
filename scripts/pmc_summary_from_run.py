#!/usr/bin/env python3
"""profiles/r06_pmc_summary.txt and the tape keys of profiles/traffic.json from the counter tables of scripts/gpu_r6_final.sh <SRC> (gpurun_out/SRC_pmc_tables.txt):
the header lines of the summary are kept, the tables replaced, the tape's traffic = the rows from k_tape_init on (one sjgpu_stage2_device call).
Usage: python scripts/pmc_summary_from_run.py SRC"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
tables = open(os.path.join(ROOT, "gpurun_out", src + "_pmc_tables.txt")).read().splitlines()
vals = {}
for tag in ("tape_tw", "tape_lr"):
    on = sect = False
    f = w = 0.0
    for l in tables:
        if l.startswith("== "):
            sect, on = tag in l, False
            continue
        if not sect:
            continue
        if l.startswith("k_tape_init"):
            on = True
        if on and l.startswith("k_"):
            m = re.match(r"(\S.*?)\s+([\d.]+)\s+([\d.]+)\s", l)
            f += float(m.group(2)); w += float(m.group(3))
    vals[tag] = (round(f, 3), round(w, 3), round(f + w, 3))
p = os.path.join(ROOT, "profiles", "r06_pmc_summary.txt")
L = open(p).read().splitlines()
head = [l for l in L[:40] if l.startswith("#")]
note = "# LAST tables: session %s -- the tape per call: twitter-like %.3f GB (%.3f fetched + %.3f written), large_random %.3f GB (%.3f + %.3f)" % (
    src, vals["tape_tw"][2], vals["tape_tw"][0], vals["tape_tw"][1], vals["tape_lr"][2], vals["tape_lr"][0], vals["tape_lr"][1])
head = [l for l in head if not l.startswith("# LAST tables:")] + [note, ""]
open(p, "w").write("\n".join(head + tables) + "\n")
tp = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tp))
t["tape:twitter_like:268435456"] = int(round(vals["tape_tw"][2] * 1e9))
t["tape:large_random:268435456"] = int(round(vals["tape_lr"][2] * 1e9))
if ("session " + src) not in t["_comment"]:
    t["_comment"] += "; session %s: %.3f / %.3f" % (src, vals["tape_tw"][2], vals["tape_lr"][2])
json.dump(t, open(tp, "w"), indent=1)
print(note)
