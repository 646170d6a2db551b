#!/bin/bash
# round 6, session AC: kernel times and instruction counts of the tape after v15
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for kind in large_random twitter_like; do
  bash scripts/gpu_pmc_cmd.sh r6ac_$kind "sq1" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $kind > $O/r6ac_pmc_$kind.log 2>&1
  python - <<PY
import csv, glob, collections, json
d = collections.defaultdict(list)
for f in glob.glob("$O/pmc_r6ac_$kind/sq1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
s = json.load(open("$O/pmc_r6ac_$kind/summary.json"))
print("== $kind")
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "sjgpu" in k and "stage1" not in k and "resolve_" not in k:
        c = next((x for n, x in s.items() if k[:60] in n or n[:60] in k), {})
        us = sum(v) / len(v); tot += us
        print("%8.1f us  VALU %7.1f M  SALU %7.1f M  %s" % (us, c.get("SQ_INSTS_VALU", 0) / 1e6, c.get("SQ_INSTS_SALU", 0) / 1e6, k.replace("sjgpu::(anonymous namespace)::", "")[:40]))
print("%8.1f us  sum" % tot)
PY
done
