#!/bin/bash
# round 6, session R: v5 = k_radix_hist with bins in registers; nonum = lab build whose k_tok_stage skips the numbers (what the classification alone costs)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py base=build/ab/libsjgpu_base.so v4=build/ab/libsjgpu_v4.so v5=build/ab/libsjgpu_v5.so nonum=build/ab/libsjgpu_nonum.so > $O/r6r_tape_ab.txt 2> $O/r6r_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6r_tape_ab.txt; tail -3 $O/r6r_tape_ab.err
cp build/ab/libsjgpu_nonum.so /tmp/nonum.so
for lib in v5 nonum; do
for kind in large_random twitter_like; do
  cp build/ab/libsjgpu_$lib.so simdjson_amd/lib/libsjgpu.so
  bash scripts/gpu_pmc_cmd.sh r6r_${lib}_$kind "sq1" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $kind > $O/r6r_pmc_${lib}_$kind.log 2>&1
  echo "== $lib $kind"; python scripts/pmc_table.py $O/pmc_r6r_${lib}_$kind | grep "k_tok_st\|radix_h"
  python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$O/pmc_r6r_${lib}_$kind/sq1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "k_tok_st" in k or "radix_h" in k: print("%8.1f us x %d  %s" % (sum(v) / len(v), len(v), k))
PY
done
done
