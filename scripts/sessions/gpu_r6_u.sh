#!/bin/bash
# round 6, session U: v8 = the token's own rule inside k_tok_apply (k_tape_rules gone, its last thread's work in k_tape_match), 32-bit indexes in k_tok_apply
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6u_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6u_pytest.log
timeout 900 python scripts/tape_ab.py base=build/ab/libsjgpu_base.so v7=build/ab/libsjgpu_v7.so v8=build/ab/libsjgpu_v8.so > $O/r6u_tape_ab.txt 2> $O/r6u_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6u_tape_ab.txt; tail -3 $O/r6u_tape_ab.err
for kind in large_random twitter_like; do
  bash scripts/gpu_pmc_cmd.sh r6u_$kind "sq1" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $kind > $O/r6u_pmc_$kind.log 2>&1
  python scripts/pmc_table.py $O/pmc_r6u_$kind | grep "kernel\|k_tok\|k_tape\|radix\|strs"
  python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$O/pmc_r6u_$kind/sq1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "sjgpu" in k and sum(v)/len(v) > 20: print("%8.1f us x %d  %s" % (sum(v) / len(v), len(v), k))
PY
done
