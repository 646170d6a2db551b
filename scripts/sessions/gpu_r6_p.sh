#!/bin/bash
# round 6, session P: k_tok_stage v2 (bytes of a number read from LDS one by one, the counters' contributions from an LDS table, 16-bit offsets: four workgroups per CU)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6p_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6p_pytest.log
timeout 900 python scripts/tape_ab.py base=build/ab/libsjgpu_base.so v1=build/ab/libsjgpu_w4k.so v2=build/ab/libsjgpu_v2.so v2o3=build/ab/libsjgpu_v2o3.so > $O/r6p_tape_ab.txt 2> $O/r6p_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6p_tape_ab.txt; tail -3 $O/r6p_tape_ab.err
for kind in large_random twitter_like; do
  bash scripts/gpu_pmc_cmd.sh r6p_$kind "sq1 sq2" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $kind > $O/r6p_pmc_$kind.log 2>&1
  python scripts/pmc_table.py $O/pmc_r6p_$kind | grep "kernel\|k_tok"
  python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$O/pmc_r6p_$kind/sq1/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "sjgpu" in k: print("%8.1f us x %d  %s" % (sum(v) / len(v), len(v), k))
PY
done
