#!/bin/bash
# round 6, session AY: k_tape_match's two scattered bracket words stored with the non-temporal hint (lab build, the tree untouched): the kernel in the trace, the tape A/B
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 900 python scripts/tape_ab.py v32=build/ab/libsjgpu_v32.so nt=build/ab/libsjgpu_lab_nt.so > $O/r6ay_tape_ab.txt 2> $O/r6ay_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ay_tape_ab.txt; grep digests_equal $O/r6ay_tape_ab.txt | cut -c1-140
for v in v32 lab_nt; do
  for kind in twitter_like large_random; do
    (cd /tmp && SJGPU_LIB=$R/build/ab/libsjgpu_$v.so timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r6ay_${v}_$kind -o t -- python $R/scripts/tape_once.py $kind 268435456 > $O/r6ay_${v}_$kind.log 2>&1); echo "$v $kind rc=$?"
    python3 scripts/rocpd_summary.py gpurun_out/prof_r6ay_${v}_$kind/t_results.db 2>/dev/null | grep "k_tape_match\|k_tok_apply" | cut -c1-100
  done
done
