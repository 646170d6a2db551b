#!/bin/bash
# round 6, session AS: where k_strs_write's time goes -- the kernel traced with one part compiled out at a time (-DLAB_NO_OUTQ: the strings' begin offsets without map.at,
# -DLAB_NO_SWEEP: no byte sweep, -DLAB_NO_PATCH: no patch rounds; outputs are wrong, times are what is read)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in v25 lab_NO_OUTQ lab_NO_SWEEP lab_NO_PATCH; do
  for kind in twitter_like large_random; do
    (cd /tmp && SJGPU_LIB=$R/build/ab/libsjgpu_$v.so timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r6as_${v}_$kind -o t -- python $R/scripts/tape_once.py $kind 268435456 > $R/gpurun_out/r6as_${v}_$kind.log 2>&1); echo "$v $kind rc=$?"
    python3 scripts/rocpd_summary.py gpurun_out/prof_r6as_${v}_$kind/t_results.db 2>/dev/null | grep "k_strs_write\|k_strs_count" | cut -c1-100
  done
done
