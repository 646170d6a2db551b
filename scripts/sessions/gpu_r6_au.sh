#!/bin/bash
# round 6, session AU: v27 = v26 with 16 patch entries per lane and round (4 KiB of list per wave: four workgroups per CU instead of five): k_strs_write in the trace
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in v26 v27; do
  for kind in twitter_like large_random; do
    (cd /tmp && SJGPU_LIB=$R/build/ab/libsjgpu_$v.so timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r6au_${v}_$kind -o t -- python $R/scripts/tape_once.py $kind 268435456 > $R/gpurun_out/r6au_${v}_$kind.log 2>&1); echo "$v $kind rc=$?"
    python3 scripts/rocpd_summary.py gpurun_out/prof_r6au_${v}_$kind/t_results.db 2>/dev/null | grep "k_strs_write" | cut -c1-100
  done
done
