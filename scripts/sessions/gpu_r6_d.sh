#!/bin/bash
# round 6, session D: the pending tile's look-back in front of the scan's barrier (the tree) against behind it (SJGPU_LATE_LOOKBACK=1: rounds 1-5), one process
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="large_random:fused:stage1,large_random:auto:minify,amazon_ndjson:fused:stage1,twitter_like:fused:stage1,deep_nesting_doc:fused:stage1"
timeout 1500 python scripts/lib_ab.py early=build/ab/libsjgpu_early.so late=build/ab/libsjgpu_late.so,SJGPU_LATE_LOOKBACK=1 early2=build/ab/libsjgpu_early2.so late2=build/ab/libsjgpu_late2.so,SJGPU_LATE_LOOKBACK=1 --rounds 12 --reps 10 > $O/r6d_lib_ab.txt 2> $O/r6d_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6d_lib_ab.txt; tail -3 $O/r6d_lib_ab.err
