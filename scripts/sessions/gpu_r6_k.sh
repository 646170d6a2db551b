#!/bin/bash
# round 6, session K: tickets from two counters (SJGPU_TWO_TICKETS=1) against one, the pipelined single-pass kernels, one process
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="large_random:auto:minify,large_random:fused:stage1,amazon_ndjson:fused:stage1,deep_nesting_doc:fused:stage1"
timeout 1500 python scripts/lib_ab.py one=build/ab/libsjgpu_one.so two=build/ab/libsjgpu_two.so,SJGPU_TWO_TICKETS=1 one2=build/ab/libsjgpu_one2.so two2=build/ab/libsjgpu_two2.so,SJGPU_TWO_TICKETS=1 --rounds 10 --reps 10 > $O/r6k_lib_ab.txt 2> $O/r6k_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6k_lib_ab.txt | head -12; tail -3 $O/r6k_lib_ab.err
