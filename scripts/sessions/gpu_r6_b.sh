#!/bin/bash
# round 6, session B: k_stage1_direct (SJGPU_DIRECT=1 blockIdx order, =2 tickets) against the tree's pipelined kernel and the split pipeline, one process
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="large_random:fused:stage1,large_random:split:stage1,amazon_ndjson:split:stage1,amazon_ndjson:fused:stage1,twitter_like:split:stage1,twitter_like:fused:stage1,escape_heavy:split:stage1,escape_heavy:fused:stage1,deep_nesting_doc:fused:stage1"
timeout 1500 python scripts/lib_ab.py tree=simdjson_amd/lib/libsjgpu.so d1=build/ab/libsjgpu_d1.so,SJGPU_DIRECT=1 d2=build/ab/libsjgpu_d2.so,SJGPU_DIRECT=2 --rounds 8 --reps 10 > $O/r6b_lib_ab.txt 2> $O/r6b_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6b_lib_ab.txt; tail -3 $O/r6b_lib_ab.err
