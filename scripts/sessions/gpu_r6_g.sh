#!/bin/bash
# round 6, session G: k_stage1_direct (persistent, additive prefixes, emission deferred by one tile) against both pipelines; then its phases (a lab build with stamps)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="amazon_ndjson:split:stage1,amazon_ndjson:fused:stage1,large_random:fused:stage1,twitter_like:split:stage1,twitter_like:fused:stage1"
timeout 1500 python scripts/lib_ab.py tree=build/ab/libsjgpu_tree.so d=build/ab/libsjgpu_d.so,SJGPU_DIRECT=1 --rounds 8 --reps 10 > $O/r6g_lib_ab.txt 2> $O/r6g_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6g_lib_ab.txt | head -8; tail -3 $O/r6g_lib_ab.err
SJGPU_LIB=$GRAFT_REPO_ROOT/build/ab/libsjgpu_stamps.so timeout 600 python scripts/direct_phases.py amazon_ndjson 16 2>&1 | tail -8
SJGPU_LIB=$GRAFT_REPO_ROOT/build/ab/libsjgpu_stamps.so timeout 600 python scripts/direct_phases.py large_random 16 2>&1 | tail -8
