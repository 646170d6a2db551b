#!/bin/bash
# round 6, session E: the pipelined kernels with TWO barriers per iteration (the tree) against three (SJGPU_TOP_BARRIER=1) and against round 5's shape
# (three barriers, the look-back behind the first: + SJGPU_LATE_LOOKBACK=1), one process; then the GPU tier (full-size digests: the race check)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="large_random:fused:stage1,large_random:auto:minify,amazon_ndjson:fused:stage1,twitter_like:fused:stage1,deep_nesting_doc:fused:stage1"
timeout 1500 python scripts/lib_ab.py tree=build/ab/libsjgpu_tree.so tb=build/ab/libsjgpu_tb.so,SJGPU_TOP_BARRIER=1 r5=build/ab/libsjgpu_r5.so,SJGPU_TOP_BARRIER=1,SJGPU_LATE_LOOKBACK=1 tree2=build/ab/libsjgpu_tree2.so --rounds 12 --reps 10 > $O/r6e_lib_ab.txt 2> $O/r6e_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6e_lib_ab.txt; tail -3 $O/r6e_lib_ab.err
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1400 -p no:cacheprovider > $O/r6e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/r6e_pytest_gpu.log | tail -3
