#!/bin/bash
# round 6, session C: sparse segments as lists + both resolve levels in one launch (the tree) against the library before, one process; then the GPU tier
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="escape_heavy:split:stage1,amazon_ndjson:split:stage1,twitter_like:split:stage1,large_random:split:stage1"
timeout 1500 python scripts/lib_ab.py before=build/ab/libsjgpu_before.so tree=simdjson_amd/lib/libsjgpu.so before2=build/ab/libsjgpu_before2.so tree2=build/ab/libsjgpu_tree2.so --rounds 10 --reps 10 > $O/r6c_lib_ab.txt 2> $O/r6c_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6c_lib_ab.txt; tail -3 $O/r6c_lib_ab.err
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1400 -p no:cacheprovider > $O/r6c_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/r6c_pytest_gpu.log | tail -3
