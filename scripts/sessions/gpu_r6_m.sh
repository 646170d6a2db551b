#!/bin/bash
# round 6, session M: k_stage1_summarize with its four chunks through ONE copy of the scan (15 KB of code) against the unrolled kernel (43 KB), three copies of each library in one process
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="escape_heavy:split:stage1,amazon_ndjson:split:stage1,twitter_like:split:stage1,large_random:split:stage1"
timeout 1500 python scripts/lib_ab.py unr=build/ab/libsjgpu_unr.so roll=build/ab/libsjgpu_roll.so unr2=build/ab/libsjgpu_unr2.so roll2=build/ab/libsjgpu_roll2.so unr3=build/ab/libsjgpu_unr3.so roll3=build/ab/libsjgpu_roll3.so --rounds 8 --reps 10 > $O/r6m_lib_ab.txt 2> $O/r6m_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6m_lib_ab.txt; tail -3 $O/r6m_lib_ab.err
