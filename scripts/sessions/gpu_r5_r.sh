#!/bin/bash
# round 5, session R: FOUR COPIES of one library as the variants of the A/B: what the harness itself contributes (workspace placement, position in the round)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py one=build/ab/libsjgpu_S5tree.so copyA=build/ab/libsjgpu_copyA.so copyB=build/ab/libsjgpu_copyB.so copyC=build/ab/libsjgpu_copyC.so --quick --rounds 12 --reps 10 > $O/r5r_lib_ab.txt 2> $O/r5r_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5r_lib_ab.txt; tail -3 $O/r5r_lib_ab.err
