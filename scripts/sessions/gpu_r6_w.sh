#!/bin/bash
# round 6, session W: k_tape_match with 2 / 8 elements per thread (m2, m8) and with the elements of a thread a workgroup apart (ms4, ms8: every load instruction covers consecutive elements)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py v10=build/ab/libsjgpu_v10.so m2=build/ab/libsjgpu_m2.so m8=build/ab/libsjgpu_m8.so ms4=build/ab/libsjgpu_ms4.so ms8=build/ab/libsjgpu_ms8.so > $O/r6w_tape_ab.txt 2> $O/r6w_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6w_tape_ab.txt; tail -3 $O/r6w_tape_ab.err
