#!/bin/bash
# round 6, session V: v9 = k_tok_apply asked for five waves per SIMD (96 VGPRs, 52 B of scratch), v10 = the token's own rule without a branch, v11 = v10 with an 8 KiB window (three workgroups per CU)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py v8=build/ab/libsjgpu_v8.so v9=build/ab/libsjgpu_v9.so v10=build/ab/libsjgpu_v10.so v11=build/ab/libsjgpu_v11.so > $O/r6v_tape_ab.txt 2> $O/r6v_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6v_tape_ab.txt; tail -3 $O/r6v_tape_ab.err
