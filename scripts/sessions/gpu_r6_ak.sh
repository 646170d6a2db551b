#!/bin/bash
# round 6, session AK: k_tok_apply asked for five waves per SIMD once more (a5: 96 VGPRs, 28 B of scratch after the round's diet; 52 B in session V)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py v19=build/ab/libsjgpu_v19.so a5=build/ab/libsjgpu_a5.so > $O/r6ak_tape_ab.txt 2> $O/r6ak_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ak_tape_ab.txt; tail -3 $O/r6ak_tape_ab.err
