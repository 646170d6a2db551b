#!/bin/bash
# round 6, session S: k_tok_stage's row loop two and four rows at a time (v6b, v6) against one (v5)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py v5=build/ab/libsjgpu_v5.so v6=build/ab/libsjgpu_v6.so v6b=build/ab/libsjgpu_v6b.so v5x=build/ab/libsjgpu_v5.so > $O/r6s_tape_ab.txt 2> $O/r6s_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6s_tape_ab.txt; tail -3 $O/r6s_tape_ab.err
