#!/bin/bash
# round 6, session AI: lab -- k_tok_stage without its second sweep (the parse from memory compiled out: 84 VGPRs instead of 128; same results on these documents)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py v18=build/ab/libsjgpu_v18.so noredo=build/ab/libsjgpu_noredo.so > $O/r6ai_tape_ab.txt 2> $O/r6ai_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ai_tape_ab.txt; tail -3 $O/r6ai_tape_ab.err
