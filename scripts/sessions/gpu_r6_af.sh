#!/bin/bash
# round 6, session AF: the sort's tile (elements per wave) 2048 / 1024 / 512
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py v16=build/ab/libsjgpu_v16.so t1024=build/ab/libsjgpu_t1024.so t512=build/ab/libsjgpu_t512.so > $O/r6af_tape_ab.txt 2> $O/r6af_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6af_tape_ab.txt; tail -3 $O/r6af_tape_ab.err
