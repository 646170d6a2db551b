#!/bin/bash
# round 6, session Y: timing experiment -- the string pass on a second stream beside the token front (lab build `two`: its decide kernel reads the previous call's count)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py v13=build/ab/libsjgpu_v13.so two=build/ab/libsjgpu_two.so > $O/r6y_tape_ab.txt 2> $O/r6y_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6y_tape_ab.txt; tail -3 $O/r6y_tape_ab.err
