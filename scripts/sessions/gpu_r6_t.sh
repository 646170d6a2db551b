#!/bin/bash
# round 6, session T: v7 = k_radix_scatter's tiles behind the m elements return at once
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py base=build/ab/libsjgpu_base.so v5=build/ab/libsjgpu_v5.so v7=build/ab/libsjgpu_v7.so > $O/r6t_tape_ab.txt 2> $O/r6t_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6t_tape_ab.txt; tail -3 $O/r6t_tape_ab.err
