#!/bin/bash
# round 5, session J: the input's loads with the non-temporal hint (nt) against the same tree with plain loads (M1), and the trees before
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py base=build/ab/libsjgpu_base.so prev=build/ab/libsjgpu_prev.so plain=build/ab/libsjgpu_M1.so nt=simdjson_amd/lib/libsjgpu.so --rounds 12 --reps 10 > $O/r5j_lib_ab.txt 2> $O/r5j_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5j_lib_ab.txt; tail -5 $O/r5j_lib_ab.err
