#!/bin/bash
# round 5, session Z: k_stage1_emit's segment_prefix with the fast road for groups without x words (the tree) against the library before
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py before=build/ab/libsjgpu_S5tree.so tree=simdjson_amd/lib/libsjgpu.so before2=build/ab/libsjgpu_prevcopy.so tree2=build/ab/libsjgpu_treecopy.so --rounds 12 --reps 10 > $O/r5z_lib_ab.txt 2> $O/r5z_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5z_lib_ab.txt; tail -3 $O/r5z_lib_ab.err
