#!/bin/bash
# round 6, session H: where k_stage1_direct's time goes -- lab switches (wrong results, timings only)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
export LIB_AB_JOBS="amazon_ndjson:fused:stage1"
timeout 1500 python scripts/lib_ab.py d=build/ab/libsjgpu_d.so,SJGPU_DIRECT=1 static=build/ab/libsjgpu_static.so,SJGPU_DIRECT=1,SJGPU_DIRECT_LAB=8 all=build/ab/libsjgpu_all.so,SJGPU_DIRECT=1,SJGPU_DIRECT_LAB=7 allstatic=build/ab/libsjgpu_allstatic.so,SJGPU_DIRECT=1,SJGPU_DIRECT_LAB=15 plain=build/ab/libsjgpu_plain.so,SJGPU_DIRECT=1 --rounds 4 --reps 5 > $O/r6h_lib_ab.txt 2> $O/r6h_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6h_lib_ab.txt | head -8; tail -3 $O/r6h_lib_ab.err
