#!/bin/bash
# round 5, session I2: in-process interleaved A/B (scripts/lib_ab.py): base = commit 4c7df53, prev = 3fc67b5 (the cheaper scan), M1 = the masks of
# k_stage1_summarize leave behind the rendezvous, M2 = M1 with 82 VGPRs (five waves per SIMD)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py base=build/ab/libsjgpu_base.so prev=build/ab/libsjgpu_prev.so M1=simdjson_amd/lib/libsjgpu.so M2_5waves=build/ab/libsjgpu_M2.so --rounds 12 --reps 10 > $O/r5i_lib_ab.txt 2> $O/r5i_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5i_lib_ab.txt; tail -5 $O/r5i_lib_ab.err
