#!/bin/bash
# round 5, session O: k_stage1_summarize with 32 KiB of LDS (five workgroups per CU: S5) and with 97 VGPRs (S6) against the tree (stream3)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py plain=build/ab/libsjgpu_M1.so stream3=simdjson_amd/lib/libsjgpu.so S5_5wg=build/ab/libsjgpu_S5.so S6_97vgpr=build/ab/libsjgpu_S6.so --quick --rounds 12 --reps 10 > $O/r5o_lib_ab.txt 2> $O/r5o_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5o_lib_ab.txt; tail -5 $O/r5o_lib_ab.err
