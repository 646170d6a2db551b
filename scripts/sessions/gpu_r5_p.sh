#!/bin/bash
# round 5, session P: 26 KiB of LDS and 80 VGPRs (six workgroups per CU, 36 B of scratch: S7) against five (S5)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py plain=build/ab/libsjgpu_M1.so stream3=simdjson_amd/lib/libsjgpu.so S5_5wg=build/ab/libsjgpu_S5.so S7_6wg=build/ab/libsjgpu_S7.so --quick --rounds 12 --reps 10 > $O/r5p_lib_ab.txt 2> $O/r5p_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5p_lib_ab.txt; tail -5 $O/r5p_lib_ab.err
