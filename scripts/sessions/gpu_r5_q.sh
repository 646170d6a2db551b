#!/bin/bash
# round 5, session Q: validate_utf8 requests its next chunk ahead (tree) and k_stage1_summarize the same (S8: 16 more VGPRs) against the tree before
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py before=build/ab/libsjgpu_S5tree.so tree=simdjson_amd/lib/libsjgpu.so S8_summ_ahead=build/ab/libsjgpu_S8.so --quick --rounds 12 --reps 10 > $O/r5q_lib_ab.txt 2> $O/r5q_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5q_lib_ab.txt; tail -3 $O/r5q_lib_ab.err
