#!/bin/bash
# round 6, session X: v12 = k_strs_resolve requests the next tile ahead, k_tok_scan_sums eight entries at a time, the token's own rule without a branch; v13 = v12 + stage 2's results read back into page-locked memory
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python scripts/tape_ab.py v8=build/ab/libsjgpu_v8.so v12=build/ab/libsjgpu_v12.so v13=build/ab/libsjgpu_v13.so > $O/r6x_tape_ab.txt 2> $O/r6x_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6x_tape_ab.txt; tail -3 $O/r6x_tape_ab.err
