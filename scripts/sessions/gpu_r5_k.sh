#!/bin/bash
# round 5, session K: chunks requested coalesced with the non-temporal hint, block order through LDS (split kernels, validate_utf8) against the same tree with the strided plain loads
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py base=build/ab/libsjgpu_base.so prev=build/ab/libsjgpu_prev.so plain=build/ab/libsjgpu_M1.so stream=simdjson_amd/lib/libsjgpu.so --rounds 12 --reps 10 > $O/r5k_lib_ab.txt 2> $O/r5k_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5k_lib_ab.txt; tail -5 $O/r5k_lib_ab.err
