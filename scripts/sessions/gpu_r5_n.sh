#!/bin/bash
# round 5, session N: + the pipelined single-pass kernel fetches its chunks streamed (stream4)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py plain=build/ab/libsjgpu_M1.so stream3=build/ab/libsjgpu_stream3.so stream4=simdjson_amd/lib/libsjgpu.so --rounds 12 --reps 10 > $O/r5n_lib_ab.txt 2> $O/r5n_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5n_lib_ab.txt; tail -5 $O/r5n_lib_ab.err
