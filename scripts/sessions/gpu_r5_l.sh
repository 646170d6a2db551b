#!/bin/bash
# round 5, session L: + the masks as [segment][chunk pair][lane] (whole lines per instruction) and streamed by emit (stream2) against session K (stream1) and plain loads
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py plain=build/ab/libsjgpu_M1.so stream1=build/ab/libsjgpu_stream1.so stream2=simdjson_amd/lib/libsjgpu.so --rounds 12 --reps 10 > $O/r5l_lib_ab.txt 2> $O/r5l_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5l_lib_ab.txt; tail -5 $O/r5l_lib_ab.err
