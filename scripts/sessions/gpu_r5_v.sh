#!/bin/bash
# round 5, session V: k_fused_pipelined with the wave's next chunk on its way into LDS while it scans this one (SJGPU_PIPE_AHEAD=1) against the same library without
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python scripts/lib_ab.py tree=simdjson_amd/lib/libsjgpu.so ahead=build/ab/libsjgpu_ahead.so,SJGPU_PIPE_AHEAD=1 before=build/ab/libsjgpu_S5tree.so --rounds 12 --reps 10 > $O/r5v_lib_ab.txt 2> $O/r5v_lib_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r5v_lib_ab.txt | head -24; tail -3 $O/r5v_lib_ab.err
