#!/bin/bash
# what do the length words cost k_tape_write?  the same call with the per-string kernels forced (they write the length words themselves)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in twitter_like; do
  (cd /tmp && SJGPU_STRING_STREAM=0 timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_old_$k -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $k > $GRAFT_REPO_ROOT/gpurun_out/prof_old_$k.log 2>&1); echo "prof $k rc=$?"; tail -1 gpurun_out/prof_old_$k.log
done
