#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in large_random twitter_like; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tape_$k -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $k > $GRAFT_REPO_ROOT/gpurun_out/prof_tape_$k.log 2>&1); echo "prof $k rc=$?"; tail -1 gpurun_out/prof_tape_$k.log
done
ls gpurun_out/prof_tape_large_random
