#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_noesc -o t -- python $GRAFT_REPO_ROOT/scripts/strings_noescape.py > $GRAFT_REPO_ROOT/gpurun_out/strings_noescape.log 2>&1); echo "rc=$?"; grep "ms per call" gpurun_out/strings_noescape.log
