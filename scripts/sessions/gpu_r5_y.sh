#!/bin/bash
# round 5, session Y: the whole GPU tier and smoke on the last tree (what the driver runs at round end)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python __graft_entry__.py --smoke > $O/r5y_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r5y_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1400 -p no:cacheprovider > $O/r5y_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r5y_pytest_gpu.log
