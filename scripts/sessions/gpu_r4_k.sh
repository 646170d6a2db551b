#!/bin/bash
# round 4, session k: a kernel trace of the standalone string pass (bench.py --legs next_f3_parse_strings)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r4k}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_strs -o t -- python $GRAFT_REPO_ROOT/bench.py --legs next_f3_parse_strings > $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_strs.log 2>&1); echo "trace rc=$?"
python3 scripts/rocpd_summary.py gpurun_out/prof_${T}_strs/t_results.db | head -24 | cut -c1-150
