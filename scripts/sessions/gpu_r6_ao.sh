#!/bin/bash
# round 6, session AO: kernel trace of the tape (v22) on both documents: what k_strs_resolve / k_strs_write take now
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in twitter_like large_random; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r6ao_$kind -o t -- python $R/scripts/tape_once.py $kind 268435456 > $R/gpurun_out/r6ao_$kind.log 2>&1); echo "$kind rc=$?"
  python3 scripts/rocpd_summary.py gpurun_out/prof_r6ao_$kind/*/t_results.db gpurun_out/prof_r6ao_$kind/t_results.db 2>/dev/null | head -30 | cut -c1-120
done
