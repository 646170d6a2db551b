#!/bin/bash
# round 6, session O: counters of the staged token front (k_tok_stage) on both documents
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
for kind in large_random twitter_like; do
  bash scripts/gpu_pmc_cmd.sh r6o_$kind "sq1 sq2 fetch write" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $kind > gpurun_out/r6o_pmc_$kind.log 2>&1
  python scripts/pmc_table.py gpurun_out/pmc_r6o_$kind | grep -v "^$" | head -40
done
