#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 bash scripts/gpu_pmc_cmd.sh r3c_tape_tw "sq1 sq2 tcp" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like > gpurun_out/pmc_r3c_tape_tw.txt 2>&1
tail -3 gpurun_out/pmc_r3c_tape_tw.txt | cut -c1-300
