#!/bin/bash
# counter passes over the restructured tape (stage 2 incl. the string stream), twitter-like and large_random, 256 MiB
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 bash scripts/gpu_pmc_cmd.sh r3b_tape_tw "fetch write sq1" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like > gpurun_out/pmc_r3b_tape_tw.txt 2>&1
timeout 600 bash scripts/gpu_pmc_cmd.sh r3b_tape_lr "fetch write" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py large_random > gpurun_out/pmc_r3b_tape_lr.txt 2>&1
tail -5 gpurun_out/pmc_r3b_tape_tw.txt
