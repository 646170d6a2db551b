#!/bin/bash
# round-3 counter passes: the headline kernel, sparse NDJSON (split), minify, and the tape -- each counter set its own rocprofv3 run
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
LAB=$GRAFT_REPO_ROOT/build/lab/perf_lab
G=1073741824
timeout 900 bash scripts/gpu_pmc_cmd.sh r3_headline "fetch write sq1 sq2 tcp ta" -- $LAB --size $G --kinds large_random --ops stage1 --fused-only --reps 5 --no-check > gpurun_out/pmc_r3_headline.txt 2>&1
timeout 600 bash scripts/gpu_pmc_cmd.sh r3_ndjson "fetch write sq1 sq2" -- $LAB --size $G --kinds amazon_ndjson --ops stage1 --reps 5 --no-check > gpurun_out/pmc_r3_ndjson.txt 2>&1
timeout 600 bash scripts/gpu_pmc_cmd.sh r3_minify "fetch write sq1 sq2" -- $LAB --size $G --kinds large_random --ops minify --fused-only --reps 5 --no-check > gpurun_out/pmc_r3_minify.txt 2>&1
timeout 600 bash scripts/gpu_pmc_cmd.sh r3_escape "fetch write" -- $LAB --size $G --kinds escape_heavy --ops stage1 --reps 5 --no-check > gpurun_out/pmc_r3_escape.txt 2>&1
timeout 600 bash scripts/gpu_pmc_cmd.sh r3_tape_tw "fetch write sq1" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like > gpurun_out/pmc_r3_tape_tw.txt 2>&1
timeout 600 bash scripts/gpu_pmc_cmd.sh r3_tape_lr "fetch write" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py large_random > gpurun_out/pmc_r3_tape_lr.txt 2>&1
tail -n 40 gpurun_out/pmc_r3_headline.txt
