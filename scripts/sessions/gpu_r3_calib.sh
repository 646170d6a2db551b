#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 bash scripts/gpu_pmc_cmd.sh r3_calib "fetch tcc tcc2" -- $GRAFT_REPO_ROOT/scripts/micro/fetch_calib.bin > gpurun_out/pmc_r3_calib.txt 2>&1
cat gpurun_out/pmc_r3_calib.txt | tail -30
