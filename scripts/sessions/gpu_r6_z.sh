#!/bin/bash
# round 6, session Z: where a wave of k_tok_stage spends its time (lab build with phase clocks)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/lab/stage_phases.py build/ab/libsjgpu_phases.so > gpurun_out/r6z_stage_phases.txt 2> gpurun_out/r6z_stage_phases.err; echo "rc=$?"; cat gpurun_out/r6z_stage_phases.txt; tail -3 gpurun_out/r6z_stage_phases.err
