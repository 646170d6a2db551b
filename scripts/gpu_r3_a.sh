#!/bin/bash
# round 3, GPU session A: the overlapped split pipeline against both existing pipelines (same box, same buffers), the
# byte-mover ceiling for every workload's read/write mix, and the prefetch-less single-pass kernel (no scratch).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
timeout 300 scripts/micro/mix_copy.bin > gpurun_out/r03_mix_copy.txt 2>&1; echo "mix_copy rc=$?"; cat gpurun_out/r03_mix_copy.txt
timeout 900 python scripts/overlap_sweep.py large_random,amazon_ndjson,twitter_like > gpurun_out/r03_overlap_sweep.jsonl 2> gpurun_out/r03_overlap_sweep.err; echo "sweep rc=$?"; cat gpurun_out/r03_overlap_sweep.jsonl; tail -5 gpurun_out/r03_overlap_sweep.err
SJGPU_PREFETCH=0 SWEEP_OVERLAPS= timeout 600 python scripts/overlap_sweep.py large_random,amazon_ndjson > gpurun_out/r03_noprefetch.jsonl 2> gpurun_out/r03_noprefetch.err; echo "nopf rc=$?"; cat gpurun_out/r03_noprefetch.jsonl
SWEEP_OVERLAPS=64 timeout 600 python scripts/overlap_sweep.py deep_nesting,escape_heavy > gpurun_out/r03_overlap_adversarial.jsonl 2> gpurun_out/r03_overlap_adversarial.err; echo "adv rc=$?"; cat gpurun_out/r03_overlap_adversarial.jsonl
