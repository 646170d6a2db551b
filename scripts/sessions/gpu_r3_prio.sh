#!/bin/bash
# A/B: wave priority by phase in k_fused_pipelined (SJGPU_PRIO = policy, sjgpu_fused.hip: phase_prio); same box, same command
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/r03_prio.jsonl
for wl in large_random amazon_ndjson; do
  for pol in 0 1 2 3 4 5 0; do
    SJGPU_PRIO=$pol timeout 300 python bench.py --legs none --pipeline fused --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'$wl','policy':$pol,'gpu_ms':r['gpu_ms_per_step'],'frac':r['frac'],'value':d['value']}))" >> gpurun_out/r03_prio.jsonl
  done
done
cat gpurun_out/r03_prio.jsonl
