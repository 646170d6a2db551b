#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/r03_prio2.jsonl
for rep in 1 2 3; do
  for pol in 0 4 1; do
    SJGPU_PRIO=$pol timeout 300 python bench.py --legs none --pipeline fused --workload large_random --steps 30 --warmup 3 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'large_random','policy':$pol,'rep':$rep,'gpu_ms':r['gpu_ms_per_step'],'frac':r['frac'],'value':d['value']}))" >> gpurun_out/r03_prio2.jsonl
  done
done
for pol in 0 4; do
  SJGPU_PRIO=$pol timeout 300 python bench.py --legs none --pipeline fused --workload deep_nesting --steps 10 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'deep_nesting','policy':$pol,'gpu_ms':r['gpu_ms_per_step'],'frac':r['frac'],'value':d['value']}))" >> gpurun_out/r03_prio2.jsonl
  SJGPU_PRIO=$pol timeout 300 python bench.py --legs none --op minify --steps 20 --warmup 3 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'minify large_random','policy':$pol,'gpu_ms':r['gpu_ms_per_step'],'frac':r['frac'],'value':d['value']}))" >> gpurun_out/r03_prio2.jsonl
done
cat gpurun_out/r03_prio2.jsonl
