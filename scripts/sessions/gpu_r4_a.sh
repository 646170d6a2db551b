#!/bin/bash
# round 4, first GPU session: smoke, the whole GPU tier, the default bench line, escape_heavy on both pipelines
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py --smoke > gpurun_out/r4a_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r4a_smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r4a_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r4a_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r4a_bench.json
for pl in split fused; do
  timeout 600 python bench.py --legs none --workload escape_heavy --pipeline $pl --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4a_escape_$pl.json 2> gpurun_out/r4a_escape_$pl.err; cut -c1-700 gpurun_out/r4a_escape_$pl.json
done
