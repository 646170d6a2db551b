#!/bin/bash
# round 4, second GPU session: the whole GPU tier again (new: comm failure path, in-tree On-Demand streams, sanitizers), escape_heavy on
# both pipelines after the parallel look-back fold, and the N = 2 dry runs of bench.py (self-launch; gloo on one shared device)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r4b_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r4b_pytest_gpu.log
for pl in split fused; do
  timeout 600 python bench.py --legs none --workload escape_heavy --pipeline $pl --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4b_escape_$pl.json 2> gpurun_out/r4b_escape_$pl.err; cut -c1-120 gpurun_out/r4b_escape_$pl.json; grep -o '"frac": [0-9.]*' gpurun_out/r4b_escape_$pl.json | head -1
done
# python bench.py --gpus 2 without a launcher: must become the launcher (one GPU here, so both ranks share it over gloo)
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > gpurun_out/r4b_bench_n2_selflaunch.json 2> gpurun_out/r4b_bench_n2_selflaunch.err; echo "n2 self-launch rc=$?"; cut -c1-2500 gpurun_out/r4b_bench_n2_selflaunch.json; tail -3 gpurun_out/r4b_bench_n2_selflaunch.err
