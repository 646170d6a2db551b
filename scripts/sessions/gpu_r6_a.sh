#!/bin/bash
# round 6, session A: the tree as round 5 left it with the compact bench line -- the driver's own command, the line's size, smoke, the GPU tier
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python __graft_entry__.py --smoke > $O/r6a_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r6a_smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6a_bench.out 2> $O/r6a_bench.err; echo "bench rc=$?"
tail -n 1 $O/r6a_bench.out | wc -c; tail -n 1 $O/r6a_bench.out
cp bench_detail.json $O/r6a_bench_detail.json 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1400 -p no:cacheprovider > $O/r6a_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6a_pytest_gpu.log
