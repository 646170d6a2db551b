#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --share-device --size 67108864 > gpurun_out/r03_bench_n2_dry.json 2> gpurun_out/r03_bench_n2_dry.err; echo "n2 dry rc=$?"; tail -c 1500 gpurun_out/r03_bench_n2_dry.json; tail -5 gpurun_out/r03_bench_n2_dry.err
