#!/bin/bash
# scripts/gpu_check.sh -- one GPU-box session: smoke, GPU parity tests, bench lines, rocprofv3 stats.
# Usage (from the repo root, via gpurun): bash scripts/gpu_check.sh [full_size_bytes] [stages]
# Everything is bounded by `timeout`; logs land in gpurun_out/.
set -u
FULL=${1:-1073741824}
STAGES=${2:-"smoke tests bench prof"}
mkdir -p gpurun_out
export TMPDIR=/tmp SJGPU_FULL_SIZE=$FULL
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
lscpu | grep -E "Model name|^CPU\(s\)|Flags" | cut -c1-400 >> gpurun_out/gpu_info.txt
for s in $STAGES; do
  case $s in
    smoke) timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" ;;
    tests) timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log ;;
    bench)
      for op in stage1 minify validate_utf8; do
        timeout 600 python bench.py --legs none --op $op --steps 20 --warmup 3 > gpurun_out/bench_$op.json 2> gpurun_out/bench_$op.err; echo "bench $op rc=$?"; cat gpurun_out/bench_$op.json
      done
      timeout 600 python bench.py --legs none --pipeline fused --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_stage1_fused.json 2> gpurun_out/bench_stage1_fused.err; cat gpurun_out/bench_stage1_fused.json
      timeout 600 python bench.py --legs none --op minify --pipeline fused --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_minify_fused.json 2> gpurun_out/bench_minify_fused.err; cat gpurun_out/bench_minify_fused.json
      timeout 600 python bench.py --ndjson-leg 1 --steps 10 --warmup 2 > gpurun_out/bench_ndjson_leg.json 2> gpurun_out/bench_ndjson_leg.err; cat gpurun_out/bench_ndjson_leg.json
      timeout 600 python bench.py --legs none --workload amazon_ndjson --steps 20 --warmup 3 > gpurun_out/bench_stage1_ndjson.json 2> gpurun_out/bench_stage1_ndjson.err; cat gpurun_out/bench_stage1_ndjson.json
      timeout 600 python bench.py --legs none --workload twitter_like --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_stage1_twitter.json 2> gpurun_out/bench_stage1_twitter.err; cat gpurun_out/bench_stage1_twitter.json
      for wl in deep_nesting escape_heavy; do
        for pl in fused split; do
          timeout 600 python bench.py --legs none --workload $wl --pipeline $pl --steps 10 --warmup 2 > gpurun_out/bench_stage1_${wl}_${pl}.json 2> gpurun_out/bench_stage1_${wl}_${pl}.err; cat gpurun_out/bench_stage1_${wl}_${pl}.json
        done
      done
      ;;
    multi) # the N > 1 code path on a 1-GPU box: two ranks over gloo sharing the device (the driver's real runs use RCCL, one GPU per rank)
      timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > gpurun_out/bench_2ranks_shared.json 2> gpurun_out/bench_2ranks_shared.err; echo "multi rc=$?"; cat gpurun_out/bench_2ranks_shared.json ;;
    sweep)
      timeout 900 python scripts/size_sweep.py twitter_like > gpurun_out/size_sweep_twitter.jsonl 2> gpurun_out/size_sweep.err; cat gpurun_out/size_sweep_twitter.jsonl
      timeout 900 python scripts/size_sweep.py large_random > gpurun_out/size_sweep_large_random.jsonl 2>> gpurun_out/size_sweep.err; cat gpurun_out/size_sweep_large_random.jsonl
      ;;
    prof)
      for op in stage1 minify validate_utf8; do
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$op -o $op -- python $GRAFT_REPO_ROOT/bench.py --legs none --op $op --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$op.log 2>&1); echo "prof $op rc=$?"
      done
      find gpurun_out -name "*kernel_stats*" | head; for f in $(find gpurun_out -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
      ;;
  esac
done
