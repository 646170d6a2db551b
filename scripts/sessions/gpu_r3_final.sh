#!/bin/bash
# round-end sequence (smoke, GPU tests, default bench) + a rocprofv3 kernel trace of the same bench command + the N = 2 dry run of bench.py
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py --smoke > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r03_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench_final -o b -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_bench_profiled.err); echo "profiled bench rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --share-device --size 67108864 > gpurun_out/r03_bench_n2_dry.json 2> gpurun_out/r03_bench_n2_dry.err; echo "n2 dry rc=$?"; tail -c 600 gpurun_out/r03_bench_n2_dry.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03_bench_default.json"))
    print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel"], "failed:", d.get("legs_failed"))
    for k, v in d.get("legs", {}).items():
        r = v.get("roofline") if isinstance(v, dict) else None
        print(k, (r or {}).get("frac"), v.get("value") if isinstance(v, dict) else None)
except Exception as e:
    print("no bench line:", e)
PY
