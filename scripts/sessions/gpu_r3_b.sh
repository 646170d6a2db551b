#!/bin/bash
# round 3, GPU session B: everything new since session A -- parity of the tape / key matching / RCCL smoke / multi-GPU host path,
# the plug-in (device stage 2 through dom::parser), and the new bench legs.  Every command has its own timeout and no stdin.
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tape or stage2_device or raw_key or comm or mgpu" --timeout 600 -p no:cacheprovider > gpurun_out/r03_b_parity.log 2>&1; echo "parity rc=$?"; tail -5 gpurun_out/r03_b_parity.log
timeout 900 python -m pytest tests/test_plugin.py -q -x --timeout 600 -p no:cacheprovider > gpurun_out/r03_b_plugin.log 2>&1; echo "plugin rc=$?"; tail -5 gpurun_out/r03_b_plugin.log
timeout 600 python bench.py --legs next_f2_finish_device,next_f3_depth_scan,plugin_host_path --steps 5 --warmup 2 > gpurun_out/r03_b_bench.json 2> gpurun_out/r03_b_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03_b_bench.json"))
    print(json.dumps(d.get("legs"), indent=1)[:5000]); print("failed:", d.get("legs_failed"))
except Exception as e:
    print("no bench line:", e)
PY
tail -3 gpurun_out/r03_b_bench.err
