#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tape or stage2_device" --timeout 600 -p no:cacheprovider > gpurun_out/r03_g_parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r03_g_parity.log
timeout 600 python bench.py --legs next_f3_tape,config2_minify --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03_g_bench.json 2> gpurun_out/r03_g_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03_g_bench.json"))
    for k, v in d["legs"]["next_f3_tape"].items():
        print(k, v["gpu_ms_per_call"], v["value"], v["roofline"]["frac"])
    m = d["legs"]["config2_minify"]; print("minify", m["roofline"]["gpu_ms_per_step"], m["roofline"]["frac"])
    print("headline", d["roofline"]["gpu_ms_per_step"], d["roofline"]["frac"], "failed:", d.get("legs_failed"))
except Exception as e:
    print("no bench line:", e)
PY
tail -2 gpurun_out/r03_g_bench.err
