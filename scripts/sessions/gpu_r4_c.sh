#!/bin/bash
# round 4, third GPU session: the GPU tier, the default bench line (new: plugin_host_path.dom_parse), counter passes for the sparse road
# (parked UTF-8 blocks: FETCH_SIZE must drop by ~0.23 GB), escape_heavy (no table: traffic ~1.3x) and the headline
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r4c_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4c_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4c_bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_slots"])
for k, v in d.get("legs", {}).items():
    if isinstance(v, dict) and "roofline" in v:
        print(k, v.get("value"), v.get("ms_per_step"), v["roofline"].get("frac"), v["roofline"].get("kernel_ms_slots"))
    elif isinstance(v, dict):
        print(k, json.dumps(v)[:700])
print("failed", d.get("legs_failed"))
PY
bash scripts/gpu_pmc.sh "--workload amazon_ndjson" r4_ndjson "fetch write sq1 sq2" 2>&1 | tail -30
bash scripts/gpu_pmc.sh "--workload escape_heavy" r4_escape "fetch write" 2>&1 | tail -12
bash scripts/gpu_pmc.sh "--op stage1" r4_headline "fetch write sq1" 2>&1 | tail -12
