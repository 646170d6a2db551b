#!/bin/bash
# round 3, GPU session C: stream windows, plug-in (device stage 2, in-tree document_stream with registration), bench legs
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "windows_of_a_registered or device_finish or raw_key or comm" --timeout 600 -p no:cacheprovider > gpurun_out/r03_c_parity.log 2>&1; echo "parity rc=$?"; tail -6 gpurun_out/r03_c_parity.log
timeout 1200 python -m pytest tests/test_plugin.py -q --timeout 900 -p no:cacheprovider > gpurun_out/r03_c_plugin.log 2>&1; echo "plugin rc=$?"; tail -12 gpurun_out/r03_c_plugin.log
timeout 600 python bench.py --legs next_f2_finish_device,plugin_host_path --steps 5 --warmup 2 > gpurun_out/r03_c_bench.json 2> gpurun_out/r03_c_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03_c_bench.json"))
    print(json.dumps(d.get("legs"), indent=1)[:4000]); print("failed:", d.get("legs_failed"))
except Exception as e:
    print("no bench line:", e)
PY
tail -3 gpurun_out/r03_c_bench.err
