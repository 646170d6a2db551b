#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "windows_of_a_registered" --timeout 600 -p no:cacheprovider > gpurun_out/r03_d_parity.log 2>&1; echo "parity rc=$?"; tail -6 gpurun_out/r03_d_parity.log
timeout 1200 python -m pytest tests/test_plugin.py -q --timeout 900 -p no:cacheprovider -k "intree or dropin" > gpurun_out/r03_d_plugin.log 2>&1; echo "plugin rc=$?"; tail -8 gpurun_out/r03_d_plugin.log
timeout 600 python bench.py --legs plugin_host_path --steps 5 --warmup 2 > gpurun_out/r03_d_bench.json 2> gpurun_out/r03_d_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03_d_bench.json"))
    print(json.dumps(d["legs"]["plugin_host_path"]["parse_many_window_1MB"], indent=1)[:1500]); print("failed:", d.get("legs_failed"))
except Exception as e:
    print("no bench line:", e)
PY
tail -3 gpurun_out/r03_d_bench.err
