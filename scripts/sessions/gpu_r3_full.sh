#!/bin/bash
# the round-end sequence the driver runs, plus a rocprof kernel trace of the default bench: smoke, GPU tests, bench
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py --smoke > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r03_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r03_bench_default.err
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
