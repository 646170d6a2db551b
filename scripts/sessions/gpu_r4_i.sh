#!/bin/bash
# round 4, session i: the boundary search walks the list from its end and stops (finish), the depth scan in one pass -- parity, the two legs, a kernel trace
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r4i}
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "finish or depth or stream or plugin" > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${T}_pytest_gpu.log
timeout 900 python bench.py --legs next_f2_finish_device,next_f3_depth_scan > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench.err
python3 - <<PY
import json
d = json.load(open("gpurun_out/${T}_bench.json"))
for k in ("next_f2_finish_device", "next_f3_depth_scan"):
    v = d["legs"][k]
    print(k, v.get("ms_per_call"), v.get("gpu_ms_per_call"), v["roofline"]["frac"], v.get("host_finish"))
print("legs_failed", d.get("legs_failed"))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T} -o t -- python $GRAFT_REPO_ROOT/bench.py --legs next_f2_finish_device,next_f3_depth_scan > $GRAFT_REPO_ROOT/gpurun_out/prof_${T}.log 2>&1); echo "trace rc=$?"
python3 scripts/rocpd_summary.py gpurun_out/prof_${T}/t_results.db | head -16 | cut -c1-130
