#!/bin/bash
# round 4, session h: the string stream without the \u walk (mask algebra + patch list) -- parity of the string / tape tests, the two legs, the kernel trace
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r4h}
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "tape or string or stage2" > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${T}_pytest_gpu.log
timeout 900 python bench.py --legs next_f3_tape,next_f3_parse_strings > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("gpurun_out/${T}_bench.json"))
t = d["legs"]["next_f3_tape"]
print("tape", {k: (t[k]["gpu_ms_per_call"], t[k]["roofline"]["frac"]) for k in ("twitter_like", "large_random")})
s = d["legs"]["next_f3_parse_strings"]
print("strings", s.get("gpu_ms_per_call"), s["roofline"]["frac"])
print("legs_failed", d.get("legs_failed"))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_tape_tw -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like > $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_tape_tw.log 2>&1); echo "tape trace rc=$?"
python3 scripts/rocpd_summary.py gpurun_out/prof_${T}_tape_tw/t_results.db | head -24 | cut -c1-110
