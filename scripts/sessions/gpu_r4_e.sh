#!/bin/bash
# round 4, fifth GPU session: the tape after the reorder (strings before k_tok_apply, which writes the string and atom words), the string stream without a table
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "tape or string or stage2 or plugin or smoke or backslash or range" > gpurun_out/r4e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r4e_pytest_gpu.log
timeout 600 python __graft_entry__.py --smoke > gpurun_out/r4e_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4e_smoke.log
timeout 900 python bench.py --legs next_f3_tape,next_f3_parse_strings > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4e_bench.json"))
t = d["legs"]["next_f3_tape"]
print("tape", {k: (t[k]["gpu_ms_per_call"], t[k]["roofline"]["frac"]) for k in ("twitter_like", "large_random")})
print("strings", d["legs"]["next_f3_parse_strings"]["value"], d["legs"]["next_f3_parse_strings"]["roofline"]["frac"])
print("failed", d.get("legs_failed"))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4e_tape_tw -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like > $GRAFT_REPO_ROOT/gpurun_out/prof_r4e_tape_tw.log 2>&1); echo "tape trace rc=$?"
python3 scripts/rocpd_summary.py gpurun_out/prof_r4e_tape_tw/t_results.db | head -24 | cut -c1-110
