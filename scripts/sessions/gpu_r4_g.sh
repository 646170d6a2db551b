#!/bin/bash
# round 4, seventh GPU session: the container kinds as a prefix scan -- k_tok_apply checks the rules, no ctx, no depth array, k_tape_rules only for deep documents
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider -k "tape or string or stage2 or plugin" > gpurun_out/r4g_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4g_pytest_gpu.log
timeout 900 python bench.py --legs next_f3_tape > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4g_bench.json"))
t = d["legs"]["next_f3_tape"]
print("tape", {k: (t[k]["gpu_ms_per_call"], t[k]["roofline"]["frac"]) for k in ("twitter_like", "large_random")})
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4g_tape_tw -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like > $GRAFT_REPO_ROOT/gpurun_out/prof_r4g_tape_tw.log 2>&1); echo "tape trace rc=$?"
python3 scripts/rocpd_summary.py gpurun_out/prof_r4g_tape_tw/t_results.db | head -14 | cut -c1-110
