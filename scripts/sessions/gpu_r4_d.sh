#!/bin/bash
# round 4, fourth GPU session: the tape after the opens fold, zero-candidate segments, the 2 MiB stage-2 threshold; counters of the tape kernels
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r4d_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4d_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r4d_bench.json 2> gpurun_out/r4d_bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4d_bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_slots"])
for k, v in d.get("legs", {}).items():
    if isinstance(v, dict) and "roofline" in v:
        print(k, v.get("value"), v.get("ms_per_step"), v["roofline"].get("frac"), v["roofline"].get("kernel_ms_slots"))
t = d["legs"]["next_f3_tape"]
print("tape", {k: (t[k]["gpu_ms_per_call"], t[k]["roofline"]["frac"]) for k in ("twitter_like", "large_random")})
for k, v in d["legs"]["plugin_host_path"]["dom_parse"]["sizes"].items():
    print(k, v.get("reference_parse_ms"), v.get("road_a_gpu_stage1_plus_reference_stage2_ms"), v.get("road_b_sjgpu_parse_ms"))
print("failed", d.get("legs_failed"))
PY
timeout 900 bash scripts/gpu_pmc_cmd.sh r4_tape_tw "fetch write sq1 sq2" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like > gpurun_out/pmc_r4_tape_tw.txt 2>&1; tail -3 gpurun_out/pmc_r4_tape_tw.txt
timeout 600 bash scripts/gpu_pmc_cmd.sh r4_tape_lr "fetch write" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py large_random > gpurun_out/pmc_r4_tape_lr.txt 2>&1; tail -3 gpurun_out/pmc_r4_tape_lr.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4_tape_tw -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like > $GRAFT_REPO_ROOT/gpurun_out/prof_r4_tape_tw.log 2>&1); echo "tape trace rc=$?"
bash scripts/gpu_pmc.sh "--workload escape_heavy" r4d_escape "fetch write" 2>&1 | tail -12
