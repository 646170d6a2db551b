#!/bin/bash
# the string stream: parity tests of the string and tape paths, then the two bench legs, then per-kernel times
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "string or tape or stage2" > gpurun_out/r3i_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r3i_tests.log
timeout 600 python bench.py --legs next_f3_parse_strings,next_f3_tape --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err; echo "bench rc=$?"
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r3i_bench.json"))
    for k, v in d["legs"].items():
        print(k, json.dumps(v)[:1800])
    print("legs_failed", d.get("legs_failed"))
except Exception as e:
    print("no bench line:", e)
P
for k in twitter_like large_random; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_strs_$k -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $k > $GRAFT_REPO_ROOT/gpurun_out/prof_strs_$k.log 2>&1); echo "prof $k rc=$?"; tail -1 gpurun_out/prof_strs_$k.log
done
