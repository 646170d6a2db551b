#!/bin/bash
# round 5, session X: k_strs_write compacts dword by dword (the tree) against the library before (build/ab/libsjgpu_S5tree.so): the string and tape legs of
# bench.py in two processes each, a kernel trace of the tape, the string / tape parity tests
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
show() { python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r5x_$1.json").read().splitlines() if l.startswith("{")][-1])
s = d["legs"]["next_f3_parse_strings"]; t = d["legs"]["next_f3_tape"]
print("$1 strings", s.get("gpu_ms_per_call"), s.get("first_reps_ms_per_call"), "tape", {k: (v["gpu_ms_per_call"], v["first_reps_ms_per_call"]) for k, v in t.items()}, d.get("legs_failed"))
PY
}
for i in 1 2; do
  SJGPU_LIB=$GRAFT_REPO_ROOT/build/ab/libsjgpu_S5tree.so timeout 600 python bench.py --legs next_f3_parse_strings,next_f3_tape > $O/r5x_before$i.json 2> $O/r5x_before$i.err; show before$i
  timeout 600 python bench.py --legs next_f3_parse_strings,next_f3_tape > $O/r5x_tree$i.json 2> $O/r5x_tree$i.err; show tree$i
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_r5x_tape -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like 268435456 > $O/r5x_tape_once.log 2>&1); python3 scripts/rocpd_summary.py $O/prof_r5x_tape/t_results.db $O/prof_r5x_tape/*/t_results.db 2>/dev/null | head -30 | cut -c1-130
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "string or tape or stage2 or strings or raw_key" > $O/r5x_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r5x_pytest.log
