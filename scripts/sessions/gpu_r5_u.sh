#!/bin/bash
# round 5, session U: the default line with an empty allocator in front of every leg, twice
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
show() { python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r5u_$1.json").read().splitlines() if l.startswith("{")][-1])
print("$1 headline", d["ms_per_step"], d.get("first_reps_ms_per_step"), d["roofline"]["frac"])
for k, v in d.get("legs", {}).items():
    if isinstance(v, dict) and "roofline" in v and "ms_per_step" in v:
        print("$1", k, v["ms_per_step"], v.get("first_reps_ms_per_step"), v["roofline"].get("kernel_ms_slots"), v["roofline"]["frac"])
PY
}
timeout 900 python bench.py > $O/r5u_full1.json 2> $O/r5u_full1.err; show full1
timeout 900 python bench.py > $O/r5u_full2.json 2> $O/r5u_full2.err; show full2
