#!/bin/bash
# round 5, session T: the whole default line with and without the CPU baselines: where does config3's sustained figure go?
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
show() { python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r5t_$1.json").read().splitlines() if l.startswith("{")][-1])
print("$1 headline", d["ms_per_step"], d.get("first_reps_ms_per_step"))
for k, v in d.get("legs", {}).items():
    if isinstance(v, dict) and "roofline" in v and "ms_per_step" in v:
        print("$1", k, v["ms_per_step"], v.get("first_reps_ms_per_step"), v["roofline"].get("kernel_ms_slots"))
PY
}
timeout 600 python bench.py --no-cpu-baseline > $O/r5t_nocpu.json 2> $O/r5t_nocpu.err; show nocpu
timeout 900 python bench.py > $O/r5t_full.json 2> $O/r5t_full.err; show full
timeout 900 python bench.py --legs config3_amazon_ndjson,config4_escape_heavy > $O/r5t_c34cpu.json 2> $O/r5t_c34cpu.err; show c34cpu
