#!/bin/bash
# round 5, session S: which leg in front of config3 makes its sustained figure 0.38 ms (alone in a process: 0.30)?
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
run() { tag=$1; legs=$2; timeout 600 python bench.py --no-cpu-baseline --legs $legs > $O/r5s_$tag.json 2> $O/r5s_$tag.err; python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r5s_$tag.json").read().splitlines() if l.startswith("{")][-1])
for k, v in d.get("legs", {}).items():
    if isinstance(v, dict) and "roofline" in v and "ms_per_step" in v:
        print("$tag", k, v["ms_per_step"], v.get("first_reps_ms_per_step"), v["roofline"].get("kernel_ms_slots"))
    elif isinstance(v, dict) and "error" in v:
        print("$tag", k, "ERROR", v["error"][:200])
PY
}
run only config3_amazon_ndjson,config4_escape_heavy
run tape next_f3_tape,config3_amazon_ndjson,config4_escape_heavy
run plugin plugin_host_path,config3_amazon_ndjson,config4_escape_heavy
run c2 config2_minify,config2_validate_utf8,config3_amazon_ndjson,config4_escape_heavy
run lists next_f2_finish_device,next_f3_depth_scan,next_f3_parse_strings,config3_amazon_ndjson,config4_escape_heavy
