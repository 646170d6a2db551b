#!/bin/bash
# round 5, session C: the sanitizer test as the driver will run it, the token stream's leg of the bench, the pinned padded_string against fresh pageable
# memory (with and without the document's buffers page-locked), the N = 2 dry run of the new line
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_plugin.py -m gpu -q --timeout 900 -p no:cacheprovider > $O/r5c_pytest_plugin.log 2>&1; echo "pytest plugin rc=$?"; tail -3 $O/r5c_pytest_plugin.log
timeout 900 python bench.py > $O/r5c_bench_default.json 2> $O/r5c_bench_default.err; echo "bench rc=$?"
timeout 600 build/tests/plugin_test --bench-pinned 268435456 > $O/r5c_plugin_pinned.log 2>&1; echo "plugin pinned rc=$?"; grep pinned_bench $O/r5c_plugin_pinned.log
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > $O/r5c_bench_n2_dry.json 2> $O/r5c_bench_n2_dry.err; echo "n2 dry rc=$?"
python3 - <<'PY'
import json
def last_line(path):
    try:
        return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"error": repr(e)}
d = last_line("gpurun_out/r5c_bench_default.json")
try:
    print({k: d[k] for k in ("metric", "value", "ms_per_step", "value_first_reps")}, d["roofline"]["frac"], d["roofline"]["gpu_ms_per_step"], "failed:", d.get("legs_failed"))
    for k, v in d.get("legs", {}).items():
        r = v.get("roofline") if isinstance(v, dict) else None
        print(" ", k, (r or {}).get("frac"), v.get("value") if isinstance(v, dict) else None, (r or {}).get("kernel_ms_slots"))
    print("  tokens", json.dumps(d["legs"]["next_f3_depth_scan"].get("with_token_stream"))[:1200])
except Exception as e:
    print("no bench line:", e, str(d)[:300])
x = last_line("gpurun_out/r5c_bench_n2_dry.json")
print("n2", x.get("value"), x.get("n_gpus"), x.get("n1_same_workload_GBps"), x.get("scaling_efficiency"), (x.get("config") or {}).get("workload"), str(x.get("config3_ndjson_sharded"))[:300])
PY
