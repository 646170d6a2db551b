#!/bin/bash
# round 5, session A: everything new since round 4 on the real device -- the loop-back RCCL worlds (2, 3, 8 x 1 GiB), the plug-in's new cases,
# the self-cleaning single-pass workspace (A/B against the memset of rounds 1-4), the new bench line, the N = 2 dry run, the pinned padded_string
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python __graft_entry__.py --smoke > $O/r5a_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r5a_smoke.log
timeout 1500 python -m pytest tests/test_gpu_comm.py -m gpu -q -x --timeout 1400 -p no:cacheprovider > $O/r5a_pytest_comm.log 2>&1; echo "pytest comm rc=$?"; tail -3 $O/r5a_pytest_comm.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_comm.py > $O/r5a_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r5a_pytest_gpu.log
timeout 900 python bench.py > $O/r5a_bench_default.json 2> $O/r5a_bench_default.err; echo "bench rc=$?"
for i in 1 2; do
  SJGPU_FUSED_MEMSET=1 timeout 300 python bench.py --legs none --no-cpu-baseline > $O/r5a_ab_memset_$i.json 2>> $O/r5a_ab.err
  timeout 300 python bench.py --legs none --no-cpu-baseline > $O/r5a_ab_clean_$i.json 2>> $O/r5a_ab.err
done
timeout 600 python bench.py --workload amazon_ndjson > $O/r5a_bench_ndjson_n1.json 2> $O/r5a_bench_ndjson_n1.err; echo "ndjson n1 rc=$?"
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > $O/r5a_bench_n2_dry.json 2> $O/r5a_bench_n2_dry.err; echo "n2 dry rc=$?"
timeout 600 build/tests/plugin_test --bench-pinned 268435456 > $O/r5a_plugin_pinned.log 2>&1; echo "plugin pinned rc=$?"; grep pinned_bench $O/r5a_plugin_pinned.log
python3 - <<'PY'
import json, glob
def last_line(path):
    try:
        return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"error": repr(e)}
d = last_line("gpurun_out/r5a_bench_default.json")
try:
    print({k: d[k] for k in ("metric", "value", "ms_per_step", "value_first_reps")}, d["roofline"]["frac"], d["roofline"]["gpu_ms_per_step"], "failed:", d.get("legs_failed"))
    for k, v in d.get("legs", {}).items():
        r = v.get("roofline") if isinstance(v, dict) else None
        print(" ", k, (r or {}).get("frac"), v.get("value") if isinstance(v, dict) else None, (r or {}).get("gpu_ms_per_step"))
    t = d["legs"]["next_f3_tape"]
    print("  tape", {k: (t[k]["gpu_ms_per_call"], t[k]["roofline"]["frac"]) for k in t})
    print("  twitter", d["legs"]["config0_twitter_json"]["device_resident"]["gpu_us_per_call"], d["legs"]["config0_twitter_json"]["host_buffers"]["us_per_call"])
except Exception as e:
    print("no bench line:", e, str(d)[:300])
for f in sorted(glob.glob("gpurun_out/r5a_ab_*.json")):
    x = last_line(f)
    print(f, x.get("value"), x.get("ms_per_step"), (x.get("roofline") or {}).get("gpu_ms_per_step"), (x.get("roofline") or {}).get("frac"))
x = last_line("gpurun_out/r5a_bench_ndjson_n1.json")
print("ndjson n1", x.get("value"), (x.get("roofline") or {}).get("frac"), (x.get("cpu_baseline_threads") or {}).get("value"), x.get("config", {}).get("workload"))
x = last_line("gpurun_out/r5a_bench_n2_dry.json")
print("n2", x.get("value"), x.get("n_gpus"), x.get("n1_same_workload_GBps"), x.get("scaling_efficiency"), x.get("parity"), str(x.get("config3_ndjson_sharded"))[:600])
PY
