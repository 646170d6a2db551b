#!/bin/bash
# round 5, session B: the sanitizer runs with their logs kept, the whole GPU tier on the kernels without agent-scope fences, the token stream,
# the self-cleaning workspace against the memset of rounds 1-4 (same box, interleaved)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 bash scripts/sanitize.sh run > $O/r5b_sanitize.log 2>&1; echo "sanitize rc=$?"; tail -2 $O/r5b_sanitize.log
cp build/san/address.log $O/r5b_asan.log 2>/dev/null; cp build/san/thread.log $O/r5b_tsan.log 2>/dev/null
grep -n "ERROR: AddressSanitizer\|SUMMARY" $O/r5b_asan.log | head -5
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --deselect tests/test_plugin.py::test_sanitizers_over_the_host_shim > $O/r5b_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r5b_pytest_gpu.log
for i in 1 2; do
  SJGPU_FUSED_MEMSET=1 timeout 300 python bench.py --legs none --no-cpu-baseline > $O/r5b_ab_memset_$i.json 2>> $O/r5b_ab.err
  timeout 300 python bench.py --legs none --no-cpu-baseline > $O/r5b_ab_clean_$i.json 2>> $O/r5b_ab.err
done
timeout 900 python bench.py > $O/r5b_bench_default.json 2> $O/r5b_bench_default.err; echo "bench rc=$?"
python3 - <<'PY'
import json, glob
def last_line(path):
    try:
        return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"error": repr(e)}
for f in sorted(glob.glob("gpurun_out/r5b_ab_*.json")):
    x = last_line(f)
    print(f, x.get("value"), x.get("ms_per_step"), (x.get("roofline") or {}).get("gpu_ms_per_step"), (x.get("roofline") or {}).get("frac"), x.get("value_first_reps"))
d = last_line("gpurun_out/r5b_bench_default.json")
try:
    print({k: d[k] for k in ("metric", "value", "ms_per_step", "value_first_reps")}, d["roofline"]["frac"], d["roofline"]["gpu_ms_per_step"], "failed:", d.get("legs_failed"))
    for k, v in d.get("legs", {}).items():
        r = v.get("roofline") if isinstance(v, dict) else None
        print(" ", k, (r or {}).get("frac"), v.get("value") if isinstance(v, dict) else None, (r or {}).get("gpu_ms_per_step"))
    t = d["legs"]["next_f3_tape"]
    print("  tape", {k: (t[k]["gpu_ms_per_call"], t[k]["roofline"]["frac"]) for k in t})
    print("  twitter", d["legs"]["config0_twitter_json"]["device_resident"]["gpu_us_per_call"], d["legs"]["config0_twitter_json"]["host_buffers"]["us_per_call"])
    print("  tokens", json.dumps(d["legs"]["next_f3_depth_scan"].get("with_token_stream"))[:900])
except Exception as e:
    print("no bench line:", e, str(d)[:300])
PY
