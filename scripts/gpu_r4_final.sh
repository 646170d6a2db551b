#!/bin/bash
# round-end sequence: smoke, the whole GPU tier, the default bench line, a rocprofv3 kernel trace of the same command, the N = 2 dry run
# (bench.py launching itself; gloo, both ranks on the one device of the box)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py --smoke > gpurun_out/r04_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04_bench_final -o b -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04_bench_profiled.err); echo "profiled bench rc=$?"
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > gpurun_out/r04_bench_n2_dry.json 2> gpurun_out/r04_bench_n2_dry.err; echo "n2 dry rc=$?"
python3 - <<'PY'
import json
def last_line(path):  # the bench line is the last line that starts with a brace (library banners may precede it)
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
try:
    d = last_line("gpurun_out/r04_bench_default.json")
    print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel"], "failed:", d.get("legs_failed"))
    for k, v in d.get("legs", {}).items():
        r = v.get("roofline") if isinstance(v, dict) else None
        print(k, (r or {}).get("frac"), v.get("value") if isinstance(v, dict) else None)
    t = d["legs"]["next_f3_tape"]
    print("tape", {k: (t[k]["gpu_ms_per_call"], t[k]["roofline"]["frac"]) for k in ("twitter_like", "large_random")})
    n2 = last_line("gpurun_out/r04_bench_n2_dry.json")
    print("n2", n2["value"], n2["n_gpus"], n2.get("parity"), n2.get("n_ranks_seen_by_rccl"), str(n2.get("index_concat"))[:80])
except Exception as e:
    print("no bench line:", e)
PY
python3 scripts/rocpd_summary.py gpurun_out/prof_r04_bench_final/*/b_results.db gpurun_out/prof_r04_bench_final/b_results.db 2>/dev/null | head -60 | cut -c1-140
