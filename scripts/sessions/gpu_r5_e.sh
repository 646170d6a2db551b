#!/bin/bash
# round 5, session E: the list passes after the branch-free classification (kernel traces with and without the token stream), the whole GPU tier, the bench line
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/r5e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r5e_pytest_gpu.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_r5e_tok_amazon_ndjson -o t -- python $GRAFT_REPO_ROOT/scripts/tokens_once.py amazon_ndjson 1073741824 > $O/r5e_tok_amazon_ndjson.log 2>&1); echo "trace ndjson rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_r5e_tok_twitter -o t -- python $GRAFT_REPO_ROOT/scripts/tokens_once.py twitter_like 268435456 > $O/r5e_tok_twitter.log 2>&1); echo "trace twitter rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/prof_r5e_tok_large_random -o t -- python $GRAFT_REPO_ROOT/scripts/tokens_once.py large_random 268435456 > $O/r5e_tok_large_random.log 2>&1); echo "trace large_random rc=$?"
timeout 900 python bench.py > $O/r5e_bench_default.json 2> $O/r5e_bench_default.err; echo "bench rc=$?"
python3 scripts/rocpd_summary.py $O/prof_r5e_tok_amazon_ndjson/t_results.db 2>/dev/null | head -16 | cut -c1-130
python3 - <<'PY'
import json
def last_line(path):
    try:
        return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"error": repr(e)}
d = last_line("gpurun_out/r5e_bench_default.json")
try:
    print({k: d[k] for k in ("metric", "value", "ms_per_step", "value_first_reps")}, d["roofline"]["frac"], d["roofline"]["gpu_ms_per_step"], "failed:", d.get("legs_failed"))
    for k, v in d.get("legs", {}).items():
        r = v.get("roofline") if isinstance(v, dict) else None
        print(" ", k, (r or {}).get("frac"), v.get("value") if isinstance(v, dict) else None, (r or {}).get("kernel_ms_slots"))
    print("  tokens", json.dumps(d["legs"]["next_f3_depth_scan"].get("with_token_stream"))[:1000])
    t = d["legs"]["next_f3_tape"]
    print("  tape", {k: (t[k]["gpu_ms_per_call"], t[k]["first_reps_ms_per_call"], t[k]["roofline"]["frac"], t[k].get("with_token_stream")) for k in t})
except Exception as e:
    print("no bench line:", e, str(d)[:300])
PY
