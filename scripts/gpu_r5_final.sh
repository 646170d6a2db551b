#!/bin/bash
# round 5, the round-end sequence: smoke, the whole GPU tier, the default bench line, a rocprofv3 kernel trace of the same command, the NDJSON line at N = 1
# (--workload amazon_ndjson: the single-rank point of the curve the 8-GPU run continues), the N = 2 dry run (bench.py launching itself; gloo, both ranks on
# the one device), counter passes (FETCH_SIZE / WRITE_SIZE / SQ_* in separate runs) over the headline, the NDJSON scan and the scan with the token stream
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python __graft_entry__.py --smoke > $O/r05_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r05_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1400 -p no:cacheprovider > $O/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r05_pytest_gpu.log
timeout 900 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05_bench_final -o b -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r05_bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_bench_profiled.err); echo "profiled bench rc=$?"
timeout 600 python bench.py --workload amazon_ndjson > $O/r05_bench_ndjson_n1.json 2> $O/r05_bench_ndjson_n1.err; echo "ndjson n1 rc=$?"
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > $O/r05_bench_n2_dry.json 2> $O/r05_bench_n2_dry.err; echo "n2 dry rc=$?"
bash scripts/gpu_pmc.sh "--op stage1" r05_headline "fetch write sq1 sq2" > $O/r05_pmc_headline.log 2>&1; echo "pmc headline rc=$?"
bash scripts/gpu_pmc.sh "--op stage1 --workload amazon_ndjson" r05_ndjson "fetch write sq1 sq2" > $O/r05_pmc_ndjson.log 2>&1; echo "pmc ndjson rc=$?"
bash scripts/gpu_pmc.sh "--op minify" r05_minify "fetch write sq1" > $O/r05_pmc_minify.log 2>&1; echo "pmc minify rc=$?"
bash scripts/gpu_pmc.sh "--op stage1 --workload escape_heavy" r05_escape "fetch write sq1" > $O/r05_pmc_escape.log 2>&1; echo "pmc escape rc=$?"
bash scripts/gpu_pmc.sh "--op validate_utf8" r05_validate "fetch sq1" > $O/r05_pmc_validate.log 2>&1; echo "pmc validate rc=$?"
bash scripts/gpu_pmc_cmd.sh r05_tokens "fetch write sq1" -- python $GRAFT_REPO_ROOT/scripts/tokens_once.py amazon_ndjson 1073741824 > $O/r05_pmc_tokens.log 2>&1; echo "pmc tokens rc=$?"
python3 scripts/pmc_table.py $O/pmc_r05_headline $O/pmc_r05_ndjson $O/pmc_r05_minify $O/pmc_r05_escape $O/pmc_r05_validate $O/pmc_r05_tokens > $O/r05_pmc_tables.txt 2>&1; tail -40 $O/r05_pmc_tables.txt | cut -c1-150
python3 - <<'PY'
import json
def last_line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
try:
    d = last_line("gpurun_out/r05_bench_default.json")
    print({k: d[k] for k in ("metric", "value", "ms_per_step", "value_first_reps")}, d["roofline"]["frac"], d["roofline"]["kernel"], "failed:", d.get("legs_failed"))
    for k, v in d.get("legs", {}).items():
        r = v.get("roofline") if isinstance(v, dict) else None
        print(k, (r or {}).get("frac"), v.get("value") if isinstance(v, dict) else None)
    t = d["legs"]["next_f3_tape"]
    print("tape", {k: (t[k]["gpu_ms_per_call"], t[k]["roofline"]["frac"]) for k in ("twitter_like", "large_random")})
    x = last_line("gpurun_out/r05_bench_ndjson_n1.json")
    print("ndjson n1", x["value"], x["roofline"]["frac"], x.get("cpu_baseline_threads", {}).get("value"))
    n2 = last_line("gpurun_out/r05_bench_n2_dry.json")
    print("n2", n2["value"], n2["n_gpus"], n2.get("n1_same_workload_GBps"), n2.get("scaling_efficiency"), n2.get("parity", {}).get("all_ranks_ok"), str(n2.get("index_concat"))[:80])
except Exception as e:
    print("no bench line:", e)
PY
python3 scripts/rocpd_summary.py gpurun_out/prof_r05_bench_final/*/b_results.db gpurun_out/prof_r05_bench_final/b_results.db 2>/dev/null | head -40 | cut -c1-140
