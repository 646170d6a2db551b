#!/bin/bash
# round 6, the round-end sequence: smoke, the whole GPU tier, the driver's own bench command (one compact line + bench_detail.json), a rocprofv3 kernel trace of
# the same command, the NDJSON line at N = 1, the N = 2 dry run (gloo, both ranks on the one device), counter passes (FETCH_SIZE / WRITE_SIZE / SQ_* in separate
# runs) over the headline, NDJSON, minify, escape_heavy, validate_utf8 and -- fresh since round 3 -- the tape of both 256 MiB documents
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
T=${1:-r06}
timeout 600 python __graft_entry__.py --smoke > $O/${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${T}_smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 1400 -p no:cacheprovider > $O/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/${T}_pytest_gpu.log | tail -2
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; echo "bench rc=$?"; tail -n 1 $O/${T}_bench_default.json | wc -c
cp bench_detail.json $O/${T}_bench_detail.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_bench_final -o b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_profiled.err); echo "profiled bench rc=$?"
cp bench_detail.json $O/${T}_bench_profiled_detail.json
timeout 600 python bench.py --workload amazon_ndjson > $O/${T}_bench_ndjson_n1.json 2> $O/${T}_bench_ndjson_n1.err; echo "ndjson n1 rc=$?"
cp bench_detail.json $O/${T}_bench_ndjson_n1_detail.json
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > $O/${T}_bench_n2_dry.json 2> $O/${T}_bench_n2_dry.err; echo "n2 dry rc=$?"; tail -n 1 $O/${T}_bench_n2_dry.json | wc -c
cp bench_detail.json $O/${T}_bench_n2_dry_detail.json
bash scripts/gpu_pmc.sh "--op stage1" ${T}_headline "fetch write sq1 sq2" > $O/${T}_pmc_headline.log 2>&1; echo "pmc headline rc=$?"
bash scripts/gpu_pmc.sh "--op stage1 --workload amazon_ndjson" ${T}_ndjson "fetch write sq1 sq2" > $O/${T}_pmc_ndjson.log 2>&1; echo "pmc ndjson rc=$?"
bash scripts/gpu_pmc.sh "--op minify" ${T}_minify "fetch write sq1" > $O/${T}_pmc_minify.log 2>&1; echo "pmc minify rc=$?"
bash scripts/gpu_pmc.sh "--op stage1 --workload escape_heavy" ${T}_escape "fetch write sq1" > $O/${T}_pmc_escape.log 2>&1; echo "pmc escape rc=$?"
bash scripts/gpu_pmc.sh "--op validate_utf8" ${T}_validate "fetch sq1" > $O/${T}_pmc_validate.log 2>&1; echo "pmc validate rc=$?"
bash scripts/gpu_pmc_cmd.sh ${T}_tape_tw "fetch write sq1" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py twitter_like 268435456 > $O/${T}_pmc_tape_tw.log 2>&1; echo "pmc tape twitter rc=$?"
bash scripts/gpu_pmc_cmd.sh ${T}_tape_lr "fetch write sq1" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py large_random 268435456 > $O/${T}_pmc_tape_lr.log 2>&1; echo "pmc tape large_random rc=$?"
python3 scripts/pmc_table.py $O/pmc_${T}_headline $O/pmc_${T}_ndjson $O/pmc_${T}_minify $O/pmc_${T}_escape $O/pmc_${T}_validate $O/pmc_${T}_tape_tw $O/pmc_${T}_tape_lr > $O/${T}_pmc_tables.txt 2>&1; tail -80 $O/${T}_pmc_tables.txt | cut -c1-150
python3 - <<PY
import json
def last_line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
try:
    d = last_line("gpurun_out/${T}_bench_default.json")
    print({k: d[k] for k in ("metric", "value", "ms_per_step", "value_first_reps")}, d["roofline"]["frac"], d["roofline"]["kernel"], "failed:", d.get("legs_failed"))
    for k, v in d.get("legs", {}).items():
        print(k, v)
    x = last_line("gpurun_out/${T}_bench_ndjson_n1.json")
    print("ndjson n1", x["value"], x["roofline"]["frac"], x.get("cpu_baseline_threads", {}).get("value"))
    n2 = last_line("gpurun_out/${T}_bench_n2_dry.json")
    print("n2", n2["value"], n2["n_gpus"], n2.get("n1_same_workload_GBps"), n2.get("scaling_efficiency"), n2.get("parity"), str(n2.get("index_concat"))[:80])
except Exception as e:
    print("no bench line:", e)
PY
python3 scripts/rocpd_summary.py gpurun_out/prof_${T}_bench_final/*/b_results.db gpurun_out/prof_${T}_bench_final/b_results.db 2>/dev/null | head -50 | cut -c1-140
