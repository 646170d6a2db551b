#!/bin/bash
# round 5, session W: the last tree once more where it changed after the round-end sequence -- smoke, the N = 2 dry run (its legs start from an empty
# allocator now), the parity tests of the scan kernels
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python __graft_entry__.py --smoke > $O/r5w_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r5w_smoke.log
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-device --size 268435456 > $O/r05_bench_n2_dry.json 2> $O/r05_bench_n2_dry.err; echo "n2 dry rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "built_from or golden or straddling or quote_parity or adversarial or control_character or fuzz or streaming or plugin or reference_own or sanitizers or full_size_device_resident" > $O/r5w_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r5w_pytest.log
python3 - <<'PY'
import json
n2 = json.loads([l for l in open("gpurun_out/r05_bench_n2_dry.json").read().splitlines() if l.startswith("{")][-1])
print("n2", n2["value"], n2["n_gpus"], n2.get("n1_same_workload_GBps"), n2.get("scaling_efficiency"), n2.get("parity", {}).get("all_ranks_ok"))
nd = n2["config3_ndjson_sharded"]
print("n2 ndjson", nd.get("value_GBps"), nd.get("n1_same_workload_GBps"), nd.get("scaling_efficiency"), nd.get("sorted_global_positions"), nd.get("error"))
PY
