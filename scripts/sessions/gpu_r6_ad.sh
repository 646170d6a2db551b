#!/bin/bash
# round 6, session AD: what k_tape_match waits for (memory-side counters) + the GPU test of the staged front's special paths
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 800 -p no:cacheprovider -k "staged_token_front" > $O/r6ad_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6ad_pytest.log
bash scripts/gpu_pmc_cmd.sh r6ad_lr "sq2 tcp ta tcc tcc2" -- python $GRAFT_REPO_ROOT/scripts/tape_once.py large_random > $O/r6ad_pmc.log 2>&1
python - <<PY
import json
s = json.load(open("$O/pmc_r6ad_lr/summary.json"))
for k, v in s.items():
    if "k_tape_match" in k or "k_radix_scatter" in k or "k_tok_apply" in k:
        print(k[:60]); print("   ", {c: float("%.4g" % x) for c, x in v.items()})
PY
