#!/bin/bash
# round 6, session F: the token stream on the single-pass road -- parity on both roads, then what each road costs
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "token or real_rccl or tape" > $O/r6f_pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/r6f_pytest.log | tail -3
timeout 1500 python scripts/tokens_roads.py > $O/r6f_tokens_roads.txt 2> $O/r6f_tokens_roads.err; echo "roads rc=$?"; cat $O/r6f_tokens_roads.txt; tail -3 $O/r6f_tokens_roads.err
