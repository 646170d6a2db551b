#!/bin/bash
# round 6, session AQ: v24 = v24 + k_strs_resolve as a workgroup per tile (totals for both states published, read by the tiles behind)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v24.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse or string or strs" > $O/r6aq_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6aq_pytest.log
timeout 900 python scripts/tape_ab.py v22=build/ab/libsjgpu_v22.so v24=build/ab/libsjgpu_v24.so > $O/r6aq_tape_ab.txt 2> $O/r6aq_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6aq_tape_ab.txt; tail -3 $O/r6aq_tape_ab.err
bash scripts/sessions/gpu_r6_ao.sh 2>&1 | grep "k_strs_\|rc=" | cut -c1-100
