#!/bin/bash
# round 6, session AP: v23 = v23 + k_strs_resolve with the summaries of four tiles in flight
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v23.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse or string or strs" > $O/r6ap_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6ap_pytest.log
timeout 900 python scripts/tape_ab.py v22=build/ab/libsjgpu_v22.so v23=build/ab/libsjgpu_v23.so > $O/r6ap_tape_ab.txt 2> $O/r6ap_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ap_tape_ab.txt; tail -3 $O/r6ap_tape_ab.err
