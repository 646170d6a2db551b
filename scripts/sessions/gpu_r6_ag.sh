#!/bin/bash
# round 6, session AG: v17 = k_strs_resolve with four segments per thread (4 steps of three scans instead of 16)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v17.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or string or parse" > $O/r6ag_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6ag_pytest.log
timeout 900 python scripts/tape_ab.py v16=build/ab/libsjgpu_v16.so v17=build/ab/libsjgpu_v17.so > $O/r6ag_tape_ab.txt 2> $O/r6ag_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ag_tape_ab.txt; tail -3 $O/r6ag_tape_ab.err
