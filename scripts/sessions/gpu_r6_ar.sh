#!/bin/bash
# round 6, session AR: v25 = v24 + k_strs_init inside k_strs_count, k_strs_decide inside k_strs_resolve (the tape s call), both results read back in one copy: 18 -> 15 launches
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v25.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse or string or strs" > $O/r6ar_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6ar_pytest.log
timeout 900 python scripts/tape_ab.py v24=build/ab/libsjgpu_v24.so v25=build/ab/libsjgpu_v25.so > $O/r6ar_tape_ab.txt 2> $O/r6ar_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ar_tape_ab.txt; tail -3 $O/r6ar_tape_ab.err
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 800 -p no:cacheprovider > $O/r6ar_pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/r6ar_pytest_all.log
