#!/bin/bash
# round 6, session AH: v18 = no second look at the digits of a number of at most 19 digits, 32-bit exponent arithmetic in decimal_to_binary64
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v18.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6ah_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6ah_pytest.log
timeout 900 python scripts/tape_ab.py v16=build/ab/libsjgpu_v16.so v18=build/ab/libsjgpu_v18.so > $O/r6ah_tape_ab.txt 2> $O/r6ah_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ah_tape_ab.txt; tail -3 $O/r6ah_tape_ab.err
