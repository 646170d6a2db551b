#!/bin/bash
# round 6, session AB: v15 = the token tables built once per call by k_tape_init instead of by every workgroup
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v15.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6ab_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6ab_pytest.log
timeout 900 python scripts/tape_ab.py v14=build/ab/libsjgpu_v14.so v15=build/ab/libsjgpu_v15.so > $O/r6ab_tape_ab.txt 2> $O/r6ab_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ab_tape_ab.txt; tail -3 $O/r6ab_tape_ab.err
