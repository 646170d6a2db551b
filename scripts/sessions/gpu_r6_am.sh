#!/bin/bash
# round 6, session AM: v21 = steps of four rows in every staged group that holds four whole rows from a multiple of four on; groups end on multiples of four rows
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v21.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6am_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6am_pytest.log
timeout 900 python scripts/tape_ab.py v20=build/ab/libsjgpu_v20.so v21=build/ab/libsjgpu_v21.so > $O/r6am_tape_ab.txt 2> $O/r6am_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6am_tape_ab.txt; tail -3 $O/r6am_tape_ab.err
