#!/bin/bash
# round 6, session AE: v16 = the sort's histogram tables as wide as the tiles that hold elements (the scans run over 128 x live entries, dead tiles write nothing)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v16.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6ae_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6ae_pytest.log
timeout 900 python scripts/tape_ab.py v15=build/ab/libsjgpu_v15.so v16=build/ab/libsjgpu_v16.so > $O/r6ae_tape_ab.txt 2> $O/r6ae_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ae_tape_ab.txt; tail -3 $O/r6ae_tape_ab.err
