#!/bin/bash
# round 6, session AN: v22 = k_strs_write stores every byte at the lane's running offset (no dump byte: 8 -> 4 vector instructions per byte), k_strs_resolve with two barriers per tile
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v22.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse or string or strs" > $O/r6an_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6an_pytest.log
timeout 900 python scripts/tape_ab.py v20=build/ab/libsjgpu_v20.so v22=build/ab/libsjgpu_v22.so > $O/r6an_tape_ab.txt 2> $O/r6an_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6an_tape_ab.txt; tail -3 $O/r6an_tape_ab.err
