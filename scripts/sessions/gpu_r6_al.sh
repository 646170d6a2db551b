#!/bin/bash
# round 6, session AL: v20 = a lane takes four consecutive tokens per step when a full wave is staged at once (dense text), the second sweep through the 8-byte window (99 VGPRs)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v20.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6al_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6al_pytest.log
timeout 900 python scripts/tape_ab.py v19=build/ab/libsjgpu_v19.so v20=build/ab/libsjgpu_v20.so > $O/r6al_tape_ab.txt 2> $O/r6al_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6al_tape_ab.txt; tail -3 $O/r6al_tape_ab.err
