#!/bin/bash
# round 6, session AJ: v19 = one store site for a token's first value word in k_tok_apply (strings, atoms, numbers), the first digit of a number kept in a register
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v19.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6aj_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6aj_pytest.log
timeout 900 python scripts/tape_ab.py v18=build/ab/libsjgpu_v18.so v19=build/ab/libsjgpu_v19.so > $O/r6aj_tape_ab.txt 2> $O/r6aj_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6aj_tape_ab.txt; tail -3 $O/r6aj_tape_ab.err
