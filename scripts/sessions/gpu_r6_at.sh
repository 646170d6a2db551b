#!/bin/bash
# round 6, session AT: v26 = v25 + the kept bytes of \\u escapes computed from one 16-byte window of the document (one round trip instead of a chain of two or three)
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cp build/ab/libsjgpu_v26.so simdjson_amd/lib/libsjgpu.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse or string or strs" > $O/r6at_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6at_pytest.log
timeout 900 python scripts/tape_ab.py v25=build/ab/libsjgpu_v25.so v26=build/ab/libsjgpu_v26.so > $O/r6at_tape_ab.txt 2> $O/r6at_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6at_tape_ab.txt; tail -3 $O/r6at_tape_ab.err
for kind in twitter_like; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r6at_$kind -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $kind 268435456 > $O/r6at_$kind.log 2>&1); echo "$kind rc=$?"
  python3 scripts/rocpd_summary.py gpurun_out/prof_r6at_$kind/t_results.db 2>/dev/null | grep "k_strs_write\|k_strs_count" | cut -c1-100
done
