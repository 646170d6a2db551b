#!/bin/bash
# round 6, session AX: v32 = v31 + one patch entry per RUN of listed bytes, what a byte becomes read off the document by whoever works the entry off
# GPU parity file, A/B and trace
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider > $O/r6ax_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6ax_pytest.log
timeout 900 python scripts/tape_ab.py v31=build/ab/libsjgpu_v31.so v32=build/ab/libsjgpu_v32.so > $O/r6ax_tape_ab.txt 2> $O/r6ax_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6ax_tape_ab.txt; tail -3 $O/r6ax_tape_ab.err
for v in v32; do
  for kind in twitter_like large_random; do
    (cd /tmp && SJGPU_LIB=$R/build/ab/libsjgpu_$v.so timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r6ax_${v}_$kind -o t -- python $R/scripts/tape_once.py $kind 268435456 > $O/r6ax_${v}_$kind.log 2>&1); echo "$v $kind rc=$?"
    python3 scripts/rocpd_summary.py gpurun_out/prof_r6ax_${v}_$kind/t_results.db 2>/dev/null | grep "k_strs_write" | cut -c1-100
  done
done
