#!/bin/bash
# round 6, session AV: k_strs_write with a window for HALF a chunk's worst case (29 KB of LDS per workgroup instead of 49.5: chunks whose output does not fit take two
# passes of 32 lanes): v29 = 105 VGPRs (four waves per SIMD), v30 = asked for five waves per SIMD (96 VGPRs, no scratch); parity first (the new documents of
# test_string_stream_takes_valid_documents_and_only_those take the two-pass road), then the tape A/B and k_strs_write in the trace
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse or string or strs" > $O/r6av_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r6av_pytest.log
timeout 900 python scripts/tape_ab.py v26=build/ab/libsjgpu_v26.so v29=build/ab/libsjgpu_v29.so v30=build/ab/libsjgpu_v30.so > $O/r6av_tape_ab.txt 2> $O/r6av_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6av_tape_ab.txt; tail -3 $O/r6av_tape_ab.err
for v in v26 v29 v30; do
  for kind in twitter_like large_random; do
    (cd /tmp && SJGPU_LIB=$R/build/ab/libsjgpu_$v.so timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r6av_${v}_$kind -o t -- python $R/scripts/tape_once.py $kind 268435456 > $O/r6av_${v}_$kind.log 2>&1); echo "$v $kind rc=$?"
    python3 scripts/rocpd_summary.py gpurun_out/prof_r6av_${v}_$kind/t_results.db 2>/dev/null | grep "k_strs_write" | cut -c1-100
  done
done
