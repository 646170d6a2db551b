#!/bin/bash
# round 6, session N: the staged token front (k_tok_stage: classification, atoms and numbers from one staged pass over the document) -- the tape tests, then
# base (HEAD 258d799) against the new front with 4 KiB and 8 KiB windows in one process, then a kernel trace of the new library on both documents
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 1400 -p no:cacheprovider -k "tape or stage2 or number or parse" > $O/r6n_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r6n_pytest.log
timeout 900 python scripts/tape_ab.py base=build/ab/libsjgpu_base.so w4k=build/ab/libsjgpu_w4k.so w8k=build/ab/libsjgpu_w8k.so > $O/r6n_tape_ab.txt 2> $O/r6n_tape_ab.err; echo "ab rc=$?"
grep -v "^{" $O/r6n_tape_ab.txt; tail -3 $O/r6n_tape_ab.err
for kind in twitter_like large_random; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r6n_$kind -o t -- python $GRAFT_REPO_ROOT/scripts/tape_once.py $kind > $O/r6n_trace_$kind.log 2>&1); echo "trace $kind rc=$?"
  db=$(ls $O/prof_r6n_$kind/*/*.db $O/prof_r6n_$kind/*.db 2>/dev/null | head -1)
  python scripts/rocpd_summary.py $db > $O/r6n_kernels_$kind.txt 2>&1; head -40 $O/r6n_kernels_$kind.txt
done
