#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
fails=0
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  SJGPU_TRACE_ERRORS=1 timeout 120 build/tests/intree_document_stream_tests -a mi355x > gpurun_out/r03_f_ds_$i.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i rc=$rc"; grep -n "sjgpu\]\|Expected\|rror" gpurun_out/r03_f_ds_$i.log | head -5; tail -3 gpurun_out/r03_f_ds_$i.log; fi
done
echo "failures: $fails of 12"
for i in 1 2 3 4 5 6; do
  SJGPU_TRACE_ERRORS=1 timeout 120 build/tests/intree_document_stream_fuzz_tests -a mi355x > gpurun_out/r03_f_fz_$i.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then echo "fuzz run $i rc=$rc"; tail -3 gpurun_out/r03_f_fz_$i.log; fi
done
echo done
