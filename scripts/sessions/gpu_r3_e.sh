#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
SJGPU_TRACE_ERRORS=1 timeout 300 build/tests/plugin_test --jsonexamples tests/golden/jsonexamples > gpurun_out/r03_e_plugin_test.log 2>&1; echo "plugin_test rc=$?"; tail -12 gpurun_out/r03_e_plugin_test.log
SJGPU_TRACE_ERRORS=1 timeout 300 build/tests/intree_document_stream_tests -a mi355x > gpurun_out/r03_e_intree_ds.log 2>&1; echo "intree ds rc=$?"; tail -6 gpurun_out/r03_e_intree_ds.log; grep -n "sjgpu\]" gpurun_out/r03_e_intree_ds.log | head
