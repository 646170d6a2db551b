#!/bin/bash
set -u
exec < /dev/null
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "string_stream or every_escape or string_buffer_is or stage2_device" -p no:cacheprovider > gpurun_out/r3m_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3m_tests.log
