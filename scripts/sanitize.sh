#!/bin/bash
# scripts/sanitize.sh -- AddressSanitizer and ThreadSanitizer builds of the HOST side of the drop-in boundary: libsjgpu's
# host code (context pool, copy threads of the overlapped path, multi-GPU shard threads), the plug-in shim and
# tests/plugin/plugin_test.cpp, all with the same clang the HIP code is built with (the reference runs its
# own suite under ASan / TSan in CI, cmake/developer-options.cmake:15-89).  Needs /root/reference (build container);
# outputs build/san/plugin_test_{address,thread} + the matching libsjgpu, which travel to the GPU box.
#   bash scripts/sanitize.sh build         (here)
#   bash scripts/sanitize.sh run           (on the GPU box; tests/test_plugin.py::test_sanitizers does this)
set -u
cd "$(dirname "$0")/.."
CLANG=/opt/rocm/lib/llvm/bin/clang++
HIPCC=/opt/rocm/bin/hipcc
S=simdjson_amd/csrc
REF=${SIMDJSON_REFERENCE:-/root/reference}
mkdir -p build/san
case ${1:-build} in
  build)
    for san in address thread; do
      $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=$san -I include -I $S \
        $S/sjgpu_kernels.hip $S/sjgpu_fused.hip $S/sjgpu_small.hip $S/sjgpu_finish.hip $S/sjgpu_strings.hip $S/sjgpu_string_stream.hip $S/sjgpu_tape.hip \
        $S/sjgpu_mgpu.hip $S/sjgpu_comm.hip $S/sjgpu_capi.hip $S/sjgpu_capi_host.hip $S/sjgpu_capi_stage2.hip $S/stage1_finish.cpp -ldl \
        -o build/san/libsjgpu_$san.so || exit 1
      $CLANG -O1 -g -std=c++17 -fsanitize=$san -DSIMDJSON_THREADS_ENABLED=1 -I $REF/include -I $REF/src -c $REF/src/simdjson.cpp -o build/san/simdjson_$san.o || exit 1
      $CLANG -O1 -g -std=c++17 -fsanitize=$san -DSIMDJSON_THREADS_ENABLED=1 -I $REF/include -I $S/plugin -I include \
        tests/plugin/plugin_test.cpp $S/plugin/mi355x_implementation.cpp build/san/simdjson_$san.o -o build/san/plugin_test_$san \
        build/san/libsjgpu_$san.so -Lsimdjson_amd/lib -lsjcorpus -lpthread -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../simdjson_amd/lib' || exit 1
    done
    ls -la build/san ;;
  run)
    # three routes through the host code: defaults; the overlapped path (copy threads, 1 MiB ranges); two shards per
    # document on the multi-GPU path (shard threads; the one device listed twice)
    rc=0
    export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0
    export TSAN_OPTIONS=report_signal_unsafe=0:halt_on_error=0:suppressions=$PWD/scripts/tsan.supp
    : > build/san/address.log; : > build/san/thread.log
    for san in address thread; do
      for route in "X=1" "SJGPU_STREAM_FROM_MB=1 SJGPU_STREAM_CHUNK_MB=1" "SJGPU_DEVICES=0,0 SJGPU_MGPU_FROM_MB=1"; do
        echo "==== $san: $route" >> build/san/$san.log
        env $route timeout 300 build/san/plugin_test_$san --jsonexamples tests/golden/jsonexamples >> build/san/$san.log 2>&1 || rc=1
      done
    done
    echo "asan: runs OK $(grep -c 'plugin test OK' build/san/address.log) of 3, reports $(grep -c 'ERROR: AddressSanitizer' build/san/address.log)"
    echo "tsan: runs OK $(grep -c 'plugin test OK' build/san/thread.log) of 3, reports $(grep -c 'WARNING: ThreadSanitizer' build/san/thread.log)"
    mkdir -p gpurun_out/r2 && cp build/san/address.log build/san/thread.log gpurun_out/r2/ 2>/dev/null
    exit $rc ;;
esac
