#!/bin/bash
# scripts/build_variant.sh <name> [-DDEFINE ...] -- the working tree's libsjgpu.so built into build/ab/libsjgpu_<name>.so (for scripts/lib_ab.py / tape_ab.py);
# scripts/build_variant.sh <name> --rev <git rev> [-D...]: the same from a commit's sources
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT
if [ "${1:-}" = "--rev" ]; then
  SRC=$(mktemp -d); git -C "$ROOT" archive "$2" simdjson_amd/csrc include | tar -x -C "$SRC"; shift 2
fi
mkdir -p "$ROOT/build/ab"
cd "$SRC/simdjson_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread "$@" -I "$SRC/include" -I . sjgpu_kernels.hip sjgpu_fused.hip sjgpu_small.hip sjgpu_finish.hip \
  sjgpu_strings.hip sjgpu_string_stream.hip sjgpu_tape.hip sjgpu_mgpu.hip sjgpu_comm.hip sjgpu_capi.hip sjgpu_capi_host.hip sjgpu_capi_stage2.hip stage1_finish.cpp -o "$ROOT/build/ab/libsjgpu_$NAME.so" -ldl
echo "built build/ab/libsjgpu_$NAME.so"
