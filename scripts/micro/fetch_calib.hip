// scripts/micro/fetch_calib.hip -- what does FETCH_SIZE count?  VERDICT r02 (item 3) asked for a calibration of the counter on the
// 32-byte re-read pattern of the deferred UTF-8 check, so that "traffic 1.39 x algorithmic" on sparse NDJSON is a number and not an
// upper bound.  Four read-only kernels over the same 1 GiB buffer (cold L2: 1 GiB >> 32 MiB), each run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace
// with a known number of bytes REQUESTED and a known number of 64-byte / 128-byte lines TOUCHED:
//   k_full16   every lane 16 B, lanes contiguous              -> every byte of the buffer, once
//   k_half32   every lane 32 B (two 16 B loads) of each 64 B  -> half of the bytes, every 64-byte line touched
//   k_quarter  every lane 16 B of each 64 B                   -> a quarter of the bytes, every 64-byte line touched
//   k_dword    every lane 4 B of each 64 B                    -> 1/16 of the bytes, every 64-byte line touched
//   k_half128  every lane 64 B (four loads) of each 128 B     -> half of the bytes, every 128-byte line touched, every other 64-byte line
// hipcc --offload-arch=gfx950 -O3 scripts/micro/fetch_calib.hip -o scripts/micro/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sink_if(v4 a, uint32_t *sink) { if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) { sink[0] = 1; } }

__global__ __launch_bounds__(256) void k_full16(const v4 *__restrict__ in, size_t vecs, uint32_t *sink) {
  v4 acc = {0, 0, 0, 0};
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < vecs; i += size_t(gridDim.x) * 256) { acc ^= in[i]; }
  sink_if(acc, sink);
}
// lane reads `take` consecutive 16-byte vectors at the start of every `stride`-vector group
template <int TAKE, int STRIDE> __global__ __launch_bounds__(256) void k_part(const v4 *__restrict__ in, size_t vecs, uint32_t *sink) {
  v4 acc = {0, 0, 0, 0};
  const size_t groups = vecs / STRIDE;
  for (size_t g = size_t(blockIdx.x) * 256 + threadIdx.x; g < groups; g += size_t(gridDim.x) * 256) {
#pragma unroll
    for (int k = 0; k < TAKE; k++) { acc ^= in[g * STRIDE + k]; }
  }
  sink_if(acc, sink);
}
__global__ __launch_bounds__(256) void k_dword(const uint32_t *__restrict__ in, size_t words, uint32_t *sink) {
  uint32_t acc = 0;
  const size_t groups = words / 16;
  for (size_t g = size_t(blockIdx.x) * 256 + threadIdx.x; g < groups; g += size_t(gridDim.x) * 256) { acc ^= in[g * 16]; }
  if (acc == 0x12345678u) { sink[0] = 1; }
}

int main() {
  const size_t bytes = size_t(1) << 30, vecs = bytes / 16;
  v4 *in;
  uint32_t *sink;
  CK(hipMalloc(&in, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(in, 1, bytes));
  CK(hipDeviceSynchronize());
  const dim3 grid(4096), block(256);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_full16, grid, block, 0, 0, in, vecs, sink);
    hipLaunchKernelGGL((k_part<2, 4>), grid, block, 0, 0, in, vecs, sink);  // k_half32
    hipLaunchKernelGGL((k_part<1, 4>), grid, block, 0, 0, in, vecs, sink);  // k_quarter
    hipLaunchKernelGGL(k_dword, grid, block, 0, 0, reinterpret_cast<const uint32_t *>(in), bytes / 4, sink);
    hipLaunchKernelGGL((k_part<4, 8>), grid, block, 0, 0, in, vecs, sink);  // k_half128
    CK(hipDeviceSynchronize());
  }
  printf("requested bytes: k_full16 %zu, k_part<2,4> %zu, k_part<1,4> %zu, k_dword %zu, k_part<4,8> %zu\n", bytes, bytes / 2, bytes / 4, bytes / 16, bytes / 2);
  return 0;
}
