// scripts/micro/valu_rate.hip -- integer VALU issue rate on gfx950 (v_bfi_b32, v_perm_b32, v_and_b32, v_add_u32)
// and streaming read / write / copy bandwidth, to price the stage-1 kernels.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP> __global__ __launch_bounds__(256) void k_valu(uint32_t *out, int iters) {
  uint32_t a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
  uint32_t m = 0xF0F0F0F0u ^ threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t b = a[(i + 1) & 7];
      if (OP == 0) a[i] = (a[i] & m) | (b & ~m);                     // v_bfi_b32
      if (OP == 1) a[i] = __builtin_amdgcn_perm(a[i], b, 0x07050301u); // v_perm_b32
      if (OP == 2) a[i] = (a[i] & b) ^ m;                            // v_and + v_xor (2 ops)
      if (OP == 3) a[i] = a[i] + b;                                  // v_add_u32
      if (OP == 4) a[i] = (a[i] << 3) | b;                           // v_lshl_or_b32
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_read(const uint4 *in, uint32_t *out, size_t n) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4 *outp, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) { outp[i] = make_uint4(uint32_t(i), 1, 2, 3); }
}
__global__ __launch_bounds__(256) void k_copy(const uint4 *in, uint4 *outp, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) { outp[i] = in[i]; }
}
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t *out; CK(hipMalloc(&out, 2048 * 256 * 4));
  const int iters = 4096, grid = 2048; // 8 WGs of 256 per CU
  const char *names[] = {"v_bfi_b32", "v_perm_b32", "v_and+v_xor", "v_add_u32", "v_lshl_or_b32"};
  for (int op = 0; op < 5; op++) {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      if (op == 0) hipLaunchKernelGGL(k_valu<0>, dim3(grid), dim3(256), 0, 0, out, iters);
      if (op == 1) hipLaunchKernelGGL(k_valu<1>, dim3(grid), dim3(256), 0, 0, out, iters);
      if (op == 2) hipLaunchKernelGGL(k_valu<2>, dim3(grid), dim3(256), 0, 0, out, iters);
      if (op == 3) hipLaunchKernelGGL(k_valu<3>, dim3(grid), dim3(256), 0, 0, out, iters);
      if (op == 4) hipLaunchKernelGGL(k_valu<4>, dim3(grid), dim3(256), 0, 0, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double wave_instrs = double(grid) * 4 * iters * 8 * (op == 2 ? 2 : 1);
    double per_simd_per_s = wave_instrs / (best * 1e-3) / 1024.0; // 256 CUs x 4 SIMDs
    printf("%-14s %8.3f ms  %.3e wave-instr/s/SIMD  => %.2f cycles per wave64 instr at 2.4 GHz (%.2f at 2.1)\n", names[op], best,
           per_simd_per_s, 2.4e9 / per_simd_per_s, 2.1e9 / per_simd_per_s);
  }
  size_t bytes = size_t(2) << 30; uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
  size_t n = bytes / 16;
  for (int which = 0; which < 3; which++) {
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, out, n);
      if (which == 1) hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, b, n);
      if (which == 2) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, a, b, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-6s 2 GiB: %7.3f ms  %.0f GB/s (%s)\n", which == 0 ? "read" : which == 1 ? "write" : "copy", best,
           (which == 2 ? 2.0 : 1.0) * bytes / (best * 1e-3) / 1e9, which == 2 ? "read+write bytes" : "bytes");
  }
  return 0;
}
