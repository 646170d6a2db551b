// scripts/micro/bw_variants.hip -- what streaming read / write / copy reach on gfx950 with different access recipes
// (plain vs nontemporal, 1 vs 4 vectors in flight per lane, grid sizes).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));

template <int NT, int UNROLL> __global__ __launch_bounds__(256) void k_copy(const v4 *in, v4 *out, size_t n) {
  const size_t stride = size_t(gridDim.x) * 256;
  size_t i = blockIdx.x * 256ull + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    v4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) out[i] = in[i];
}
template <int NT, int UNROLL> __global__ __launch_bounds__(256) void k_write(v4 *out, size_t n) {
  const size_t stride = size_t(gridDim.x) * 256;
  size_t i = blockIdx.x * 256ull + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { v4 v = {uint32_t(i), 1u, 2u, uint32_t(u)}; if (NT) __builtin_nontemporal_store(v, out + i + u * stride); else out[i + u * stride] = v; }
  }
  for (; i < n; i += stride) out[i] = v4{1, 2, 3, 4};
}
// contiguous-per-workgroup variants (each workgroup owns a contiguous 64 KiB span, like our tiles)
template <int NT> __global__ __launch_bounds__(256) void k_copy_tiled(const v4 *in, v4 *out, size_t n) {
  const size_t tile = 4096; // v4 per workgroup pass = 64 KiB
  for (size_t t = blockIdx.x; t * tile < n; t += gridDim.x) {
    const size_t b = t * tile + threadIdx.x;
    v4 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = (b + u * 256 < n) ? (NT ? __builtin_nontemporal_load(in + b + u * 256) : in[b + u * 256]) : v4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 16; u++) if (b + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], out + b + u * 256); else out[b + u * 256] = v[u]; }
  }
}
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t bytes = size_t(2) << 30; v4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
  size_t n = bytes / 16;
  auto time = [&](auto launch, const char *name, double factor) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-44s %7.3f ms  %6.0f GB/s\n", name, best, factor * bytes / (best * 1e-3) / 1e9);
  };
  for (int grid : {2048, 4096, 16384}) {
    printf("-- grid %d x 256\n", grid);
    time([&] { hipLaunchKernelGGL((k_copy<0, 1>), dim3(grid), dim3(256), 0, 0, a, b, n); }, "copy plain x1 (read+write bytes)", 2);
    time([&] { hipLaunchKernelGGL((k_copy<0, 4>), dim3(grid), dim3(256), 0, 0, a, b, n); }, "copy plain x4", 2);
    time([&] { hipLaunchKernelGGL((k_copy<1, 4>), dim3(grid), dim3(256), 0, 0, a, b, n); }, "copy nontemporal x4", 2);
    time([&] { hipLaunchKernelGGL((k_copy_tiled<0>), dim3(grid), dim3(256), 0, 0, a, b, n); }, "copy tiled 64KiB plain", 2);
    time([&] { hipLaunchKernelGGL((k_copy_tiled<1>), dim3(grid), dim3(256), 0, 0, a, b, n); }, "copy tiled 64KiB nontemporal", 2);
    time([&] { hipLaunchKernelGGL((k_write<0, 4>), dim3(grid), dim3(256), 0, 0, b, n); }, "write plain x4", 1);
    time([&] { hipLaunchKernelGGL((k_write<1, 4>), dim3(grid), dim3(256), 0, 0, b, n); }, "write nontemporal x4", 1);
  }
  return 0;
}
