// scripts/micro/mix_copy.hip -- the ceiling a stage-1 kernel can reach on this part: a kernel that only MOVES the bytes stage 1
// has to move -- reads every 64 KiB tile of the input once, writes OUT_PER_TILE bytes of output for it -- in the access pattern of
// the product kernels (16-byte accesses, a workgroup owns a contiguous tile).  Ratios: large_random 1.22 B out per byte in
// (configs[1]), amazon NDJSON 0.22, twitter-like 0.49, deep nesting 4.0, minify 0.89, read-only 0.
// Also the same with reads and writes issued by DIFFERENT kernels running concurrently on two streams (what the overlapped split
// pipeline does: scan kernel = reads, emit kernel = writes).  hipcc --offload-arch=gfx950 -O3 scripts/micro/mix_copy.hip -o mix_copy.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));

// tile t: read 4096 vectors (64 KiB) at in + t * 4096, write out_vecs vectors at out + t * out_vecs
__global__ __launch_bounds__(256) void k_mix(const v4 *__restrict__ in, v4 *__restrict__ out, size_t tiles, uint32_t out_vecs) {
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const v4 *src = in + t * 4096 + threadIdx.x;
    v4 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { v[u] = src[u * 256]; }
    v4 acc = v[0];
#pragma unroll
    for (int u = 1; u < 16; u++) { acc ^= v[u]; }
    v4 *dst = out + t * size_t(out_vecs);
    for (uint32_t o = threadIdx.x; o < out_vecs; o += 256) { dst[o] = acc + v4{o, o, o, o}; }
  }
}
__global__ __launch_bounds__(256) void k_read_only(const v4 *__restrict__ in, uint32_t *__restrict__ sink, size_t tiles) {
  v4 acc = {0, 0, 0, 0};
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const v4 *src = in + t * 4096 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; u++) { acc ^= src[u * 256]; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) { sink[0] = 1; }
}
__global__ __launch_bounds__(256) void k_write_only(v4 *__restrict__ out, size_t tiles, uint32_t out_vecs) {
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    v4 *dst = out + t * size_t(out_vecs);
    for (uint32_t o = threadIdx.x; o < out_vecs; o += 256) { dst[o] = v4{o, uint32_t(t), 2, 3}; }
  }
}

int main() {
  const size_t in_bytes = size_t(1) << 30, tiles = in_bytes / 65536;
  v4 *in, *out;
  uint32_t *sink;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, size_t(5) << 30));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(in, 1, in_bytes));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, j;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
  struct mix { const char *name; double ratio; };
  const mix mixes[] = {{"read only", 0.0}, {"amazon NDJSON  0.22 B out / B in", 0.2175}, {"twitter-like   0.49", 0.49}, {"minify         0.89", 0.891},
                       {"large_random   1.22 (configs[1])", 1.2174}, {"deep nesting   4.00", 4.0}};
  printf("# 1 GiB in, tiles of 64 KiB; GB/s = (bytes read + bytes written) / time; best of 5; frac = of 8000 GB/s\n");
  for (const mix &m : mixes) {
    const uint32_t out_vecs = uint32_t(m.ratio * 4096 + 0.5);
    const double bytes = double(in_bytes) + double(tiles) * out_vecs * 16;
    for (int grid : {2048, 4096, 16384}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, s1));
        if (out_vecs) { hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, s1, in, out, tiles, out_vecs); }
        else { hipLaunchKernelGGL(k_read_only, dim3(grid), dim3(256), 0, s1, in, sink, tiles); }
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; }
      }
      printf("one kernel   %-36s grid %5d  %7.3f ms  %6.0f GB/s  frac %.3f\n", m.name, grid, best, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9 / 8000);
    }
    if (!out_vecs) { continue; }
    // reads and writes from two kernels on two streams (8 pieces each, reader one piece ahead: the shape of the overlapped split pipeline)
    float best = 1e9f;
    const int pieces = 8;
    const size_t tp = tiles / pieces;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0, s1));
      CK(hipEventRecord(j, s1));
      CK(hipStreamWaitEvent(s2, j, 0));
      for (int p = 0; p < pieces; p++) {
        hipLaunchKernelGGL(k_read_only, dim3(4096), dim3(256), 0, s1, in + size_t(p) * tp * 4096, sink, tp);
        CK(hipEventRecord(j, s1));
        CK(hipStreamWaitEvent(s2, j, 0));
        hipLaunchKernelGGL(k_write_only, dim3(4096), dim3(256), 0, s2, out + size_t(p) * tp * out_vecs, tp, out_vecs);
      }
      CK(hipEventRecord(j, s2));
      CK(hipStreamWaitEvent(s1, j, 0));
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) { best = ms; }
    }
    printf("two kernels  %-36s 8 pieces    %7.3f ms  %6.0f GB/s  frac %.3f\n", m.name, best, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9 / 8000);
  }
  return 0;
}
