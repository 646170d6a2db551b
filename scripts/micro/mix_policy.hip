// scripts/micro/mix_policy.hip -- does a cache policy on the loads or the stores move the ceiling of scripts/micro/mix_copy.hip?  The same mover (reads every
// 64 KiB tile once, writes `ratio` bytes per byte read, 16-byte accesses) with: plain accesses; non-temporal stores (global_store ... nt); non-temporal
// loads; both; stores with sc0 sc1 (write-through).  Every byte is read once and written once: nothing the L2 keeps is ever asked for again.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mix_policy.hip -o scripts/micro/mix_policy.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t v4 __attribute__((ext_vector_type(4)));

template <int LD, int ST> // LD: 0 plain, 1 nt; ST: 0 plain, 1 nt, 2 sc0 sc1, 3 sc1
__global__ __launch_bounds__(256) void k_mix(const v4 *__restrict__ in, v4 *__restrict__ out, size_t tiles, uint32_t out_vecs) {
  for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const v4 *src = in + t * 4096 + threadIdx.x;
    v4 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { v[u] = LD == 1 ? __builtin_nontemporal_load(src + u * 256) : src[u * 256]; }
    v4 acc = v[0];
#pragma unroll
    for (int u = 1; u < 16; u++) { acc ^= v[u]; }
    v4 *dst = out + t * size_t(out_vecs);
    for (uint32_t o = threadIdx.x; o < out_vecs; o += 256) {
      const v4 val = acc + v4{o, o, o, o};
      if (ST == 0) { dst[o] = val; }
      else if (ST == 1) { __builtin_nontemporal_store(val, dst + o); }
      else if (ST == 2) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + o), "v"(val) : "memory"); }
      else { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + o), "v"(val) : "memory"); }
    }
  }
}

int main() {
  const size_t in_bytes = size_t(1) << 30, tiles = in_bytes / 65536;
  v4 *in, *out;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, size_t(5) << 30));
  CK(hipMemset(in, 1, in_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct mix { const char *name; double ratio; };
  const mix mixes[] = {{"amazon NDJSON 0.22", 0.2175}, {"masks only    0.125", 0.125}, {"minify        0.89", 0.891}, {"large_random  1.22", 1.2174}};
  typedef void (*kern)(const v4 *, v4 *, size_t, uint32_t);
  struct var { const char *name; kern k; };
  const var vars[] = {{"plain", k_mix<0, 0>}, {"nt stores", k_mix<0, 1>}, {"nt loads", k_mix<1, 0>}, {"nt both", k_mix<1, 1>}, {"sc0 sc1 stores", k_mix<0, 2>}, {"sc1 stores", k_mix<0, 3>}};
  printf("# 1 GiB in, tiles of 64 KiB, grid 16384 and 4096; ms = median of 9 interleaved rounds; GB/s = (bytes read + bytes written) / time\n");
  for (const mix &m : mixes) {
    const uint32_t out_vecs = uint32_t(m.ratio * 4096 + 0.5);
    const double bytes = double(in_bytes) + double(tiles) * out_vecs * 16;
    for (int grid : {16384, 4096}) {
      float t[6][9];
      for (int rep = 0; rep < 9; rep++) {
        for (int v = 0; v < 6; v++) {
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(vars[v].k, dim3(grid), dim3(256), 0, 0, in, out, tiles, out_vecs);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&t[v][rep], e0, e1));
        }
      }
      for (int v = 0; v < 6; v++) {
        for (int a = 0; a < 9; a++) { for (int b = a + 1; b < 9; b++) { if (t[v][b] < t[v][a]) { float x = t[v][a]; t[v][a] = t[v][b]; t[v][b] = x; } } }
        printf("%-20s grid %5d  %-16s %7.3f ms (best %7.3f)  %6.0f GB/s\n", m.name, grid, vars[v].name, t[v][4], t[v][0], bytes / (t[v][4] * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}
